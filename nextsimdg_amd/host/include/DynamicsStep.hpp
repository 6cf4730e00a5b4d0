// DynamicsStep.hpp -- IModelStep implementation of the dynamics core: per time step the mEVP stress/velocity
// sub-cycle followed by the DG2 transport of the mean thickness H and concentration A, optionally preceded by
// the column thermodynamics (BASELINE config 5 coupling).  Everything numerical happens behind the C ABI
// (include/nsdg.h); this class owns the device arrays and the call sequence.
//
// The reference has no dynamics component (CMakeLists.txt:43-46); the class plugs into the reference's
// batched seam (IModelStep, core/src/include/IModelStep.hpp:16-34) exactly like HipStep does and is selected with
//     [Modules] Nextsim::IModelStep = Nextsim::DynamicsStep
// The rectangular mesh is split into ROW BLOCKS with ghost rows; every block has its own context and is advanced
// by the row-block drivers of the ABI (nsdg_rb_mevp_run / nsdg_rb_transport_run: kernel passes and ghost-row
// exchanges of a step in one call each).  Three ways to run:
//   * one block (default): the whole grid on one GPU;
//   * dynamics.row_blocks = N: N blocks in THIS process, one host thread each, ghost rows through the in-process
//     transport (nsdg_comm_init_local) -- devices round-robin over dynamics.devices (default: all on device 0);
//   * one process per GPU under a launcher that sets WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT
//     (Rendezvous.hpp): every process owns ONE block, ghost rows over RCCL send/recv (nsdg_comm_init).  Every rank
//     reads the same initial state; at the end the owned rows of every rank travel to rank 0 (TCP, once per run),
//     which writes ONE restart file -- the same file a single-process run writes.
// Configuration keys (all optional):
//     dynamics.domain_size   side of the square box in m        (512e3)
//     dynamics.nsub          mEVP sub-iterations per step        (120)
//     dynamics.subcycle      how the sub-cycle satisfies its stability bound (nsdg_mevp_stable_params, include/nsdg.h): adaptive (default
//                            since round 6: local, solution-adaptive alpha and beta at the literature's Delta_min, alpha_min = 50) |
//                            adaptive_converged (alpha_min from the mesh so that the sub-cycle converges; wants a time step that fits
//                            the mesh) | keep_alpha (round 5:
//                            uniform alpha = beta = dynamics.alpha or 1500, Delta_min raised to what the mesh needs for it) |
//                            keep_delta_min (rounds 1-4: uniform alpha = beta from the bound for dynamics.delta_min or 2e-9)
//     dynamics.alpha/.beta   uniform mEVP parameters (keep_alpha; 0 = 1500, the BASELINE's value); given WITHOUT dynamics.subcycle they
//                            select keep_alpha and are used as they are
//     dynamics.delta_min     regularisation of Delta [1/s] (0 = the literature's 2e-9)
//     dynamics.closure       ridging cap + scaling limiter in the transport, free drift at ice-free nodes (default true)
//     dynamics.min_conc/.min_thick   the ice-free-node rule's thresholds (defaults 1e-12, 0.01: the column model's cut-off)
//     dynamics.thermodynamics  run the column physics first       (false)
//     dynamics.forcing       thermodynamic forcing: host (the structure's planes, constant in time) | dummy | winter
//                            (generated on the device at every step's model time, wind speed from the dynamics' wind)
//     dynamics.row_blocks, dynamics.devices, dynamics.passes_per_exchange (2), dynamics.overlap (true),
//     dynamics.graph (false), dynamics.loopback_world (0: off; N: rehearse an interior block of N on one GPU with
//     real RCCL send/recv to the rank itself -- values wrap around, for timing and call-path checks only)
// The structure's cell means initialise the DG fields: H <- hice, A <- cice (coefficient 0; higher coefficients
// start at zero) and receive them back at stop().  Ocean current and wind come from the device-side forcing
// provider nsdg_boxtest_forcing (the wind is re-evaluated at every step's model time).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "Configured.hpp"
#include "IStructure.hpp"
#include "Iterator.hpp"

struct nsdg_ctx;

namespace Nextsim {

class DynamicsBlock; // one row block: context, device arrays, driver plans (DynamicsStep.cpp)

class DynamicsStep : public IModelStep, public Configured<DynamicsStep> {
public:
    DynamicsStep();
    ~DynamicsStep() override;
    DynamicsStep(const DynamicsStep&) = delete;
    DynamicsStep& operator=(const DynamicsStep&) = delete;

    void configure() override;
    void setInitialData(IStructure& dataStructure) override { pStructure = &dataStructure; }
    void writeRestartFile(const std::string& filePath) override;
    void init() override;
    void start(const Iterator::TimePoint& startTime) override;
    void iterate(const Iterator::Duration& dt) override;
    void stop(const Iterator::TimePoint& stopTime) override;
    long launches() const override { return m_steps; }

    //! diagnostics after stop(): max |u|, sum of the H and A cell means (of the rows this process owns)
    double maxSpeed() const { return m_umax; }
    double sumH() const { return m_sumH; }
    double sumA() const { return m_sumA; }
    int blocks() const { return (int)m_blocks.size(); }
    //! the sub-cycle parameters this configuration runs with on cells of size h and a time step dt (nsdg_mevp_stable_params: the one copy
    //! of the stability rule); returns the creep threshold in percent per day beside them
    struct SubcycleChoice {
        std::string mode;
        double alpha, beta, deltaMin, aevpC, aevpAlphaMin, creepPercentPerDay;
    };
    SubcycleChoice subcycleChoice(double h, double dt) const;

    //! rows [r0, r1) of block `rank` of `world` (the same split as nextsimdg_amd/rowblock.py split_rows)
    static void splitRows(int ny, int world, int rank, int& r0, int& r1);

    //! The prognostic planes a run changes, in the order they travel to rank 0 for the restart file.
    static std::vector<std::vector<double>*> restartPlanes(FieldStore& f, bool thermodynamics);
    //! rows [r0, r1) (nx values each) of every plane, one after the other -- what a rank sends ...
    static std::vector<double> packRows(FieldStore& f, bool thermodynamics, int nx, int r0, int r1);
    //! ... and how rank 0 puts it into its structure; throws std::runtime_error when the size is not that of the rows
    static void placeRows(FieldStore& f, bool thermodynamics, int nx, int r0, int r1, const double* data, std::size_t count);

private:
    void release();
    template <class F> void forEachBlock(F&& f); //!< one thread per block when there are several
    IStructure* pStructure = nullptr;
    std::vector<std::unique_ptr<DynamicsBlock>> m_blocks;
    int nxf = 0, nyf = 0; // fast / slow grid dimensions as the dynamics ABI names them
    double L = 512e3, alpha = 0, beta = 0;
    int nsub = 120, rowBlocks = 1, passesPerExchange = 2, loopbackWorld = 0;
    bool thermo = false, overlap = true, graph = false, m_inited = false;
    bool closure = true; // ridging cap + scaling limiter in the transport, free drift at ice-free nodes (dynamics.closure)
    double deltaMin = 0; // dynamics.delta_min; 0: the literature's 2e-9 (keep_alpha: raised to what the mesh needs)
    std::string subcycle = "adaptive"; // dynamics.subcycle
    double minConc = 1e-12, minThick = 0.01; // ice-free-node rule (dynamics.min_conc / min_thick; the column model's cut-off values)
    std::string forcing = "host", devices;
    int m_world = 1, m_rank = 0; // multi-process run (one block per process)
    long m_steps = 0;
    double m_time = 0; // model time of the next step [s]
    double m_umax = 0, m_sumH = 0, m_sumA = 0;
};

} // namespace Nextsim
