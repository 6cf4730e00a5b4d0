// DynamicsStep.hpp -- IModelStep implementation of the dynamics core on one GPU: per time step the mEVP
// stress/velocity sub-cycle followed by the DG2 transport of the mean thickness H and concentration A,
// optionally preceded by the column thermodynamics (BASELINE config 5 coupling).  Everything numerical
// happens behind the C ABI (include/nsdg.h); this class owns the device arrays and the call sequence.
//
// The reference has no dynamics component (CMakeLists.txt:43-46); the class plugs into the reference's
// batched seam exactly like HipStep does and is selected with
//     [Modules] Nextsim::IModelStep = Nextsim::DynamicsStep
// Configuration keys (all optional):
//     dynamics.domain_size   side of the square box in m        (512e3)
//     dynamics.nsub          mEVP sub-iterations per step        (120)
//     dynamics.alpha/.beta   mEVP parameters (0 = stability bound of the mesh, see stableAlpha())
//     dynamics.thermodynamics  run the column physics first       (false)
// The structure's cell means initialise the DG fields: H <- hice, A <- cice (coefficient 0; higher
// coefficients start at zero) and receive them back at stop().  Ocean current and wind come from the
// device-side forcing provider nsdg_boxtest_forcing (the wind is re-evaluated at every step's model time).
#pragma once
#include <vector>

#include "Configured.hpp"
#include "Iterator.hpp"

struct nsdg_ctx;

namespace Nextsim {

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

    //! diagnostics after stop(): max |u|, sum of the H and A cell means
    double maxSpeed() const { return m_umax; }
    double sumH() const { return m_sumH; }
    double sumA() const { return m_sumA; }
    static double stableAlpha(double h, double dt);

private:
    void release();
    IStructure* pStructure = nullptr;
    nsdg_ctx* ctx = nullptr;
    double* d_block = nullptr;
    std::vector<double*> d; // named sub-arrays of the block
    int nxf = 0, nyf = 0; // fast / slow grid dimensions as the dynamics ABI names them
    long N = 0, NN = 0;
    double L = 512e3, alpha = 0, beta = 0;
    int nsub = 120;
    bool thermo = false;
    long m_steps = 0;
    double m_time = 0; // model time of the next step [s]
    double m_umax = 0, m_sumH = 0, m_sumA = 0;
};

} // namespace Nextsim
