#include "DynamicsStep.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "../../../include/nsdg.h"
#include "ModuleLoader.hpp"
#include "PhysicsModules.hpp"
#include "Rendezvous.hpp"
#include "Timer.hpp"

namespace Nextsim {

namespace {
bool g_multiProcess = false; // one row block per process, ghost rows over RCCL
void check(int rc, const char* what)
{
    if (rc == NSDG_ERR_COMM && g_multiProcess) {
        // a neighbour rank has died (the bounded wait of nsdg_ctx_synchronize ran out): the device streams of this rank may
        // never drain, so no destructor may run (hipFree synchronises the device) -- leave at once with a non-zero
        // status; the launcher ends the remaining ranks
        std::fprintf(stderr, "nextsim_amd rank: %s: %s\n", what, nsdg_last_error());
        std::fflush(stderr);
        std::_Exit(3);
    }
    if (rc != NSDG_OK)
        throw std::runtime_error(std::string(what) + ": " + nsdg_last_error());
}
void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
std::atomic<long> g_localGroup { 1 }; // ids of the in-process communicator groups of this process

// The stress lives on the device in the tiled layout of include/nsdg.h (tiles of 64 elements of a row, the 8 coefficients of a tile
// together, in pairs interleaved by element); a restart file holds coefficient planes [8][rows][nx].
inline std::size_t tiledIndex(int nx, int ix, int iy, int c)
{
    const std::size_t ntx = (std::size_t)(nx + 63) / 64;
    return (((std::size_t)iy * ntx + (std::size_t)(ix >> 6)) * 8) * 64 + (std::size_t)(c / 2) * 128 + 2 * (std::size_t)(ix & 63) + (std::size_t)(c % 2);
}
// rows [row0, row0 + rows) of the tiled array `t` (local array of nx columns) -> planes[c * planeStride + (dstRow0 + r) * nx + ix]
void untileRows(const std::vector<double>& t, int nx, int row0, int rows, double* planes, std::size_t planeStride, std::size_t dstRow0)
{
    for (int c = 0; c < 8; ++c)
        for (int r = 0; r < rows; ++r)
            for (int ix = 0; ix < nx; ++ix)
                planes[c * planeStride + (dstRow0 + r) * nx + ix] = t[tiledIndex(nx, ix, row0 + r, c)];
}
void tileRows(const double* planes, std::size_t planeStride, std::size_t srcRow0, int nx, int rows, std::vector<double>& t)
{
    for (int c = 0; c < 8; ++c)
        for (int r = 0; r < rows; ++r)
            for (int ix = 0; ix < nx; ++ix)
                t[tiledIndex(nx, ix, r, c)] = planes[c * planeStride + (srcRow0 + r) * nx + ix];
}

// device arrays of a block; the advected fields, the stress and the velocity exist twice (ping-pong)
enum Arr { H0, A0, H1, A1, T2H, T2A, S11a, S12a, S22a, S11b, S12b, S22b, PG, VXDG, VYDG, UNX, UNY, Ua, Va, Ub, Vb, UA, VA, UO, VO, PACKED, COL, NARR };
// column planes inside COL
enum Col { C_HSNOW, C_TICE, C_SST, C_SSS, C_TAIR, C_TDEW, C_SLP, C_QSW, C_QLW, C_MLD, C_SNOWFALL, C_WIND, C_NEWICE, NCOL };
} // namespace

// ------------------------------------------------------------------------------------------------ one row block
class DynamicsBlock {
public:
    // geometry: global rows [r0, r1) owned, [lo, hi) held locally; j0/j1 = owned local rows
    int rank = 0, world = 1, device = 0;
    int nx = 0, nyGlobal = 0, r0 = 0, r1 = 0, lo = 0, hi = 0, ny = 0, j0 = 0, j1 = 0;
    int depthBelow = 0, depthAbove = 0, peerBelow = -1, peerAbove = -1;
    long N = 0, NN = 0;
    nsdg_ctx* ctx = nullptr;
    double* block = nullptr;
    std::vector<double*> d;
    nsdg_rb_mevp* mevp = nullptr;
    nsdg_rb_transport* transport = nullptr;
    int par = 0, tpar = 0; // which buffers hold the velocity/stress iterate and the advected state

    ~DynamicsBlock() { release(); }
    void release()
    {
        if (ctx)
            (void)hipSetDevice(device);
        nsdg_rb_mevp_destroy(mevp);
        nsdg_rb_transport_destroy(transport);
        mevp = nullptr, transport = nullptr;
        if (block)
            (void)hipFree(block);
        block = nullptr;
        if (ctx)
            nsdg_ctx_destroy(ctx); // finalises the communicator too
        ctx = nullptr;
    }
    double* curH() const { return d[tpar == 0 ? H0 : H1]; }
    double* curA() const { return d[tpar == 0 ? A0 : A1]; }
    double* curU() const { return d[par == 0 ? Ua : Ub]; }
    double* curV() const { return d[par == 0 ? Va : Vb]; }
    double* col(int k) const { return d[COL] + (long)k * N; }
};

template <>
const std::map<int, std::string> Configured<DynamicsStep>::keyMap = { { 0, "dynamics.domain_size" }, { 1, "dynamics.nsub" },
    { 2, "dynamics.alpha" }, { 3, "dynamics.beta" }, { 4, "dynamics.thermodynamics" }, { 5, "dynamics.row_blocks" },
    { 6, "dynamics.passes_per_exchange" }, { 7, "dynamics.overlap" }, { 8, "dynamics.graph" }, { 9, "dynamics.forcing" },
    { 10, "dynamics.devices" }, { 11, "dynamics.loopback_world" }, { 12, "dynamics.closure" }, { 13, "dynamics.min_conc" },
    { 14, "dynamics.min_thick" }, { 15, "dynamics.delta_min" }, { 16, "dynamics.subcycle" } };

DynamicsStep::DynamicsStep() = default;
DynamicsStep::~DynamicsStep() { release(); }

void DynamicsStep::release() { m_blocks.clear(); }

DynamicsStep::SubcycleChoice DynamicsStep::subcycleChoice(double h, double dt) const
{ // the stability rule lives in the library (nsdg_mevp_stable_params): the host only says which of its forms it wants
    nsdg_mevp_params p;
    nsdg_mevp_default_params(&p);
    if (deltaMin > 0)
        p.delta_min = deltaMin;
    const std::string mode = subcycle;
    if (mode == "keep_alpha") {
        p.alpha = alpha > 0 ? alpha : 1500.;
        check(nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_ALPHA, h, dt), "nsdg_mevp_stable_params");
        if (beta > 0)
            p.beta = beta;
    } else if (mode == "keep_delta_min")
        check(nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_DELTA_MIN, h, dt), "nsdg_mevp_stable_params");
    else
        check(nsdg_mevp_stable_params(&p, mode == "adaptive_converged" ? NSDG_SUBCYCLE_ADAPTIVE_CONVERGED : NSDG_SUBCYCLE_ADAPTIVE, h, dt), "nsdg_mevp_stable_params");
    return SubcycleChoice { mode, p.alpha, p.beta, p.delta_min, p.aevp_c, p.aevp_alpha_min, nsdg_mevp_creep_percent_per_day(&p) };
}

void DynamicsStep::splitRows(int ny, int world, int rank, int& r0, int& r1)
{
    r0 = (int)(((long)rank * ny) / world);
    r1 = (int)(((long)(rank + 1) * ny) / world);
}

void DynamicsStep::configure()
{
    L = getConfiguration(keyMap.at(0), 512e3);
    nsub = getConfiguration(keyMap.at(1), 120);
    alpha = getConfiguration(keyMap.at(2), 0.);
    beta = getConfiguration(keyMap.at(3), 0.);
    thermo = getConfiguration(keyMap.at(4), false);
    rowBlocks = getConfiguration(keyMap.at(5), 1);
    passesPerExchange = getConfiguration(keyMap.at(6), 2); // the best of the rehearsed 8-block runs at both modelled link rates
    overlap = getConfiguration(keyMap.at(7), true);
    graph = getConfiguration(keyMap.at(8), false);
    forcing = getConfiguration(keyMap.at(9), std::string("host"));
    devices = getConfiguration(keyMap.at(10), std::string(""));
    loopbackWorld = getConfiguration(keyMap.at(11), 0);
    // the closure that keeps the dynamics inside the physical range (include/nsdg.h "INPUT DOMAIN AND CLOSURE"): ridging cap and
    // scaling limiter at the end of a transport step, free drift at ice-free nodes with the column model's cut-off values
    // (nextsim_thermo.min_conc / min_thick, physics/src/modules/NextsimPhysics.cpp:81-82); closure = false: the bare scheme
    closure = getConfiguration(keyMap.at(12), true);
    minConc = getConfiguration(keyMap.at(13), 1e-12);
    minThick = getConfiguration(keyMap.at(14), 0.01);
    // sub-cycle parameters (include/nsdg.h: nsdg_mevp_stable_params): adaptive alpha / beta unless the configuration names a uniform
    // alpha (then keep_alpha: round 5) or says otherwise
    deltaMin = getConfiguration(keyMap.at(15), 0.);
    subcycle = getConfiguration(keyMap.at(16), std::string(alpha > 0 ? "keep_alpha" : "adaptive"));
    if (subcycle != "adaptive" && subcycle != "adaptive_converged" && subcycle != "keep_alpha" && subcycle != "keep_delta_min")
        throw std::invalid_argument("dynamics.subcycle must be adaptive, adaptive_converged, keep_alpha or keep_delta_min");
    if (rowBlocks < 1 || passesPerExchange < 1 || nsub < 0)
        throw std::invalid_argument("dynamics.row_blocks and dynamics.passes_per_exchange must be >= 1, dynamics.nsub >= 0");
    if (forcing != "host" && forcing != "dummy" && forcing != "winter")
        throw std::invalid_argument("dynamics.forcing must be host, dummy or winter");
    if (loopbackWorld != 0 && loopbackWorld < 3)
        throw std::invalid_argument("dynamics.loopback_world needs an interior block: at least 3");
}

void DynamicsStep::init()
{
    configure();
    const RankEnvironment env = RankEnvironment::fromEnv();
    m_world = env.world, m_rank = env.rank;
    g_multiProcess = m_world > 1;
    if (m_world > 1 && (rowBlocks > 1 || loopbackWorld))
        throw std::invalid_argument("a multi-process run (WORLD_SIZE > 1) owns one row block per process: leave dynamics.row_blocks / loopback_world unset");
    if (rowBlocks > 1 && loopbackWorld)
        throw std::invalid_argument("dynamics.row_blocks and dynamics.loopback_world exclude each other");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw std::runtime_error(std::string("DynamicsStep::init: no HIP device available"));
    if (thermo) { // the column-physics plugins read their keys once (shared static instances, as in the reference)
        tryConfigure(ModuleLoader::getLoader().getImplementation<IPhysics1d>());
        tryConfigure(ModuleLoader::getLoader().getImplementation<IFreezingPoint>());
    }
    m_inited = true;
}

template <class F> void DynamicsStep::forEachBlock(F&& f)
{
    if (m_blocks.size() == 1) {
        f(*m_blocks[0]);
        return;
    }
    // the in-process transport hands ghost rows over between host threads: one thread per block
    std::vector<std::thread> threads;
    std::vector<std::exception_ptr> errors(m_blocks.size());
    for (std::size_t k = 0; k < m_blocks.size(); ++k)
        threads.emplace_back([&, k] {
            try {
                f(*m_blocks[k]);
            } catch (...) {
                errors[k] = std::current_exception();
            }
        });
    for (auto& t : threads)
        t.join();
    for (auto& e : errors)
        if (e)
            std::rethrow_exception(e);
}

void DynamicsStep::start(const Iterator::TimePoint& startTime)
{
    ScopedTimer timer("start (upload)");
    if (!pStructure)
        throw std::logic_error("DynamicsStep: setInitialData() was not called");
    if (!m_inited)
        init();
    FieldStore& f = pStructure->fields();
    nxf = pStructure->ny(); // fast dimension of the x-major index i*ny + j
    nyf = pStructure->nx();
    m_time = (double)startTime;

    // ---- the blocks of this process
    const bool loop = loopbackWorld > 0;
    const int world = m_world > 1 ? m_world : (loop ? loopbackWorld : rowBlocks);
    int ndev = 1;
    (void)hipGetDeviceCount(&ndev);
    std::vector<int> devs;
    {
        std::stringstream ss(devices);
        std::string item;
        while (std::getline(ss, item, ','))
            if (!item.empty())
                devs.push_back(std::stoi(item));
        if (devs.empty())
            devs.push_back(m_world > 1 ? RankEnvironment::fromEnv().localRank % std::max(ndev, 1) : 0);
        for (int dv : devs)
            if (dv < 0 || dv >= ndev)
                throw std::invalid_argument("dynamics.devices names a device that does not exist");
    }
    // ghost depth: (v k, v k - 1) for k passes of the v-iterations-per-pass kernel between two exchanges
    const int v = NSDG_MEVP_DEFAULT_VARIANT; // the library's default kernel: sub-iterations per pass
    int k = world > 1 ? std::max(1, std::min(passesPerExchange, (nyf / world) / 16)) : 1;
    const int depthBelow = world > 1 ? v * k : 0, depthAbove = world > 1 ? v * k - 1 : 0;
    if (world > 1 && nyf < world * 2 * std::max(depthBelow, 1))
        throw std::invalid_argument("too few element rows per block for the ghost depth");
    m_blocks.clear();
    const long group = g_localGroup++;
    std::vector<int> ranks;
    if (m_world > 1)
        ranks.push_back(m_rank);
    else if (loop)
        ranks.push_back(loopbackWorld / 2);
    else
        for (int r = 0; r < world; ++r)
            ranks.push_back(r);
    for (std::size_t i = 0; i < ranks.size(); ++i) {
        auto b = std::make_unique<DynamicsBlock>();
        b->rank = ranks[i], b->world = world, b->device = devs[i % devs.size()];
        b->nx = nxf, b->nyGlobal = nyf;
        splitRows(nyf, world, b->rank, b->r0, b->r1);
        b->depthBelow = depthBelow, b->depthAbove = depthAbove;
        const bool below = b->rank > 0, above = b->rank < world - 1;
        b->lo = b->r0 - (below ? depthBelow : 0), b->hi = b->r1 + (above ? depthAbove : 0);
        b->ny = b->hi - b->lo;
        b->j0 = b->r0 - b->lo, b->j1 = b->r1 - b->lo;
        b->peerBelow = below ? (loop ? 0 : b->rank - 1) : -1;
        b->peerAbove = above ? (loop ? 0 : b->rank + 1) : -1;
        m_blocks.push_back(std::move(b));
    }
    // the RCCL communicator id travels before anything else (multi-process run)
    unsigned char commId[NSDG_COMM_ID_BYTES];
    std::memset(commId, 0, sizeof commId);
    if (m_world > 1 || loop) {
        if (m_rank == 0)
            check(nsdg_comm_unique_id(commId), "nsdg_comm_unique_id");
        if (m_world > 1)
            broadcastFromRankZero(RankEnvironment::fromEnv(), commId, sizeof commId);
    }

    const double hx = L / nxf, hy = L / nyf;
    forEachBlock([&](DynamicsBlock& b) {
        checkHip(hipSetDevice(b.device), "hipSetDevice");
        check(nsdg_ctx_create(b.device, nullptr, &b.ctx), "nsdg_ctx_create");
        if (world > 1) {
            if (m_world > 1)
                check(nsdg_comm_init(b.ctx, m_rank, m_world, commId), "nsdg_comm_init");
            else if (loop)
                check(nsdg_comm_init(b.ctx, 0, 1, commId), "nsdg_comm_init (loopback)");
            else
                check(nsdg_comm_init_local(b.ctx, group, b.rank, b.world), "nsdg_comm_init_local");
        }
        if (thermo) {
            nsdg_column_params p;
            IPhysics1d& phys = ModuleLoader::getLoader().getImplementation<IPhysics1d>();
            phys.describe(p);
            check(nsdg_column_params_set(b.ctx, &p), "nsdg_column_params_set");
        }
        b.N = (long)b.nx * b.ny;
        b.NN = (long)(2 * b.nx + 1) * (2 * b.ny + 1);
        const long N = b.N, NN = b.NN;
        const long TS = nsdg_tiled_len(b.nx, b.ny, 8), TP = nsdg_tiled_len(b.nx, b.ny, 9);
        long sizes[NARR];
        for (int a = 0; a < NARR; ++a)
            sizes[a] = 0;
        for (int a : { H0, A0, H1, A1, T2H, T2A, VXDG, VYDG })
            sizes[a] = 6 * N;
        for (int a : { S11a, S12a, S22a, S11b, S12b, S22b })
            sizes[a] = TS;
        sizes[PG] = TP;
        sizes[UNX] = 3L * (b.nx + 1) * b.ny, sizes[UNY] = 3L * b.nx * (b.ny + 1);
        for (int a : { Ua, Va, Ub, Vb, UA, VA, UO, VO })
            sizes[a] = NN;
        sizes[PACKED] = 8 * NN;
        sizes[COL] = (long)NCOL * N;
        long total = 0;
        for (long s : sizes)
            total += (s + 1) & ~1L; // keep every sub-array 16-byte aligned
        checkHip(hipMalloc(reinterpret_cast<void**>(&b.block), total * sizeof(double)), "DynamicsStep: hipMalloc");
        checkHip(hipMemset(b.block, 0, total * sizeof(double)), "DynamicsStep: hipMemset");
        b.d.assign(NARR, nullptr);
        long off = 0;
        for (int a = 0; a < NARR; ++a) {
            b.d[a] = b.block + off;
            off += (sizes[a] + 1) & ~1L;
        }
        check(nsdg_grid_set(b.ctx, b.nx, b.ny, hx, hy), "nsdg_grid_set");
        check(nsdg_block_set(b.ctx, b.lo, b.nyGlobal), "nsdg_block_set");
        // cell means -> DG coefficient 0 (the local rows, ghost rows included, are one contiguous slice)
        const std::size_t first = (std::size_t)b.lo * b.nx;
        checkHip(hipMemcpy(b.d[H0], f.hice.data() + first, N * sizeof(double), hipMemcpyHostToDevice), "upload H");
        checkHip(hipMemcpy(b.d[A0], f.cice.data() + first, N * sizeof(double), hipMemcpyHostToDevice), "upload A");
        if (f.dyn.present) {
            // a restart: the state a dynamics run left (FieldStore::dyn) -- higher DG2 coefficients, velocity, stress -- for the local
            // rows, ghost rows included (they hold what an exchange would deliver: the neighbours' own values)
            const DynamicsState& dy = f.dyn;
            const std::size_t NG = (std::size_t)nxf * nyf;
            for (int c = 1; c < 6; ++c) {
                checkHip(hipMemcpy(b.d[H0] + (long)c * N, dy.hdg.data() + (std::size_t)(c - 1) * NG + first, N * sizeof(double), hipMemcpyHostToDevice), "upload H (DG)");
                checkHip(hipMemcpy(b.d[A0] + (long)c * N, dy.adg.data() + (std::size_t)(c - 1) * NG + first, N * sizeof(double), hipMemcpyHostToDevice), "upload A (DG)");
            }
            const std::size_t nn = 2 * (std::size_t)b.nx + 1, nfirst = 2 * (std::size_t)b.lo * nn;
            checkHip(hipMemcpy(b.d[Ua], dy.u.data() + nfirst, NN * sizeof(double), hipMemcpyHostToDevice), "upload u");
            checkHip(hipMemcpy(b.d[Va], dy.v.data() + nfirst, NN * sizeof(double), hipMemcpyHostToDevice), "upload v");
            std::vector<double> t((std::size_t)TS, 0.);
            const std::vector<double>* comps[3] = { &dy.s11, &dy.s12, &dy.s22 };
            const int dst[3] = { S11a, S12a, S22a };
            for (int k = 0; k < 3; ++k) {
                tileRows(comps[k]->data(), NG, (std::size_t)b.lo, b.nx, b.ny, t);
                checkHip(hipMemcpy(b.d[dst[k]], t.data(), (std::size_t)TS * sizeof(double), hipMemcpyHostToDevice), "upload stress");
            }
        }
        // analytic box-test forcing, evaluated on the device (ocean once, wind at the current model time every step)
        check(nsdg_boxtest_forcing(b.ctx, L, m_time, b.d[UA], b.d[VA], b.d[UO], b.d[VO]), "nsdg_boxtest_forcing");
        if (thermo) {
            const std::vector<double>* planes[NCOL] = { &f.hsnow, &f.tice, &f.sst, &f.sss, &f.tair, &f.tdew, &f.slp, &f.qsw, &f.qlw, &f.mld,
                &f.snowfall, &f.wind, &f.newice };
            for (int c = 0; c < NCOL; ++c)
                checkHip(hipMemcpy(b.col(c), planes[c]->data() + first, N * sizeof(double), hipMemcpyHostToDevice), "upload column fields");
        }
        // driver plans
        nsdg_rb_mevp_desc m;
        std::memset(&m, 0, sizeof m);
        m.nx = b.nx, m.ny = b.ny, m.j0 = b.j0, m.j1 = b.j1, m.depth_below = b.depthBelow, m.depth_above = b.depthAbove;
        m.rank_below = b.peerBelow, m.rank_above = b.peerAbove;
        m.nsub = nsub, m.overlap = overlap, m.use_graph = graph;
        m.s11[0] = b.d[S11a], m.s12[0] = b.d[S12a], m.s22[0] = b.d[S22a], m.u[0] = b.d[Ua], m.v[0] = b.d[Va];
        m.s11[1] = b.d[S11b], m.s12[1] = b.d[S12b], m.s22[1] = b.d[S22b], m.u[1] = b.d[Ub], m.v[1] = b.d[Vb];
        m.packed = b.d[PACKED], m.pg = b.d[PG];
        check(nsdg_rb_mevp_create(b.ctx, &m, &b.mevp), "nsdg_rb_mevp_create");
        nsdg_rb_transport_desc t;
        std::memset(&t, 0, sizeof t);
        t.nx = b.nx, t.ny = b.ny, t.j0 = b.j0, t.j1 = b.j1, t.depth_below = b.depthBelow, t.depth_above = b.depthAbove;
        t.rank_below = b.peerBelow, t.rank_above = b.peerAbove;
        t.order = 2, t.nfields = 2;
        t.phi[0] = b.d[H0], t.phi[1] = b.d[A0], t.t1[0] = b.d[H1], t.t1[1] = b.d[A1], t.t2[0] = b.d[T2H], t.t2[1] = b.d[T2A];
        t.vx_dg = b.d[VXDG], t.vy_dg = b.d[VYDG], t.un_x = b.d[UNX], t.un_y = b.d[UNY];
        // the closure travels with the plan (its own bounds, not the context's): mean thickness H >= 0; concentration in [0, 1] with the
        // cell mean capped at 1
        t.own_bounds = 1, t.nbounds = closure ? 2 : 0;
        t.bounds[0] = nsdg_field_bounds { 0., HUGE_VAL, 0, 0 }, t.bounds[1] = nsdg_field_bounds { 0., 1., 1, 0 };
        check(nsdg_rb_transport_create(b.ctx, &t, &b.transport), "nsdg_rb_transport_create");
        b.par = b.tpar = 0;
    });
}

void DynamicsStep::iterate(const Iterator::Duration& dtSeconds)
{
    ScopedTimer timer("iterate");
    if (m_blocks.empty())
        start(0);
    const double dt = dtSeconds;
    nsdg_mevp_params p;
    nsdg_mevp_default_params(&p);
    const SubcycleChoice sc = subcycleChoice(std::min(L / nxf, L / nyf), dt);
    p.alpha = sc.alpha, p.beta = sc.beta, p.delta_min = sc.deltaMin, p.aevp_c = sc.aevpC, p.aevp_alpha_min = sc.aevpAlphaMin;
    p.min_conc = closure ? minConc : 0.;
    p.min_thick = closure ? minThick : 0.;
    const double t = m_time;
    const int kind = forcing == "winter" ? NSDG_FORCING_WINTER : NSDG_FORCING_DUMMY;
    forEachBlock([&](DynamicsBlock& b) {
        checkHip(hipSetDevice(b.device), "hipSetDevice");
        nsdg_ctx* ctx = b.ctx;
        check(nsdg_mevp_params_set(ctx, &p), "nsdg_mevp_params_set");
        check(nsdg_boxtest_forcing(ctx, L, t, b.d[UA], b.d[VA], nullptr, nullptr), "nsdg_boxtest_forcing"); // the cyclone moves
        if (thermo) {
            if (forcing != "host") { // DummyExternalData's replacement, and the wind speed the reference never sets
                check(nsdg_column_forcing(ctx, kind, t, b.col(C_TAIR), b.col(C_TDEW), b.col(C_SLP), b.col(C_QSW), b.col(C_QLW), b.col(C_MLD),
                          b.col(C_SNOWFALL)),
                    "nsdg_column_forcing");
                check(nsdg_column_wind(ctx, b.d[UA], b.d[VA], b.col(C_WIND)), "nsdg_column_wind");
            }
            // the column physics needs no exchange: it runs on the ghost rows too, redundantly
            check(nsdg_column_step(ctx, b.N, dt, b.curH(), b.curA(), b.col(C_HSNOW), b.col(C_TICE), b.col(C_SST), b.col(C_SSS), b.col(C_TAIR),
                      b.col(C_TDEW), b.col(C_SLP), b.col(C_QSW), b.col(C_QLW), b.col(C_MLD), b.col(C_SNOWFALL), b.col(C_WIND), b.col(C_NEWICE),
                      nullptr),
                "nsdg_column_step");
        }
        check(nsdg_ice_strength(ctx, 0, b.ny, b.curH(), b.curA(), b.d[PG]), "nsdg_ice_strength");
        check(nsdg_mevp_prepare(ctx, dt, b.curH(), b.curA(), b.d[UA], b.d[VA], b.d[UO], b.d[VO], b.curU(), b.curV(), b.d[PACKED]), "nsdg_mevp_prepare");
        int32_t out = 0;
        check(nsdg_rb_mevp_run(ctx, b.mevp, b.par, &out), "nsdg_rb_mevp_run");
        b.par = out;
        check(nsdg_prepare_advection(ctx, 2, b.curU(), b.curV(), b.d[VXDG], b.d[VYDG], b.d[UNX], b.d[UNY]), "nsdg_prepare_advection");
        check(nsdg_rb_transport_run(ctx, b.transport, dt, b.tpar, &out), "nsdg_rb_transport_run");
        b.tpar = out;
    });
    ++m_steps;
    m_time += dt;
}

void DynamicsStep::stop(const Iterator::TimePoint&)
{
    ScopedTimer timer("stop (download)");
    if (m_blocks.empty())
        return;
    FieldStore& f = pStructure->fields();
    std::vector<double> umax(m_blocks.size(), 0.);
    bool finite = true;
    std::size_t idx = 0;
    for (auto& bp : m_blocks) { // sequential: the blocks write disjoint row ranges of the host structure
        DynamicsBlock& b = *bp;
        checkHip(hipSetDevice(b.device), "hipSetDevice");
        check(nsdg_ctx_synchronize(b.ctx), "DynamicsStep::stop");
        uint32_t gaveUp = 0; // bounded waits of the mEVP pipeline that hit their bound: the fields would be wrong
        check(nsdg_mevp_pipeline_health(b.ctx, &gaveUp), "nsdg_mevp_pipeline_health");
        if (gaveUp)
            throw std::runtime_error("DynamicsStep: " + std::to_string(gaveUp) + " wait(s) of the mEVP pipeline gave up: the fields are not to be trusted");
        const std::size_t first = (std::size_t)b.r0 * b.nx, count = (std::size_t)(b.r1 - b.r0) * b.nx, skip = (std::size_t)b.j0 * b.nx;
        checkHip(hipMemcpy(f.hice.data() + first, b.curH() + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download H");
        checkHip(hipMemcpy(f.cice.data() + first, b.curA() + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download A");
        if (thermo) {
            checkHip(hipMemcpy(f.hsnow.data() + first, b.col(C_HSNOW) + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download hsnow");
            checkHip(hipMemcpy(f.tice.data() + first, b.col(C_TICE) + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download tice");
            checkHip(hipMemcpy(f.newice.data() + first, b.col(C_NEWICE) + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download newice");
        }
        // owned node rows: [2 j0, 2 j1) plus the top boundary row on the last block
        const long nn = 2L * b.nx + 1;
        const long rows = 2L * (b.j1 - b.j0) + (b.peerAbove < 0 ? 1 : 0);
        // the rest of the state a restart needs (FieldStore::dyn): higher DG2 coefficients, velocity, stress of the owned rows
        DynamicsState& dy = f.dyn;
        const std::size_t NG = (std::size_t)nxf * nyf;
        if (dy.hdg.size() != 5 * NG)
            dy.resize((std::size_t)nyf, (std::size_t)nxf);
        for (int c = 1; c < 6; ++c) {
            checkHip(hipMemcpy(dy.hdg.data() + (std::size_t)(c - 1) * NG + first, b.curH() + (long)c * b.N + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download H (DG)");
            checkHip(hipMemcpy(dy.adg.data() + (std::size_t)(c - 1) * NG + first, b.curA() + (long)c * b.N + skip, count * sizeof(double), hipMemcpyDeviceToHost), "download A (DG)");
        }
        checkHip(hipMemcpy(dy.u.data() + 2L * b.r0 * nn, b.curU() + 2L * b.j0 * nn, rows * nn * sizeof(double), hipMemcpyDeviceToHost), "download u");
        checkHip(hipMemcpy(dy.v.data() + 2L * b.r0 * nn, b.curV() + 2L * b.j0 * nn, rows * nn * sizeof(double), hipMemcpyDeviceToHost), "download v");
        {
            const std::size_t TS = (std::size_t)nsdg_tiled_len(b.nx, b.ny, 8);
            std::vector<double> t(TS);
            std::vector<double>* comps[3] = { &dy.s11, &dy.s12, &dy.s22 };
            const int src[3] = { b.par == 0 ? S11a : S11b, b.par == 0 ? S12a : S12b, b.par == 0 ? S22a : S22b };
            for (int k = 0; k < 3; ++k) {
                checkHip(hipMemcpy(t.data(), b.d[src[k]], TS * sizeof(double), hipMemcpyDeviceToHost), "download stress");
                untileRows(t, b.nx, b.j0, b.j1 - b.j0, comps[k]->data(), NG, (std::size_t)b.r0);
            }
        }
        for (long k = 2L * b.r0 * nn; k < (2L * b.r0 + rows) * nn; ++k) {
            umax[idx] = std::max(umax[idx], std::max(std::fabs(dy.u[k]), std::fabs(dy.v[k])));
            finite = finite && std::isfinite(dy.u[k]) && std::isfinite(dy.v[k]);
        }
        ++idx;
    }
    f.dyn.present = true;
    m_umax = *std::max_element(umax.begin(), umax.end());
    m_sumH = m_sumA = 0;
    for (auto& bp : m_blocks)
        for (long e = (long)bp->r0 * nxf; e < (long)bp->r1 * nxf; ++e) {
            m_sumH += f.hice[e];
            m_sumA += f.cice[e];
        }
    // A run that has left the physical range (profiles/r04_soak_divergence_cause.md) must fail loudly: the
    // caller gets an exception (non-zero exit of nextsim_amd) and -- because writeRestartFile() comes through here first -- no
    // restart file full of NaN is written.  (The reference never checks its fields; it also has no dynamics to go unstable.)
    if (!finite || !std::isfinite(m_sumH) || !std::isfinite(m_sumA))
        throw std::runtime_error("DynamicsStep: the run left the physical range (non-finite velocity, thickness or concentration after "
            + std::to_string(m_steps) + " steps); no restart file is written");
}

NSDG_REGISTER_MODULE(IModelStep, DynamicsStep, "Nextsim::IModelStep", "Nextsim::DynamicsStep");

void DynamicsStep::writeRestartFile(const std::string& filePath)
{
    if (!pStructure)
        throw std::logic_error("DynamicsStep::writeRestartFile: setInitialData() was not called");
    if (m_world <= 1) {
        stop(0);
        pStructure->dump(filePath);
        return;
    }
    // A rank holds only its own rows up to date: they travel to rank 0, which writes the ONE restart file (every row of it
    // current) and ACKNOWLEDGES it: no rank returns before rank 0 has written the file, and if anything fails -- a rank's
    // fields are non-finite (stop() throws), a rank has died, the write fails -- EVERY rank throws, so that a run without a
    // restart file never exits with status 0 on some of its ranks.  The waits are bounded by the communicator's deadline
    // (NSDG_COMM_TIMEOUT_S, the same variable the library and bench.py read; 0 = for ever; default 120 s here).
    const char* tv = std::getenv("NSDG_COMM_TIMEOUT_S");
    const int timeout = (tv && *tv) ? std::atoi(tv) : 120;
    const RankEnvironment env = RankEnvironment::fromEnv();
    std::exception_ptr own;
    try {
        stop(0);
    } catch (...) {
        own = std::current_exception();
    }
    if (own && m_rank != 0) {
        reportFailureToRankZero(env, std::min(timeout > 0 ? timeout : 86400, 30));
        std::rethrow_exception(own);
    }
    FieldStore& f = pStructure->fields();
    int r0, r1;
    splitRows(nyf, m_world, m_rank, r0, r1);
    const std::vector<double> mine = (m_rank == 0 || own) ? std::vector<double>() : packRows(f, thermo, nxf, r0, r1);
    gatherToRankZero(
        env, mine.data(), mine.size() * sizeof(double),
        [&](int rank, const char* data, std::size_t bytes) {
            if (own)
                return; // rank 0 itself failed: the rows are received and dropped, the senders are told below
            int a, b;
            splitRows(nyf, m_world, rank, a, b);
            placeRows(f, thermo, nxf, a, b, reinterpret_cast<const double*>(data), bytes / sizeof(double));
        },
        timeout,
        [&]() {
            if (own)
                std::rethrow_exception(own);
            pStructure->dump(filePath);
        });
}

std::vector<std::vector<double>*> DynamicsStep::restartPlanes(FieldStore& f, bool thermodynamics)
{
    std::vector<std::vector<double>*> planes = { &f.hice, &f.cice };
    if (thermodynamics)
        for (auto* p : { &f.hsnow, &f.tice, &f.newice })
            planes.push_back(p);
    return planes;
}

namespace {
// the element planes of the dynamics state (FieldStore::dyn): 5 + 5 + 3 x 8 planes of n values
std::vector<double*> dynamicsPlanes(FieldStore& f)
{
    std::vector<double*> p;
    DynamicsState& d = f.dyn;
    for (auto* v : { &d.hdg, &d.adg })
        for (int c = 0; c < 5; ++c)
            p.push_back(v->data() + (std::size_t)c * f.n);
    for (auto* v : { &d.s11, &d.s12, &d.s22 })
        for (int c = 0; c < 8; ++c)
            p.push_back(v->data() + (std::size_t)c * f.n);
    return p;
}
// node rows a block of element rows [r0, r1) owns: [2 r0, 2 r1), and the top boundary row on the last block
std::size_t ownedNodeRows(int r0, int r1, int ny) { return 2 * (std::size_t)(r1 - r0) + (r1 == ny ? 1 : 0); }
} // namespace

std::vector<double> DynamicsStep::packRows(FieldStore& f, bool thermodynamics, int nx, int r0, int r1)
{
    std::vector<double> out;
    const std::size_t first = (std::size_t)r0 * nx, count = (std::size_t)(r1 - r0) * nx;
    for (auto* p : restartPlanes(f, thermodynamics))
        out.insert(out.end(), p->begin() + first, p->begin() + first + count);
    if (f.dyn.present) { // the state of the dynamics travels with the rows: element planes, then the owned node rows of u and v
        for (double* p : dynamicsPlanes(f))
            out.insert(out.end(), p + first, p + first + count);
        const std::size_t nn = 2 * (std::size_t)nx + 1, nfirst = 2 * (std::size_t)r0 * nn, ncount = ownedNodeRows(r0, r1, (int)(f.n / nx)) * nn;
        for (auto* v : { &f.dyn.u, &f.dyn.v })
            out.insert(out.end(), v->begin() + nfirst, v->begin() + nfirst + ncount);
    }
    return out;
}

void DynamicsStep::placeRows(FieldStore& f, bool thermodynamics, int nx, int r0, int r1, const double* data, std::size_t count)
{
    const auto planes = restartPlanes(f, thermodynamics);
    const std::size_t first = (std::size_t)r0 * nx, rows = (std::size_t)(r1 - r0) * nx;
    const std::size_t nn = 2 * (std::size_t)nx + 1, ny = f.n / nx, nrows = ownedNodeRows(r0, r1, (int)ny) * nn;
    const std::size_t plain = rows * planes.size(), full = plain + rows * (5 + 5 + 24) + 2 * nrows;
    if (count != plain && count != full)
        throw std::runtime_error("DynamicsStep: a rank delivered " + std::to_string(count) + " values for its rows, expected "
            + std::to_string(plain) + " (or " + std::to_string(full) + " with the state of the dynamics)");
    for (std::size_t k = 0; k < planes.size(); ++k)
        std::copy(data + k * rows, data + (k + 1) * rows, planes[k]->begin() + first);
    if (count == full) {
        if (f.dyn.hdg.size() != 5 * f.n)
            f.dyn.resize(ny, (std::size_t)nx);
        const double* p = data + plain;
        for (double* dst : dynamicsPlanes(f)) {
            std::copy(p, p + rows, dst + first);
            p += rows;
        }
        for (auto* v : { &f.dyn.u, &f.dyn.v }) {
            std::copy(p, p + nrows, v->begin() + 2 * (std::size_t)r0 * nn);
            p += nrows;
        }
        f.dyn.present = true;
    }
}

} // namespace Nextsim
