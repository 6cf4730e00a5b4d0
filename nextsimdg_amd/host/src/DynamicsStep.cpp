#include "DynamicsStep.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "../../../include/nsdg.h"
#include "ModuleLoader.hpp"
#include "Timer.hpp"
#include "PhysicsModules.hpp"

namespace Nextsim {

namespace {
void check(int rc, const char* what)
{
    if (rc != NSDG_OK)
        throw std::runtime_error(std::string(what) + ": " + nsdg_last_error());
}
void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
// layout of the device block
enum Arr { H, A, T1, T2, S11, S12, S22, PG, VXDG, VYDG, UNX, UNY, U, V, U0, V0, UA, VA, TAX, TAY, UO, VO, CGH, CGA, SCRATCH, COL, NARR };
} // namespace

template <>
const std::map<int, std::string> Configured<DynamicsStep>::keyMap = { { 0, "dynamics.domain_size" }, { 1, "dynamics.nsub" },
    { 2, "dynamics.alpha" }, { 3, "dynamics.beta" }, { 4, "dynamics.thermodynamics" } };

DynamicsStep::DynamicsStep() = default;
DynamicsStep::~DynamicsStep() { release(); }

void DynamicsStep::release()
{
    if (d_block)
        (void)hipFree(d_block);
    d_block = nullptr;
    if (ctx)
        nsdg_ctx_destroy(ctx);
    ctx = nullptr;
}

double DynamicsStep::stableAlpha(double h, double dt)
{ // alpha*beta >= pi^2 zeta_max dt / (m h^2), zeta_max = P* H / (2 Delta_min); same rule as synthetic.BoxTest.stable_alpha
    const double pstar = 27.5e3, dmin = 2e-9, rho = 900., hice = 0.3;
    const double zeta = pstar * hice / (2. * dmin);
    const double pi = 3.14159265358979323846;
    return std::max(1500., 1.2 * std::sqrt(pi * pi * zeta * dt / (rho * hice * h * h)));
}

void DynamicsStep::configure()
{
    L = getConfiguration(keyMap.at(0), 512e3);
    nsub = getConfiguration(keyMap.at(1), 120);
    alpha = getConfiguration(keyMap.at(2), 0.);
    beta = getConfiguration(keyMap.at(3), 0.);
    thermo = getConfiguration(keyMap.at(4), false);
}

void DynamicsStep::init()
{
    configure();
    if (!ctx)
        check(nsdg_ctx_create(0, nullptr, &ctx), "DynamicsStep::init");
    if (thermo) {
        nsdg_column_params p;
        IPhysics1d& phys = ModuleLoader::getLoader().getImplementation<IPhysics1d>();
        tryConfigure(phys);
        tryConfigure(ModuleLoader::getLoader().getImplementation<IFreezingPoint>());
        phys.describe(p);
        check(nsdg_column_params_set(ctx, &p), "DynamicsStep::init");
    }
}

void DynamicsStep::start(const Iterator::TimePoint& startTime)
{
    ScopedTimer timer("start (upload)");
    if (!pStructure)
        throw std::logic_error("DynamicsStep: setInitialData() was not called");
    if (!ctx)
        init();
    FieldStore& f = pStructure->fields();
    nxf = pStructure->ny(); // fast dimension of the x-major index i*ny + j
    nyf = pStructure->nx();
    N = (long)nxf * nyf;
    NN = (long)(2 * nxf + 1) * (2 * nyf + 1);
    // stress and ice strength are private to the sub-cycle and use the ABI's tiled layout
    const long TS = nsdg_tiled_len(nxf, nyf, 8), TP = nsdg_tiled_len(nxf, nyf, 9);
    const long sizes[NARR] = { 6 * N, 6 * N, 12 * N, 12 * N, TS, TS, TS, TP, 6 * N, 6 * N, 3L * (nxf + 1) * nyf, 3L * nxf * (nyf + 1), NN, NN,
        NN, NN, NN, NN, NN, NN, NN, NN, NN, NN, 10 * NN + 3 * TS, 13 * N };
    long total = 0;
    for (long s : sizes)
        total += s + 2; // keep every sub-array 16-byte aligned
    if (d_block)
        (void)hipFree(d_block);
    checkHip(hipMalloc(reinterpret_cast<void**>(&d_block), total * sizeof(double)), "DynamicsStep: hipMalloc");
    checkHip(hipMemset(d_block, 0, total * sizeof(double)), "DynamicsStep: hipMemset");
    d.assign(NARR, nullptr);
    long off = 0;
    for (int k = 0; k < NARR; ++k) {
        d[k] = d_block + off;
        off += (sizes[k] + 1) & ~1L;
    }
    const double hx = L / nxf, hy = L / nyf;
    check(nsdg_grid_set(ctx, nxf, nyf, hx, hy), "DynamicsStep::start");
    // cell means -> DG coefficient 0
    checkHip(hipMemcpy(d[H], f.hice.data(), N * sizeof(double), hipMemcpyHostToDevice), "upload H");
    checkHip(hipMemcpy(d[A], f.cice.data(), N * sizeof(double), hipMemcpyHostToDevice), "upload A");
    // analytic box-test forcing, evaluated on the device (ocean once, wind at the current model time every step)
    check(nsdg_boxtest_forcing(ctx, L, (double)startTime, d[UA], d[VA], d[UO], d[VO]), "forcing");
    m_time = (double)startTime;
    if (thermo) { // column planes: hsnow, tice0, sst, sss, tair, tdew, slp, qsw, qlw, mld, snowfall, wind, newice
        const std::vector<double>* planes[13] = { &f.hsnow, nullptr, &f.sst, &f.sss, &f.tair, &f.tdew, &f.slp, &f.qsw, &f.qlw, &f.mld,
            &f.snowfall, &f.wind, &f.newice };
        for (int k = 0; k < 13; ++k)
            checkHip(hipMemcpy(d[COL] + k * N, k == 1 ? f.tice.data() : planes[k]->data(), N * sizeof(double), hipMemcpyHostToDevice),
                "upload column fields");
    }
}

void DynamicsStep::iterate(const Iterator::Duration& dtSeconds)
{
    ScopedTimer timer("iterate");
    if (!d_block)
        start(0);
    const double dt = dtSeconds;
    nsdg_mevp_params p;
    nsdg_mevp_default_params(&p);
    const double a = alpha > 0 ? alpha : stableAlpha(std::min(L / nxf, L / nyf), dt);
    p.alpha = a;
    p.beta = beta > 0 ? beta : a;
    check(nsdg_mevp_params_set(ctx, &p), "DynamicsStep::iterate");
    if (thermo) {
        double* c = d[COL];
        check(nsdg_column_step(ctx, N, dt, d[H], d[A], c, c + N, c + 2 * N, c + 3 * N, c + 4 * N, c + 5 * N, c + 6 * N, c + 7 * N, c + 8 * N,
                  c + 9 * N, c + 10 * N, c + 11 * N, c + 12 * N, nullptr),
            "column step");
    }
    check(nsdg_boxtest_forcing(ctx, L, m_time, d[UA], d[VA], nullptr, nullptr), "forcing"); // the cyclone moves
    check(nsdg_ice_strength(ctx, 0, nyf, d[H], d[A], d[PG]), "ice_strength");
    // nsdg_mevp_subcycle takes the unpacked nodal fields (general entry point); u0/v0 may alias u/v
    check(nsdg_dg_to_cg(ctx, 6, d[H], d[CGH]), "dg_to_cg");
    check(nsdg_dg_to_cg(ctx, 6, d[A], d[CGA]), "dg_to_cg");
    check(nsdg_wind_stress(ctx, NN, d[UA], d[VA], d[TAX], d[TAY]), "wind_stress");
    check(nsdg_mevp_subcycle(ctx, dt, nsub, d[S11], d[S12], d[S22], d[U], d[V], d[U], d[V], d[TAX], d[TAY], d[UO], d[VO], d[CGH], d[CGA],
              d[PG], d[SCRATCH]),
        "mevp_subcycle");
    check(nsdg_prepare_advection(ctx, 2, d[U], d[V], d[VXDG], d[VYDG], d[UNX], d[UNY]), "prepare_advection");
    double* fields[2] = { d[H], d[A] };
    check(nsdg_transport_step(ctx, 2, dt, 2, fields, d[VXDG], d[VYDG], d[UNX], d[UNY], d[T1]), "transport_step"); // T1,T2 contiguous: 24N scratch
    ++m_steps;
    m_time += dt;
}

void DynamicsStep::stop(const Iterator::TimePoint&)
{
    ScopedTimer timer("stop (download)");
    if (!d_block)
        return;
    check(nsdg_ctx_synchronize(ctx), "DynamicsStep::stop");
    FieldStore& f = pStructure->fields();
    checkHip(hipMemcpy(f.hice.data(), d[H], N * sizeof(double), hipMemcpyDeviceToHost), "download H");
    checkHip(hipMemcpy(f.cice.data(), d[A], N * sizeof(double), hipMemcpyDeviceToHost), "download A");
    if (thermo) {
        checkHip(hipMemcpy(f.hsnow.data(), d[COL], N * sizeof(double), hipMemcpyDeviceToHost), "download hsnow");
        checkHip(hipMemcpy(f.tice.data(), d[COL] + N, N * sizeof(double), hipMemcpyDeviceToHost), "download tice");
        checkHip(hipMemcpy(f.newice.data(), d[COL] + 12 * N, N * sizeof(double), hipMemcpyDeviceToHost), "download newice");
    }
    std::vector<double> u(NN);
    checkHip(hipMemcpy(u.data(), d[U], NN * sizeof(double), hipMemcpyDeviceToHost), "download u");
    m_umax = 0;
    for (double x : u)
        m_umax = std::max(m_umax, std::fabs(x));
    m_sumH = m_sumA = 0;
    for (long e = 0; e < N; ++e) {
        m_sumH += f.hice[e];
        m_sumA += f.cice[e];
    }
}

NSDG_REGISTER_MODULE(IModelStep, DynamicsStep, "Nextsim::IModelStep", "Nextsim::DynamicsStep");

void DynamicsStep::writeRestartFile(const std::string& filePath)
{
    if (!pStructure)
        throw std::logic_error("DynamicsStep::writeRestartFile: setInitialData() was not called");
    stop(0);
    pStructure->dump(filePath);
}

} // namespace Nextsim
