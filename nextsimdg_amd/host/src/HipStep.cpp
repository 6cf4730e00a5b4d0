#include "HipStep.hpp"

#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>

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
// plane order inside the device block
enum { P_HICE, P_CICE, P_HSNOW, P_TICE0, P_SST, P_SSS, P_TAIR, P_TDEW, P_SLP, P_QSW, P_QLW, P_MLD, P_SNOWFALL, P_WIND, P_NEWICE, NPLANES };
} // namespace

HipStep::HipStep() = default;

HipStep::~HipStep() { release(); }

void HipStep::release()
{
    if (d_block)
        (void)hipFree(d_block);
    d_block = nullptr;
    if (ctx)
        nsdg_ctx_destroy(ctx);
    ctx = nullptr;
    resident = false;
}

void HipStep::init()
{
    if (!ctx)
        check(nsdg_ctx_create(0, nullptr, &ctx), "HipStep::init");
    // the selected plugins describe themselves into the parameter block that crosses the ABI
    nsdg_column_params p;
    IPhysics1d& phys = ModuleLoader::getLoader().getImplementation<IPhysics1d>();
    tryConfigure(phys);
    tryConfigure(ModuleLoader::getLoader().getImplementation<IFreezingPoint>());
    phys.describe(p);
    check(nsdg_column_params_set(ctx, &p), "HipStep::init");
}

void HipStep::upload()
{
    if (!pStructure)
        throw std::logic_error("HipStep: setInitialData() was not called");
    if (!ctx)
        init();
    FieldStore& f = pStructure->fields();
    if (f.n != n || !d_block) {
        if (d_block)
            (void)hipFree(d_block);
        n = f.n;
        checkHip(hipMalloc(reinterpret_cast<void**>(&d_block), NPLANES * n * sizeof(double)), "HipStep: hipMalloc");
    }
    const std::vector<double>* planes[NPLANES] = { &f.hice, &f.cice, &f.hsnow, nullptr, &f.sst, &f.sss, &f.tair, &f.tdew, &f.slp, &f.qsw,
        &f.qlw, &f.mld, &f.snowfall, &f.wind, &f.newice };
    for (int k = 0; k < NPLANES; ++k) {
        const double* src = (k == P_TICE0) ? f.tice.data() : planes[k]->data(); // layer 0 is the first tice plane
        checkHip(hipMemcpy(d_block + k * n, src, n * sizeof(double), hipMemcpyHostToDevice), "HipStep: upload");
    }
    resident = true;
}

void HipStep::start(const Iterator::TimePoint&) { upload(); }

void HipStep::iterate(const Iterator::Duration& dt)
{
    ScopedTimer timer("iterate");
    if (!resident)
        upload();
    double* b = d_block;
    check(nsdg_column_step(ctx, (int64_t)n, (double)dt, b + P_HICE * n, b + P_CICE * n, b + P_HSNOW * n, b + P_TICE0 * n, b + P_SST * n,
              b + P_SSS * n, b + P_TAIR * n, b + P_TDEW * n, b + P_SLP * n, b + P_QSW * n, b + P_QLW * n, b + P_MLD * n, b + P_SNOWFALL * n,
              b + P_WIND * n, b + P_NEWICE * n, nullptr),
        "HipStep::iterate");
    ++m_launches;
}

void HipStep::syncToHost()
{
    if (!resident)
        return;
    check(nsdg_ctx_synchronize(ctx), "HipStep::syncToHost");
    FieldStore& f = pStructure->fields();
    std::vector<double>* out[] = { &f.hice, &f.cice, &f.hsnow };
    const int plane[] = { P_HICE, P_CICE, P_HSNOW };
    for (int k = 0; k < 3; ++k)
        checkHip(hipMemcpy(out[k]->data(), d_block + plane[k] * n, n * sizeof(double), hipMemcpyDeviceToHost), "HipStep: download");
    checkHip(hipMemcpy(f.tice.data(), d_block + P_TICE0 * n, n * sizeof(double), hipMemcpyDeviceToHost), "HipStep: download");
    checkHip(hipMemcpy(f.newice.data(), d_block + P_NEWICE * n, n * sizeof(double), hipMemcpyDeviceToHost), "HipStep: download");
    // only TiceNew[0] is ever written by the physics; the other layers take the constructor value 0 on
    // write-back (PhysicsData.hpp:35,69 + PrognosticData.cpp:63-71)
    if (m_launches > 0)
        for (int l = 1; l < f.nLayers; ++l)
            std::fill(f.tice.begin() + (std::size_t)l * n, f.tice.begin() + (std::size_t)(l + 1) * n, 0.);
}

void HipStep::stop(const Iterator::TimePoint&) { syncToHost(); }

void HipStep::writeRestartFile(const std::string& filePath)
{
    if (!pStructure)
        throw std::logic_error("HipStep::writeRestartFile: setInitialData() was not called");
    syncToHost();
    pStructure->dump(filePath);
}

} // namespace Nextsim

namespace Nextsim {
// the model step is a plugin too; HipStep first = default
NSDG_REGISTER_MODULE(IModelStep, HipStep, "Nextsim::IModelStep", "Nextsim::HipStep");
}
