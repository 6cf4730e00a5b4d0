#include "PhysicsModules.hpp"

#include <cmath>

#include "ModuleLoader.hpp"

namespace Nextsim {

double UnescoFreezing::operator()(double sss) const
{
    return sss * (-0.0575 + 1.710523e-3 * std::sqrt(sss) - 2.154996e-4 * sss);
}

double SMUIceAlbedo::albedo(double, double snowThickness)
{
    return snowThickness > 0. ? 0.85 : 0.64 + 0.4 * (1 - 0.64) * NextsimPhysics::i0();
}
double SMU2IceAlbedo::albedo(double, double snowThickness)
{
    return snowThickness > 0. ? std::fmin(0.85, 0.64 + (0.85 - 0.64) * snowThickness / 0.2) : 0.64 + 0.4 * (1 - 0.64) * NextsimPhysics::i0();
}

double CCSMIceAlbedo::iceAlbedo = 0.538;
double CCSMIceAlbedo::snowAlbedo = 0.8256;
template <> const std::map<int, std::string> Configured<CCSMIceAlbedo>::keyMap = { { 0, "CCSMIceAlbedo.iceAlbedo" }, { 1, "CCSMIceAlbedo.snowAlbedo" } };
void CCSMIceAlbedo::configure()
{
    iceAlbedo = getConfiguration(keyMap.at(0), 0.538);
    snowAlbedo = getConfiguration(keyMap.at(1), 0.8256);
}
double CCSMIceAlbedo::albedo(double temperature, double snowThickness)
{
    const double ai = iceAlbedo - std::fmax(0., 0.075 * (temperature + 1.));
    const double as = snowAlbedo - std::fmax(0., 0.124 * (temperature + 1.));
    const double f = snowThickness / (snowThickness + 0.02);
    return f * as + (1 - f) * ai;
}
void CCSMIceAlbedo::describe(nsdg_column_params& p) const
{
    p.albedo_kind = NSDG_ALBEDO_CCSM;
    p.ccsm_ice_albedo = iceAlbedo;
    p.ccsm_snow_albedo = snowAlbedo;
}

double HiblerConcentration::h0 = 0.25;
double HiblerConcentration::phiM = 0.5;
template <> const std::map<int, std::string> Configured<HiblerConcentration>::keyMap = { { HiblerConcentration::H0_KEY, "Hibler.h0" }, { HiblerConcentration::PHIM_KEY, "Hibler.phiM" } };
void HiblerConcentration::configure()
{
    h0 = getConfiguration(keyMap.at(H0_KEY), 0.25);
    phiM = getConfiguration(keyMap.at(PHIM_KEY), 0.5);
}
void HiblerConcentration::describe(nsdg_column_params& p) const
{
    p.h0 = h0;
    p.phi_m = phiM;
}

double ThermoIce0::k_s = 0.3096;
bool ThermoIce0::doFlooding = true;
template <> const std::map<int, std::string> Configured<ThermoIce0>::keyMap = { { ThermoIce0::KS_KEY, "thermoice0.ks" }, { ThermoIce0::FLOODING_KEY, "thermoice0.flooding" } };
void ThermoIce0::configure()
{
    k_s = getConfiguration(keyMap.at(KS_KEY), 0.3096);
    doFlooding = getConfiguration(keyMap.at(FLOODING_KEY), true);
}
void ThermoIce0::describe(nsdg_column_params& p) const
{
    p.ks = k_s;
    p.flooding = doFlooding ? 1 : 0;
}

double NextsimPhysics::dragOcean_q = 1.5e-3, NextsimPhysics::dragOcean_t = 0.83e-3, NextsimPhysics::dragIce_t = 1.3e-3;
double NextsimPhysics::m_oceanAlbedo = 0.07, NextsimPhysics::m_I0 = 0.17, NextsimPhysics::minc = 1e-12, NextsimPhysics::minh = 0.01;
template <>
const std::map<int, std::string> Configured<NextsimPhysics>::keyMap = {
    { NextsimPhysics::DRAGOCEANQ_KEY, "nextsim_thermo.drag_ocean_q" },
    { NextsimPhysics::DRAGOCEANT_KEY, "nextsim_thermo.drag_ocean_t" },
    { NextsimPhysics::DRAGICET_KEY, "nextsim_thermo.drag_ice_t" },
    { NextsimPhysics::OCEANALBEDO_KEY, "nextsim_thermo.albedoW" },
    { NextsimPhysics::I0_KEY, "nextsim_thermo.I_0" },
    { NextsimPhysics::MINC_KEY, "nextsim_thermo.min_conc" },
    { NextsimPhysics::MINH_KEY, "nextsim_thermo.min_thick" },
};
void NextsimPhysics::configure()
{
    ModuleLoader& loader = ModuleLoader::getLoader();
    tryConfigure(loader.getImplementation<IIceOceanHeatFlux>());
    tryConfigure(loader.getImplementation<IIceAlbedo>());
    tryConfigure(loader.getImplementation<IThermodynamics>());
    tryConfigure(loader.getImplementation<IConcentrationModel>());
    dragOcean_q = getConfiguration(keyMap.at(DRAGOCEANQ_KEY), 1.5e-3);
    dragOcean_t = getConfiguration(keyMap.at(DRAGOCEANT_KEY), 0.83e-3);
    dragIce_t = getConfiguration(keyMap.at(DRAGICET_KEY), 1.3e-3);
    m_oceanAlbedo = getConfiguration(keyMap.at(OCEANALBEDO_KEY), 0.07);
    m_I0 = getConfiguration(keyMap.at(I0_KEY), 0.17);
    minc = getConfiguration(keyMap.at(MINC_KEY), 1e-12);
    minh = getConfiguration(keyMap.at(MINH_KEY), 0.01);
}
void NextsimPhysics::describe(nsdg_column_params& p) const
{
    nsdg_column_default_params(&p);
    p.drag_ocean_q = dragOcean_q;
    p.drag_ocean_t = dragOcean_t;
    p.drag_ice_t = dragIce_t;
    p.ocean_albedo = m_oceanAlbedo;
    p.i0 = m_I0;
    p.min_conc = minc;
    p.min_thick = minh;
    ModuleLoader& loader = ModuleLoader::getLoader();
    loader.getImplementation<IIceOceanHeatFlux>().describe(p);
    loader.getImplementation<IIceAlbedo>().describe(p);
    loader.getImplementation<IThermodynamics>().describe(p);
    loader.getImplementation<IConcentrationModel>().describe(p);
    loader.getImplementation<IFreezingPoint>().describe(p);
}

// registration order = the order of the reference's modules.json files (first = default)
NSDG_REGISTER_MODULE(IFreezingPoint, LinearFreezing, "Nextsim::IFreezingPoint", "Nextsim::LinearFreezing");
NSDG_REGISTER_MODULE(IFreezingPoint, UnescoFreezing, "Nextsim::IFreezingPoint", "Nextsim::UnescoFreezing");
NSDG_REGISTER_MODULE(IIceAlbedo, SMUIceAlbedo, "Nextsim::IIceAlbedo", "Nextsim::SMUIceAlbedo");
NSDG_REGISTER_MODULE(IIceAlbedo, SMU2IceAlbedo, "Nextsim::IIceAlbedo", "Nextsim::SMU2IceAlbedo");
NSDG_REGISTER_MODULE(IIceAlbedo, CCSMIceAlbedo, "Nextsim::IIceAlbedo", "Nextsim::CCSMIceAlbedo");
NSDG_REGISTER_MODULE(IIceOceanHeatFlux, BasicIceOceanHeatFlux, "Nextsim::IIceOceanHeatFlux", "Nextsim::BasicIceOceanHeatFlux");
NSDG_REGISTER_MODULE(IConcentrationModel, HiblerConcentration, "Nextsim::IConcentrationModel", "Nextsim::HiblerConcentration");
NSDG_REGISTER_MODULE(IThermodynamics, ThermoIce0, "Nextsim::IThermodynamics", "Nextsim::ThermoIce0");
NSDG_REGISTER_MODULE(IPhysics1d, NextsimPhysics, "Nextsim::IPhysics1d", "Nextsim::NextsimPhysics");

} // namespace Nextsim
