// PhysicsModules.hpp -- the column-physics plugin interfaces and implementations, with the
// reference's fully qualified names (core/src/modules/modules.json:1-15,
// physics/src/modules/modules.json:1-34) so that existing [Modules] configuration lines keep working.
//
// Re-design for a batched GPU step: in the reference every element calls the selected plugins
// through virtual functions (~10 virtual calls per element-step).  Here a plugin is
// a DESCRIPTOR: configure() reads its keys exactly as the reference class does, describe() writes
// its choice and parameters into the POD nsdg_column_params that crosses the C ABI, and the HIP
// kernel branches wave-uniformly on it.  The scalar formulae stay available on the host
// (operator() / albedo()) for diagnostics and tests.
#pragma once
#include "../../../include/nsdg.h"
#include "Configured.hpp"

namespace Nextsim {

class IFreezingPoint {
public:
    virtual ~IFreezingPoint() = default;
    virtual double operator()(double sss) const = 0; // core/src/modules/include/IFreezingPoint.hpp:14-28
    virtual void describe(nsdg_column_params& p) const = 0;
};
class LinearFreezing : public IFreezingPoint { // LinearFreezing.hpp:30-34
public:
    double operator()(double sss) const override { return -0.055 * sss; }
    void describe(nsdg_column_params& p) const override { p.freezing_kind = NSDG_FREEZING_LINEAR; }
};
class UnescoFreezing : public IFreezingPoint { // UnescoFreezing.hpp:28-38
public:
    double operator()(double sss) const override;
    void describe(nsdg_column_params& p) const override { p.freezing_kind = NSDG_FREEZING_UNESCO; }
};

class IIceAlbedo {
public:
    virtual ~IIceAlbedo() = default;
    virtual double albedo(double temperature, double snowThickness) = 0; // IIceAlbedo.hpp
    virtual void describe(nsdg_column_params& p) const = 0;
};
class SMUIceAlbedo : public IIceAlbedo { // SMUIceAlbedo.cpp:19-26
public:
    double albedo(double temperature, double snowThickness) override;
    void describe(nsdg_column_params& p) const override { p.albedo_kind = NSDG_ALBEDO_SMU; }
};
class SMU2IceAlbedo : public IIceAlbedo { // SMU2IceAlbedo.cpp:21-29
public:
    double albedo(double temperature, double snowThickness) override;
    void describe(nsdg_column_params& p) const override { p.albedo_kind = NSDG_ALBEDO_SMU2; }
};
class CCSMIceAlbedo : public IIceAlbedo, public Configured<CCSMIceAlbedo> { // CCSMIceAlbedo.cpp:22-42
public:
    void configure() override;
    double albedo(double temperature, double snowThickness) override;
    void describe(nsdg_column_params& p) const override;
    static double iceAlbedo, snowAlbedo;
};

class IIceOceanHeatFlux {
public:
    virtual ~IIceOceanHeatFlux() = default;
    virtual void describe(nsdg_column_params& p) const = 0;
};
class BasicIceOceanHeatFlux : public IIceOceanHeatFlux { // BasicIceOceanHeatFlux.cpp:16-25 (no parameters)
public:
    void describe(nsdg_column_params&) const override { }
};

class IConcentrationModel {
public:
    virtual ~IConcentrationModel() = default;
    virtual void describe(nsdg_column_params& p) const = 0;
};
class HiblerConcentration : public IConcentrationModel, public Configured<HiblerConcentration> { // HiblerConcentration.cpp:17-47
public:
    enum { H0_KEY, PHIM_KEY };
    void configure() override;
    void describe(nsdg_column_params& p) const override;
    static double h0, phiM;
};

class IThermodynamics {
public:
    virtual ~IThermodynamics() = default;
    virtual void describe(nsdg_column_params& p) const = 0;
};
class ThermoIce0 : public IThermodynamics, public Configured<ThermoIce0> { // ThermoIce0.cpp:19-32
public:
    enum { KS_KEY, FLOODING_KEY };
    void configure() override;
    void describe(nsdg_column_params& p) const override;
    static double k_s;
    static bool doFlooding;
};

class IPhysics1d {
public:
    virtual ~IPhysics1d() = default;
    //! Everything the device kernel needs to reproduce this implementation's column step.
    virtual void describe(nsdg_column_params& p) const = 0;
};
class NextsimPhysics : public IPhysics1d, public Configured<NextsimPhysics> { // NextsimPhysics.cpp:26-83
public:
    enum { DRAGOCEANQ_KEY, DRAGOCEANT_KEY, DRAGICET_KEY, OCEANALBEDO_KEY, I0_KEY, MINC_KEY, MINH_KEY };
    void configure() override; //!< configures itself and the four helper plugins it uses
    void describe(nsdg_column_params& p) const override;
    static double minimumIceConcentration() { return minc; }
    static double minimumIceThickness() { return minh; }
    static double i0() { return m_I0; }

private:
    static double dragOcean_q, dragOcean_t, dragIce_t, m_oceanAlbedo, m_I0, minc, minh;
};

} // namespace Nextsim
