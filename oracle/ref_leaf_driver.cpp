// ref_leaf_driver.cpp -- TEST INFRASTRUCTURE.
//
// Thin extern "C" driver around the only pieces of the reference that compile from their own
// sources with g++ alone (no Boost, no generated .ipp): the two header-only freezing-point models
// and the constants header.  The reference headers are #included BY PATH from /root/reference at
// build time (oracle/Makefile passes -I/root/reference/core/src ...); nothing from the reference is
// copied into this repository, and the resulting library goes to oracle/_ref/ (git-ignored).
//
// Everything else on the column-physics path (NextsimPhysics.cpp, ThermoIce0.cpp, the albedo and
// concentration modules) includes Configured.hpp -> <boost/program_options.hpp> and
// ModuleLoader.cpp -> generated moduleLoader*.ipp, neither of which exists in this image, so it is
// treated as unbuildable here.
#include "modules/include/LinearFreezing.hpp" // core/src/modules/include/LinearFreezing.hpp:30-34
#include "modules/include/UnescoFreezing.hpp" // core/src/modules/include/UnescoFreezing.hpp:28-38
#include "include/constants.hpp" // core/src/include/constants.hpp:11-144

extern "C" {
double ref_freezing_linear(double sss)
{
    Nextsim::LinearFreezing f;
    const Nextsim::IFreezingPoint& i = f;
    return i(sss);
}
double ref_freezing_unesco(double sss)
{
    Nextsim::UnescoFreezing f;
    const Nextsim::IFreezingPoint& i = f;
    return i(sss);
}
// constants by index, so the oracle's literals can be compared one by one
double ref_constant(int k)
{
    switch (k) {
    case 0: return PhysicalConstants::sigma;
    case 1: return Ice::cp;
    case 2: return Ice::epsilon;
    case 3: return Ice::kappa;
    case 4: return Ice::Lf;
    case 5: return Ice::rho;
    case 6: return Ice::rhoSnow;
    case 7: return Ice::s;
    case 8: return Ice::Tm;
    case 9: return Air::cp;
    case 10: return Air::Ra;
    case 11: return Vapour::cp;
    case 12: return Vapour::Lv0;
    case 13: return Vapour::Ra;
    case 14: return Water::cp;
    case 15: return Water::Lf;
    case 16: return Water::mu;
    case 17: return Water::rhoOcean;
    case 18: return Water::Tf;
    case 19: return Nextsim::kelvin(0.);
    default: return 0. / 0.;
    }
}
}
