// host_tests.cpp -- tests of the C++ host layer.  The cases restate, against this repo's own
// implementation, the behaviour the reference's Catch2 tests pin (file:line given per case); the GPU
// cases (run with --gpu) drive HipStep / Model through the C ABI and compare with the reference's
// known answers.  Tiny self-contained harness: CHECK() records failures, exit code = #failures.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <cstring>
#include <functional>
#include <iostream>
#include <sstream>
#include <thread>
#include <typeinfo>

#include "Configurator.hpp"
#include "DynamicsStep.hpp"
#include "EnumWrapper.hpp"
#include "Hdf5Subset.hpp"
#include "Model.hpp"
#include "ModuleLoader.hpp"
#include "PhysicsModules.hpp"
#include "Rendezvous.hpp"
#include <atomic>
#include "Timer.hpp"

using namespace Nextsim;

static int failures = 0, checks = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        ++checks;                                                                \
        if (!(cond)) {                                                           \
            ++failures;                                                          \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);          \
        }                                                                        \
    } while (0)
#define CHECK_THROWS_AS(expr, Ex)                                                \
    do {                                                                         \
        ++checks;                                                                \
        bool ok_ = false;                                                        \
        try {                                                                    \
            expr;                                                                \
        } catch (const Ex&) {                                                    \
            ok_ = true;                                                          \
        } catch (...) {                                                          \
        }                                                                        \
        if (!ok_) {                                                              \
            ++failures;                                                          \
            std::printf("FAIL %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expr, #Ex); \
        }                                                                        \
    } while (0)
static bool approx(double got, double want, double eps) { return std::fabs(got - want) <= eps * std::fabs(want) + 1e-300; }

// fake backend of the reference's loader tests (core/test/moduleTestClasses.hpp:12-28)
class ITest {
public:
    virtual ~ITest() = default;
    virtual int operator()() = 0;
};
class Impl1 : public ITest {
public:
    int operator()() override { return 1; }
};
class Impl2 : public ITest {
public:
    int operator()() override { return 2; }
};
NSDG_REGISTER_MODULE(ITest, Impl1, "ITest", "Impl1");
NSDG_REGISTER_MODULE(ITest, Impl2, "ITest", "Impl2");

struct ArgV { // fake argv (core/test/ArgV.cpp:12-41)
    std::vector<std::string> s;
    std::vector<char*> p;
    ArgV(std::initializer_list<const char*> a)
    {
        for (auto x : a)
            s.emplace_back(x);
        for (auto& x : s)
            p.push_back(&x[0]);
        p.push_back(nullptr);
    }
    int argc() { return (int)s.size(); }
    char** operator()() { return p.data(); }
};
static void addConfig(const std::string& text) { Configurator::addStream(std::unique_ptr<std::istream>(new std::stringstream(text))); }

static void test_module_loader()
{ // core/test/ModuleLoader_test.cpp:14-26
    ModuleLoader& ldr = ModuleLoader::getLoader();
    CHECK(ldr.listModules().count("ITest") == 1);
    CHECK(ldr.listImplementations("ITest").size() == 2);
    ldr.setImplementation("ITest", "Impl1");
    Impl1 i1;
    CHECK(typeid(i1) == typeid(*(ldr.getInstance<ITest>())));
    CHECK(&ldr.getImplementation<ITest>() == &ldr.getImplementation<ITest>()); // one shared static instance
    CHECK(ldr.getInstance<ITest>().get() != ldr.getInstance<ITest>().get()); // fresh objects
    ldr.setImplementation("ITest", "Impl2");
    CHECK(ldr.getImplementation<ITest>()() == 2);
    CHECK_THROWS_AS(ldr.setImplementation("ITest", "Impl3"), std::invalid_argument); // ModuleLoader.cpp:16-21
    CHECK_THROWS_AS(ldr.listImplementations("NoSuchModule"), std::out_of_range); // ModuleLoader.hpp:51
    ldr.setAllDefaults(); // default = first listed (ModuleLoader.cpp:51-54)
    CHECK(ldr.getImplementation<ITest>()() == 1);
    // the real registry: names and order of the reference's modules.json files
    const auto& alb = ldr.listImplementations("Nextsim::IIceAlbedo");
    CHECK(alb.size() == 3 && alb.front() == "Nextsim::SMUIceAlbedo" && alb.back() == "Nextsim::CCSMIceAlbedo");
    CHECK(ldr.listImplementations("Nextsim::IFreezingPoint").front() == "Nextsim::LinearFreezing");
    for (const char* m : { "Nextsim::IStructure", "Nextsim::IIceOceanHeatFlux", "Nextsim::IConcentrationModel", "Nextsim::IThermodynamics", "Nextsim::IPhysics1d" })
        CHECK(ldr.listModules().count(m) == 1);
    ModuleLoader::VariablesMap vm = { { "Nextsim::IFreezingPoint", "Nextsim::UnescoFreezing" } };
    ldr.init(vm);
    CHECK(std::fabs(ldr.getImplementation<IFreezingPoint>()(32.) - (-1.751)) < 5e-3);
    ldr.setAllDefaults();
    CHECK(ldr.getImplementation<IFreezingPoint>()(32.) == -0.055 * 32.);
}

static void test_configured_module()
{ // core/test/ConfiguredModule_test.cpp:22-88
    Configurator::clear();
    ArgV argvee({ "cmtest", "--Modules.ITest=Impl1" });
    Configurator::setCommandLine(argvee.argc(), argvee());
    ConfiguredModule::parseConfigurator();
    CHECK(ModuleLoader::getLoader().getImplementation<ITest>()() == 1);

    Configurator::clear();
    addConfig("[Modules]\nITest = Impl2\n");
    ConfiguredModule::parseConfigurator();
    CHECK(ModuleLoader::getLoader().getImplementation<ITest>()() == 2);

    Configurator::clear();
    ModuleLoader::getLoader().setImplementation("ITest", "Impl2");
    addConfig("[Modules]\nITestNotReally = NotImpl2\n"); // unknown module: ignored, selection unchanged
    ConfiguredModule::parseConfigurator();
    CHECK(ModuleLoader::getLoader().getImplementation<ITest>()() == 2);

    Configurator::clear();
    addConfig("[Modules]\nITest = Graham\n");
    CHECK_THROWS_AS(ConfiguredModule::parseConfigurator(), std::domain_error);
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

class Config1 : public Configured<Config1> { // core/test/Configurator_test.cpp:23-117 style fixtures
public:
    int value = 0;
    std::string name;
    void configure() override
    {
        value = getConfiguration<int>("config.value", -1);
        name = getConfiguration<std::string>("config.name", "");
    }
};
class Config2 : public Configured<Config2> {
public:
    int value = 0;
    double weight = 0;
    void configure() override
    {
        value = getConfiguration<int>("config.value", -2);
        weight = getConfiguration<double>("data.weight", 1.5);
    }
};
class NotConfigured {
public:
    int x = 7;
};

static void test_configurator()
{ // core/test/Configurator_test.cpp:119-230
    Configurator::clear();
    Config1 c1;
    c1.configure();
    CHECK(c1.value == -1 && c1.name.empty()); // defaults when nothing is configured
    addConfig("[config]\nvalue = 42\nname = Zaphod\n\n# comment\n[data]\nweight = 2.5\n");
    tryConfigure(c1); // by reference
    CHECK(c1.value == 42 && c1.name == "Zaphod");
    Config2 c2;
    tryConfigure(&c2); // by pointer; two classes share config.value
    CHECK(c2.value == 42 && c2.weight == 2.5);
    NotConfigured nc;
    tryConfigure(nc); // silently ignored
    tryConfigure(&nc);
    CHECK(nc.x == 7);
    // two streams: the first one that defines a key wins (Configurator.hpp:23-26)
    addConfig("[config]\nvalue = 43\n[extra]\nthing = 9\n");
    c1.configure();
    CHECK(c1.value == 42);
    CHECK(Configured<Config1>::getConfiguration<int>("extra.thing", 0) == 9);
    // the command line overrides every stream
    ArgV a({ "prog", "--config.value=7", "--unrelated", "--data.weight", "0.125" });
    Configurator::setCommandLine(a.argc(), a());
    c2.configure();
    CHECK(c2.value == 7 && c2.weight == 0.125);
    Configurator::clear();
    c2.configure();
    CHECK(c2.value == -2 && c2.weight == 1.5);
    CHECK(Configured<Config1>::getConfiguration<bool>("a.flag", true) == true);
    addConfig("[a]\nflag = false\n");
    CHECK(Configured<Config1>::getConfiguration<bool>("a.flag", true) == false);
    Configurator::clear();
}

enum class Colour { red, green, blue }; // core/test/EnumWrapper_test.cpp:20-24

static void test_enum_wrapper()
{ // core/test/EnumWrapper_test.cpp:26-52: an enum-valued option read from a configuration stream
    Configurator::clear();
    const std::string targetColourString = "red";
    EnumWrap::EnumWrapper<Colour>::setMap({ { targetColourString, Colour::red }, { "green", Colour::green }, { "blue", Colour::blue } });
    const std::string colourName = "option.colour";
    addConfig("[option]\ncolour = " + targetColourString + "\n");
    typedef EnumWrap::EnumWrapper<Colour> Wrapped;
    Colour col = Configured<Config1>::getConfiguration<Wrapped>(colourName, Wrapped(Colour::blue));
    CHECK(col == Colour::red);
    // the other values, the default for an absent key, and the command line overriding the stream
    CHECK(Colour(Configured<Config1>::getConfiguration<Wrapped>("option.missing", Wrapped(Colour::blue))) == Colour::blue);
    ArgV a({ "prog", "--option.colour=green" });
    Configurator::setCommandLine(a.argc(), a());
    CHECK(Colour(Configured<Config1>::getConfiguration<Wrapped>(colourName, Wrapped(Colour::blue))) == Colour::green);
    Configurator::clear();
    // stream extraction and the call operator (EnumWrapper.hpp:68-76,99-111)
    Wrapped w;
    std::istringstream is("blue green");
    is >> w;
    CHECK(Colour(w) == Colour::blue);
    is >> w;
    CHECK(Colour(w) == Colour::green);
    CHECK(w("red") == Colour::red && Colour(w) == Colour::red);
    CHECK_THROWS_AS(w("purple"), std::out_of_range); // map.at
    std::istringstream bad("purple");
    CHECK_THROWS_AS(bad >> w, EnumWrap::validation_error); // an unknown token is a validation error of the option
    CHECK_THROWS_AS(bad >> w, std::logic_error); // ... which is a std::logic_error, as boost's validation_error
    addConfig("[option]\ncolour = mauve\n");
    CHECK_THROWS_AS(Configured<Config1>::getConfiguration<Wrapped>(colourName, Wrapped(Colour::blue)), EnumWrap::validation_error);
    Configurator::clear();
    // setMap replaces the map (EnumWrapper.hpp:84-88); MAP_ENUM is the helper its documentation describes
    MAP_ENUM(Colour, { "rouge", Colour::red }, { "vert", Colour::green });
    CHECK(w("vert") == Colour::green);
    CHECK_THROWS_AS(w("green"), std::out_of_range);
}

static void test_command_line_parser()
{ // core/test/CommandLineParser_test.cpp:19-40
    ArgV argv1({ "nextsimdg", "--config-file", "config.cfg" });
    CommandLineParser clp1(argv1.argc(), argv1());
    CHECK(clp1.getConfigFileNames().size() == 1 && clp1.getConfigFileNames()[0] == "config.cfg");
    ArgV argv2({ "nextsimdg", "--config-file", "config.cfg", "--config-files", "test.cfg", "more.cfg", "final.cfg" });
    CommandLineParser clp2(argv2.argc(), argv2());
    const auto cfgs = clp2.getConfigFileNames();
    CHECK(cfgs.size() == 4 && cfgs[0] == "config.cfg" && cfgs[1] == "test.cfg" && cfgs.back() == "final.cfg");
    ArgV argv3({ "nextsimdg", "-h" });
    CHECK(CommandLineParser(argv3.argc(), argv3()).helpRequested());
}

class Counter : public Iterator::Iterant { // core/test/Iterator_test.cpp:17-48
public:
    int starts = 0, iterates = 0, stops = 0, lastDt = 0;
    void init() override { }
    void start(const Iterator::TimePoint&) override { ++starts; }
    void iterate(const Iterator::Duration& dt) override
    {
        ++iterates;
        lastDt = dt;
    }
    void stop(const Iterator::TimePoint&) override { ++stops; }
};

static void test_iterator()
{ // core/test/Iterator_test.cpp:50-65
    Counter c;
    Iterator it(&c);
    it.setStartStopStep(0, 5, 1);
    it.run();
    CHECK(c.iterates == 5 && c.starts == 1 && c.stops == 1);
    Counter d;
    it.setIterant(&d);
    it.parseAndSet("10", "999", "30", "7"); // run_length takes precedence over stop
    it.run();
    CHECK(d.iterates == 5 && d.lastDt == 7); // t = 10, 17, 24, 31, 38 < 40
    it.parseAndSet("0", "1", "", "1"); // run/dev1.cfg: one iterate(1)
    Counter e;
    it.setIterant(&e);
    it.run();
    CHECK(e.iterates == 1);
}

static void test_timer()
{ // tick/tock tree (core/test/Timer_test.cpp:18-86 only prints; here the tree is asserted)
    Timer t("root");
    t.tick("outer");
    for (int i = 0; i < 3; ++i) {
        t.tick("inner");
        t.tock("inner");
    }
    t.tick("other");
    t.tock();
    CHECK_THROWS_AS(t.tock("inner"), std::logic_error); // the running node is "outer"
    t.tock("outer");
    t.tock(); // at the root: no-op
    CHECK(t.ticks({ "outer" }) == 1 && t.ticks({ "outer", "inner" }) == 3 && t.ticks({ "outer", "other" }) == 1);
    CHECK(t.ticks({ "inner" }) == 0 && t.wallSeconds({ "nope" }) == 0.);
    CHECK(t.wallSeconds({ "outer" }) >= t.wallSeconds({ "outer", "inner" }));
    int synced = 0;
    t.setDeviceSync([&synced] { ++synced; });
    {
        t.tick("with sync");
        t.tock();
    }
    CHECK(synced == 1);
    std::ostringstream os;
    t.report(os);
    const std::string rep = os.str();
    CHECK(rep.find("root: ticks = 1") == 0);
    CHECK(rep.find("+- outer: ticks = 1") != std::string::npos && rep.find("inner: ticks = 3") != std::string::npos);
    CHECK(rep.find("% of parent") != std::string::npos && rep.find("ms/tick") != std::string::npos);
    {
        ScopedTimer s("scoped");
    }
    CHECK(Timer::main.ticks({ "scoped" }) == 1);
}

static void test_physics_config()
{ // physics/test/NextsimPhysics_test.cpp:21-45 + [Modules] selection as at :178-189
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    addConfig("[nextsim_thermo]\nmin_conc = 2e-12\nmin_thick = 0.02\nI_0 = 0.18\n");
    NextsimPhysics nsphys;
    nsphys.configure();
    CHECK(NextsimPhysics::minimumIceConcentration() == 2e-12);
    CHECK(NextsimPhysics::minimumIceThickness() == 0.02);
    CHECK(NextsimPhysics::i0() == 0.18);
    nsdg_column_params p;
    nsphys.describe(p);
    CHECK(p.min_conc == 2e-12 && p.min_thick == 0.02 && p.i0 == 0.18);
    CHECK(p.albedo_kind == NSDG_ALBEDO_SMU && p.freezing_kind == NSDG_FREEZING_LINEAR && p.ks == 0.3096 && p.flooding == 1);
    Configurator::clear();
    addConfig("[Modules]\nNextsim::IFreezingPoint = Nextsim::UnescoFreezing\nNextsim::IIceAlbedo = Nextsim::CCSMIceAlbedo\n\n"
              "[CCSMIceAlbedo]\niceAlbedo = 0.63\nsnowAlbedo = 0.88\n[thermoice0]\nflooding = false\n[Hibler]\nh0 = 0.3\n");
    ConfiguredModule::parseConfigurator();
    nsphys.configure();
    nsphys.describe(p);
    CHECK(p.albedo_kind == NSDG_ALBEDO_CCSM && p.freezing_kind == NSDG_FREEZING_UNESCO);
    CHECK(p.ccsm_ice_albedo == 0.63 && p.ccsm_snow_albedo == 0.88 && p.flooding == 0 && p.h0 == 0.3 && p.phi_m == 0.5);
    CHECK(p.min_conc == 1e-12 && p.i0 == 0.17); // back to the defaults
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    nsphys.configure();
}

static void test_subcycle_policy()
{ // the stability rule of the sub-cycle lives in the library, once (nsdg_mevp_stable_params); DESIGN.md section 3.4
    auto fresh = [] {
        nsdg_mevp_params p;
        nsdg_mevp_default_params(&p);
        return p;
    };
    CHECK(fresh().aevp_c == 0. && fresh().delta_min == 2e-9); // a plain ABI user gets the uniform form of ABI 5
    for (double h : { 125., 250., 500. }) { // keep alpha = 1500, raise Delta_min -- and the way back
        nsdg_mevp_params p = fresh();
        CHECK(nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_ALPHA, h, 120.) == NSDG_OK && p.delta_min > 2e-9 && p.alpha == 1500. && p.beta == 1500. && p.aevp_c == 0.);
        nsdg_mevp_params q = fresh();
        q.delta_min = p.delta_min;
        CHECK(nsdg_mevp_stable_params(&q, NSDG_SUBCYCLE_KEEP_DELTA_MIN, h, 120.) == NSDG_OK && approx(q.alpha, 1500., 1e-9));
    }
    nsdg_mevp_params p = fresh();
    nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_ALPHA, 250., 120.);
    CHECK(approx(p.delta_min, 7.41e-7, 2e-3) && approx(nsdg_mevp_creep_percent_per_day(&p), 6.4, 1e-2)); // below 6.4 % per day the ice creeps
    p = fresh();
    nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_ALPHA, 500., 120.);
    CHECK(approx(p.delta_min, 1.853e-7, 2e-3));
    p = fresh();
    nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_ALPHA, 8000., 120.);
    CHECK(p.delta_min == 2e-9); // coarse meshes keep the literature's regularisation
    p = fresh();
    nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_KEEP_DELTA_MIN, 250., 120.);
    CHECK(approx(p.alpha, 28875., 1e-3) && p.beta == p.alpha); // ... which on a fine mesh asks for the alpha of rounds 1-4
    p = fresh();
    CHECK(nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_ADAPTIVE, 250., 120.) == NSDG_OK && approx(p.aevp_c, 2.4 * 2.4 * 9.869604401089358, 1e-12)
        && p.aevp_alpha_min == 50. && p.delta_min == 2e-9); // the hosts' default: local alpha / beta at the literature's Delta_min
    p = fresh();
    CHECK(nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_ADAPTIVE_CONVERGED, 500., 120.) == NSDG_OK && p.aevp_c > 56. && p.delta_min == 2e-9);
    CHECK(approx(p.aevp_alpha_min, 500., 1e-2)); // the converging form's floor = the bound's alpha at NSDG_AEVP_DELTA_REF: 500 / 1000 / 2000 at 500 / 250 / 125 m ...
    nsdg_mevp_stable_params(&p, NSDG_SUBCYCLE_ADAPTIVE_CONVERGED, 8000., 120.);
    CHECK(p.aevp_alpha_min == 50.); // ... and never below the literature's 50
    CHECK(nsdg_mevp_stable_params(&p, 9, 250., 120.) == NSDG_ERR_ARG && nsdg_mevp_stable_params(nullptr, 0, 250., 120.) == NSDG_ERR_ARG);
    // the host's configuration keys choose among the three
    Configurator::clear();
    DynamicsStep d0;
    d0.configure();
    CHECK(d0.subcycleChoice(250., 120.).mode == "adaptive" && d0.subcycleChoice(250., 120.).aevpC > 50. && d0.subcycleChoice(250., 120.).aevpAlphaMin == 50.);
    addConfig("[dynamics]\nsubcycle = adaptive_converged\n");
    DynamicsStep d4;
    d4.configure();
    CHECK(d4.subcycleChoice(250., 120.).aevpC > 50. && approx(d4.subcycleChoice(250., 120.).aevpAlphaMin, 1000., 1e-2));
    Configurator::clear();
    addConfig("[dynamics]\nalpha = 1500\n");
    DynamicsStep d1;
    d1.configure();
    CHECK(d1.subcycleChoice(250., 120.).mode == "keep_alpha" && approx(d1.subcycleChoice(250., 120.).deltaMin, 7.41e-7, 2e-3) && d1.subcycleChoice(250., 120.).aevpC == 0.);
    Configurator::clear();
    addConfig("[dynamics]\nsubcycle = keep_delta_min\n");
    DynamicsStep d2;
    d2.configure();
    CHECK(approx(d2.subcycleChoice(250., 120.).alpha, 28875., 1e-3) && d2.subcycleChoice(250., 120.).deltaMin == 2e-9);
    Configurator::clear();
    addConfig("[dynamics]\nsubcycle = sometimes\n");
    DynamicsStep d3;
    CHECK_THROWS_AS(d3.configure(), std::invalid_argument);
    Configurator::clear();
}

static void test_structure()
{ // core/test/DevGrid_test.cpp:24-103, StructureFactory_test.cpp:19-47, ElementData_test.cpp:57-63
    Configurator::clear();
    DevGrid grid;
    CHECK(grid.nx() == 10 && grid.ny() == 10 && grid.size() == 100 && grid.nIceLayers() == 1);
    CHECK(grid.structureTypeCheck("DevGrid") && grid.structureTypeCheck("devgrid") && !grid.structureTypeCheck("rectgrid"));
    const int nx = 10;
    int count = 0;
    for (grid.cursor = 0; grid.cursor; ++grid.cursor) {
        const int slow = count / nx, fast = count % nx; // cursor order = linear index slow*nx + fast
        const double fractional = slow * 0.01 + fast * 0.0001;
        *grid.cursor = PrognosticGenerator().hice(1 + fractional).cice(2 + fractional).sst(3 + fractional).sss(4 + fractional).hsnow(5 + fractional).tice({ -(1. + fractional) });
        ++count;
    }
    CHECK(count == 100);
    const std::string path = "/tmp/nsdg_host_test_restart.nsdg";
    grid.dump(path);
    auto again = StructureFactory::generateFromFile(path);
    CHECK(again->structureType() == "devgrid");
    again->init(path);
    int targetIndex = 7 * nx + 3; // linear index i*nx + j (core/src/DevGridIO.cpp:107-109)
    again->cursor = 0;
    for (int k = 0; k < targetIndex; ++k)
        ++again->cursor;
    CHECK(again->cursor->iceThickness() == 1.0703);
    CHECK(again->cursor->iceConcentration() == 2.0703 && again->cursor->iceTemperature(0) == -1.0703);
    CHECK(again->fields().hice[targetIndex] == 1.0703); // SoA plane and cursor agree
    CHECK(StructureFactory::generate("devgrid")->structureType() == "devgrid");
    CHECK_THROWS_AS(StructureFactory::generate("notagrid"), std::invalid_argument);
    // size-configurable grid
    addConfig("[rectgrid]\nnx = 7\nny = 5\nnLayers = 3\n");
    auto rect = StructureFactory::generate("rectgrid");
    rect->init("");
    CHECK(rect->nx() == 7 && rect->ny() == 5 && rect->nIceLayers() == 3 && rect->size() == 35);
    rect->cursor = 0;
    ElementData& d = *rect->cursor;
    d = PrognosticGenerator().hice(0.1).cice(0.5).sst(-1).sss(32).hsnow(0.01).tice({ -1., -2., -3. });
    CHECK(d.iceThickness() == 0.1 && d.iceConcentration() == 0.5 && d.snowThickness() == 0.01);
    CHECK(d.seaSurfaceTemperature() == -1 && d.seaSurfaceSalinity() == 32 && d.iceTemperature(0) == -1. && d.iceTemperature(2) == -3.);
    CHECK(d.iceTrueThickness() == 0.2 && d.snowTrueThickness() == 0.02);
    DummyExternalData::setAll(*rect);
    CHECK(rect->fields().tair[34] == -1 && rect->fields().qlw[0] == 311 && rect->fields().mixrat[3] == -1.);
    Configurator::clear();
    std::remove(path.c_str());
}

// ---------------------------------------------------------------------------------- GPU cases
static void test_hipstep_melting()
{ // core/test/ElementData_test.cpp:19-87 / physics/test/NextsimPhysics_test.cpp:176-243 through the plugin path
    Configurator::clear();
    addConfig("[Modules]\nNextsim::IFreezingPoint = Nextsim::UnescoFreezing\nNextsim::IIceAlbedo = Nextsim::CCSMIceAlbedo\n"
              "Nextsim::IPhysics1d = Nextsim::NextsimPhysics\n\n[CCSMIceAlbedo]\niceAlbedo = 0.63\nsnowAlbedo = 0.88\n"
              "[rectgrid]\nnx = 3\nny = 4\nnLayers = 3\n");
    ModuleLoader::getLoader().setAllDefaults();
    ConfiguredModule::parseConfigurator();
    auto grid = StructureFactory::generate("rectgrid");
    grid->init("");
    for (grid->cursor = 0; grid->cursor; ++grid->cursor) {
        ElementData& data = *grid->cursor;
        data = PrognosticGenerator().hice(0.1).cice(0.5).sst(-1).sss(32).hsnow(0.01).tice({ -1., -1., -1. });
        data.airTemperature() = 3;
        data.dewPoint2m() = 2;
        data.airPressure() = 100000;
        data.mixedLayerDepth() = 10.;
        data.incomingLongwave() = 330;
        data.incomingShortwave() = 50;
        data.snowfall() = 0;
        data.windSpeed() = 5;
    }
    HipStep step;
    step.setInitialData(*grid);
    step.init();
    step.start(0);
    step.iterate(600);
    step.stop(600);
    CHECK(step.launches() == 1);
    const FieldStore& f = grid->fields();
    for (std::size_t e = 0; e < f.n; e += 5) {
        // expected updated true thickness/concentration of the reference test, as cell means
        CHECK(approx(f.cice[e], 0.368269, 1e-4));
        CHECK(approx(f.hice[e] / f.cice[e], 0.12846, 1e-4));
        CHECK(approx(f.hsnow[e] / f.cice[e], 0.01957732, 1e-4));
        CHECK(f.tice[e] == 0.0);
        CHECK(approx(f.cice[e], 0.36826938031129286, 1e-12)); // SURVEY.md Appendix C probe
        CHECK(f.tice[f.n + e] == 0.0 && f.tice[2 * f.n + e] == 0.0); // layers > 0 are zeroed on write-back (A.7 quirk 5)
        CHECK(f.sst[e] == -1 && f.sss[e] == 32); // never updated
    }
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

static std::string goldenDir()
{
    const char* d = std::getenv("NSDG_GOLDEN_DIR");
    return d ? std::string(d) : std::string("tests/golden");
}

static void test_restart_hdf5()
{ // the reference's NetCDF-4 restart format (core/src/DevGridIO.cpp:65-210)
    const std::string ref = goldenDir() + "/dev1.res.nc"; // the reference's run/dev1.res.nc (written by run/dev_res.py)
    CHECK(Hdf5File::isHdf5(ref));
    CHECK(!Hdf5File::isHdf5("/nonexistent/file.nc"));
    // lookup3 known answers (the superblock / object header checksums of the file are verified while parsing)
    CHECK(hdf5Checksum(reinterpret_cast<const unsigned char*>(""), 0) == 0xdeadbeefu);
    CHECK(hdf5Checksum(reinterpret_cast<const unsigned char*>("Four score and seven years ago"), 30) == 0x17770551u);
    {
        const Hdf5File f(ref);
        const std::vector<std::string> root = f.listGroup("/");
        CHECK(root.size() == 2 && root[0] == "data" && root[1] == "structure");
        CHECK(f.stringAttribute("/structure", "type") == "devgrid"); // run/dev_res.py:4
        CHECK(f.hasAttribute("/", "_NCProperties") && !f.hasAttribute("/structure", "nope"));
        CHECK(f.listGroup("/data").size() == 9); // 6 variables + the dimension scales x, y, nLayers (dense link storage)
        CHECK((f.dims("/data/hice") == std::vector<std::uint64_t> { 10, 10 }));
        CHECK((f.dims("/data/tice") == std::vector<std::uint64_t> { 10, 10, 1 }));
        const std::pair<const char*, double> expect[] = { { "cice", 0.5 }, { "hice", 0.1 }, { "hsnow", 0.0 }, { "sss", 32. }, { "sst", -1. },
            { "tice", -1. } }; // run/dev_res.py:10-20
        for (const auto& e : expect) {
            const std::vector<double> v = f.readDoubles(std::string("/data/") + e.first);
            bool all = v.size() == 100;
            for (double x : v)
                all = all && x == e.second;
            CHECK(all);
        }
        CHECK_THROWS_AS(f.readDoubles("/data/x"), Hdf5Error); // a dimension without coordinate values has no storage
        CHECK_THROWS_AS(f.readDoubles("/data/nothere"), Hdf5Error);
        CHECK(f.exists("/data/sst") && !f.exists("/data/nothere"));
    }
    // the structure factory and DevGrid::init read the reference's file directly
    CHECK(RectGrid::typeInFile(ref) == "devgrid");
    auto grid = StructureFactory::generateFromFile(ref);
    CHECK(grid->structureType() == "devgrid");
    grid->init(ref);
    CHECK(grid->nx() == 10 && grid->ny() == 10 && grid->nIceLayers() == 1);
    int n = 0;
    bool same = true;
    for (grid->cursor = 0; grid->cursor; ++grid->cursor, ++n)
        same = same && grid->cursor->iceThickness() == 0.1 && grid->cursor->iceConcentration() == 0.5 && grid->cursor->snowThickness() == 0.
            && grid->cursor->seaSurfaceTemperature() == -1. && grid->cursor->seaSurfaceSalinity() == 32. && grid->cursor->iceTemperature(0) == -1.;
    CHECK(n == 100 && same);
    // write -> read round trip in the same format, the index pattern of core/test/DevGrid_test.cpp:36-96
    addConfig("[rectgrid]\nnx = 6\nny = 9\nnLayers = 2\n");
    auto rect = StructureFactory::generate("rectgrid");
    rect->init("");
    CHECK(rect->nx() == 6 && rect->ny() == 9 && rect->nIceLayers() == 2);
    int count = 0;
    for (rect->cursor = 0; rect->cursor; ++rect->cursor, ++count) {
        const double fr = (count / 9) * 0.01 + (count % 9) * 0.0001;
        *rect->cursor = PrognosticGenerator().hice(1 + fr).cice(2 + fr).sst(3 + fr).sss(4 + fr).hsnow(5 + fr).tice({ -(1. + fr), -(2. + fr) });
    }
    const std::string path = "/tmp/nsdg_host_test_restart.nc";
    rect->dump(path);
    if (const char* keep = std::getenv("NSDG_KEEP_RESTART")) // lets the Python test hand the file to the HDF5 library's h5dump
        rect->dump(keep);
    CHECK(Hdf5File::isHdf5(path) && RectGrid::typeInFile(path) == "rectgrid");
    {
        const Hdf5File f(path);
        CHECK((f.dims("/data/tice") == std::vector<std::uint64_t> { 6, 9, 2 }));
        CHECK(f.readDoubles("/data/hice")[4 * 9 + 3] == 1.0403); // element (i, j) at i*ny + j (core/src/DevGridIO.cpp:107-109)
        CHECK(f.readDoubles("/data/tice")[(4 * 9 + 3) * 2 + 1] == -(2. + (4 * 0.01 + 3 * 0.0001))); // (x, y, nLayers), layers fastest
    }
    auto again = StructureFactory::generateFromFile(path);
    again->init(path);
    CHECK(again->nx() == 6 && again->ny() == 9 && again->nIceLayers() == 2);
    CHECK(again->fields().hice == rect->fields().hice && again->fields().sss == rect->fields().sss && again->fields().tice == rect->fields().tice);
    // damaged files are rejected, not misread
    {
        std::ifstream in(path, std::ios::binary);
        std::vector<char> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        std::vector<char> cut(bytes.begin(), bytes.begin() + 200);
        std::ofstream("/tmp/nsdg_host_test_cut.nc", std::ios::binary).write(cut.data(), (std::streamsize)cut.size());
        CHECK_THROWS_AS(Hdf5File("/tmp/nsdg_host_test_cut.nc").listGroup("/data"), Hdf5Error);
        bytes[60] ^= 0x40; // inside the root object header: the checksum no longer matches
        std::ofstream("/tmp/nsdg_host_test_bad.nc", std::ios::binary).write(bytes.data(), (std::streamsize)bytes.size());
        CHECK_THROWS_AS(Hdf5File("/tmp/nsdg_host_test_bad.nc").listGroup("/"), Hdf5Error);
    }
    std::remove(path.c_str());
    std::remove("/tmp/nsdg_host_test_cut.nc");
    std::remove("/tmp/nsdg_host_test_bad.nc");
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

static void test_model_dev1()
{ // BASELINE config 1: run/dev1.cfg (start 0, stop 1, time_step 1) on the 10x10 devgrid, Dummy forcing
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    // init_file = the reference's own NetCDF-4 restart file (run/dev1.res.nc), final_file in the same format
    addConfig("[model]\ninit_file = " + goldenDir() + "/dev1.res.nc\nstart = 0\nstop = 1\ntime_step = 1\nfinal_file = /tmp/nsdg_dev1_restart.nc\n");
    ConfiguredModule::parseConfigurator();
    {
        Model model;
        model.configure();
        model.run();
        const FieldStore& f = model.structure().fields();
        CHECK(f.n == 100 && model.step().launches() == 1 && dynamic_cast<HipStep*>(&model.step()) != nullptr);
        for (std::size_t e = 0; e < f.n; ++e) {
            CHECK(approx(f.hice[e], 0.04668325240678619, 1e-12));
            CHECK(approx(f.cice[e], 0.36670813101696548, 1e-12));
            CHECK(f.hsnow[e] == 0.0);
            CHECK(approx(f.tice[e], -1.444501803353837, 1e-12));
            CHECK(f.sst[e] == -1.0);
        }
    } // ~Model writes the restart file
    CHECK(Hdf5File::isHdf5("/tmp/nsdg_dev1_restart.nc") && RectGrid::typeInFile("/tmp/nsdg_dev1_restart.nc") == "devgrid");
    auto again = StructureFactory::generateFromFile("/tmp/nsdg_dev1_restart.nc");
    again->init("/tmp/nsdg_dev1_restart.nc");
    CHECK(approx(again->fields().cice[99], 0.36670813101696548, 1e-12));
    std::remove("/tmp/nsdg_dev1_restart.nc");
    Configurator::clear();
}

static void test_dynamics_step()
{ // the dynamics core as an IModelStep plugin: [Modules] Nextsim::IModelStep = Nextsim::DynamicsStep
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    addConfig("[Modules]\nNextsim::IModelStep = Nextsim::DynamicsStep\n[model]\nstructure = rectgrid\nstart = 0\nstop = 240\ntime_step = 120\n"
              "final_file = /tmp/nsdg_dyn_restart.nsdg\n[rectgrid]\nnx = 40\nny = 56\n[init]\nhice = 0.3\ncice = 0.9\n[dynamics]\nnsub = 10\n");
    ConfiguredModule::parseConfigurator();
    {
        Model model;
        model.configure();
        CHECK(dynamic_cast<DynamicsStep*>(&model.step()) != nullptr);
        model.run();
        DynamicsStep& dyn = dynamic_cast<DynamicsStep&>(model.step());
        const FieldStore& f = model.structure().fields();
        CHECK(dyn.launches() == 2 && f.n == 40 * 56);
        CHECK(dyn.maxSpeed() > 1e-6 && dyn.maxSpeed() < 1.0);
        CHECK(approx(dyn.sumH(), 0.3 * f.n, 1e-12)); // closed box: mass is conserved to round-off
        CHECK(approx(dyn.sumA(), 0.9 * f.n, 1e-12));
        double lo = 1e9, hi = -1e9;
        for (double h : f.hice) {
            lo = std::min(lo, h);
            hi = std::max(hi, h);
        }
        CHECK(lo > 0.29 && hi < 0.31 && hi > lo); // advected, not blown up
    }
    CHECK(RectGrid::typeInFile("/tmp/nsdg_dyn_restart.nsdg") == "rectgrid");
    std::remove("/tmp/nsdg_dyn_restart.nsdg");
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

static void test_dynamics_reports_a_run_that_blows_up()
{ // a relaxation parameter of 1e-200 makes the stress overflow within two sub-iterations (1 / alpha = 1e200): the fields are
  // non-finite at the end of the run; the step must say so (exception -> non-zero exit of nextsim_amd) and no restart file may appear
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    std::remove("/tmp/nsdg_dyn_blowup.nsdg");
    addConfig("[Modules]\nNextsim::IModelStep = Nextsim::DynamicsStep\n[model]\nstructure = rectgrid\nstart = 0\nstop = 1200\ntime_step = 120\n"
              "final_file = /tmp/nsdg_dyn_blowup.nsdg\n[rectgrid]\nnx = 64\nny = 48\n[init]\nhice = 0.3\ncice = 1.0\n[dynamics]\nnsub = 12\nalpha = 1e-200\nbeta = 1e-200\n");
    ConfiguredModule::parseConfigurator();
    bool threw = false;
    try {
        Model model;
        model.configure();
        model.run();
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("left the physical range") != std::string::npos;
    }
    CHECK(threw);
    std::ifstream probe("/tmp/nsdg_dyn_blowup.nsdg");
    CHECK(!probe.good()); // ~Model's restart write went through the same check and was abandoned
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

static void test_rendezvous()
{ // two "ranks" of a multi-process run agree on 128 bytes over the loopback interface
    int r0, r1, covered = 0;
    for (int world : { 1, 3, 8 })
        for (int r = 0; r < world; ++r) {
            DynamicsStep::splitRows(2048, world, r, r0, r1);
            CHECK(r0 == covered % 2048 || r == 0);
            covered = r1;
            CHECK(r1 > r0 && (r != world - 1 || r1 == 2048));
        }
    RankEnvironment a, b;
    a.world = b.world = 2;
    a.rank = 0, b.rank = 1;
    a.masterPort = b.masterPort = 20000 + (int)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 20000);
    unsigned char sent[128], got[128];
    for (int i = 0; i < 128; ++i)
        sent[i] = (unsigned char)(3 * i + 1), got[i] = 0;
    std::exception_ptr err;
    std::thread client([&] {
        try {
            broadcastFromRankZero(b, got, sizeof got, 20);
        } catch (...) {
            err = std::current_exception();
        }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(300)); // the client starts first and has to retry
    broadcastFromRankZero(a, sent, sizeof sent, 20);
    client.join();
    CHECK(!err);
    CHECK(std::memcmp(sent, got, sizeof sent) == 0);
    RankEnvironment lonely; // nobody listens: a clear error after the timeout, not a hang
    lonely.world = 2, lonely.rank = 1, lonely.masterPort = a.masterPort + 100;
    CHECK_THROWS_AS(broadcastFromRankZero(lonely, got, sizeof got, 1), std::runtime_error);
    // ... and the other way round: rank 0 serves, a rank has died before it connected
    RankEnvironment server;
    server.world = 3, server.rank = 0, server.masterPort = a.masterPort + 200;
    CHECK_THROWS_AS(broadcastFromRankZero(server, sent, sizeof sent, 1), std::runtime_error);
}

static void test_restart_gather()
{ // the owned rows of every rank of a multi-process run reach rank 0, which writes the ONE restart file
    const int nx = 7, ny = 10, world = 3;
    auto filled = [&](int rank) { // what rank `rank` holds at the end: its own rows current, every other row stale
        FieldStore f;
        f.resize((std::size_t)nx * ny, 1);
        int r0, r1;
        DynamicsStep::splitRows(ny, world, rank, r0, r1);
        for (int e = 0; e < nx * ny; ++e) {
            const bool mine = e / nx >= r0 && e / nx < r1;
            const double stale = -1000. - rank;
            f.hice[e] = mine ? 1. + e : stale, f.cice[e] = mine ? 2. + e : stale, f.hsnow[e] = mine ? 3. + e : stale;
            f.tice[e] = mine ? 4. + e : stale, f.newice[e] = mine ? 5. + e : stale;
        }
        return f;
    };
    for (bool thermo : { false, true }) {
        RankEnvironment env[world];
        const int port = 20000 + (int)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 20000) + (thermo ? 50 : 40);
        for (int r = 0; r < world; ++r)
            env[r].world = world, env[r].rank = r, env[r].masterPort = port;
        FieldStore root = filled(0);
        std::vector<std::thread> senders;
        std::exception_ptr err[world];
        for (int r = 1; r < world; ++r)
            senders.emplace_back([&, r] {
                try {
                    FieldStore f = filled(r);
                    int r0, r1;
                    DynamicsStep::splitRows(ny, world, r, r0, r1);
                    const std::vector<double> rows = DynamicsStep::packRows(f, thermo, nx, r0, r1);
                    gatherToRankZero(env[r], rows.data(), rows.size() * sizeof(double), nullptr, 20);
                } catch (...) {
                    err[r] = std::current_exception();
                }
            });
        int delivered = 0;
        gatherToRankZero(env[0], nullptr, 0, [&](int rank, const char* data, std::size_t bytes) {
            int a, b;
            DynamicsStep::splitRows(ny, world, rank, a, b);
            DynamicsStep::placeRows(root, thermo, nx, a, b, reinterpret_cast<const double*>(data), bytes / sizeof(double));
            ++delivered;
        }, 20);
        for (auto& t : senders)
            t.join();
        CHECK(delivered == world - 1 && !err[1] && !err[2]);
        bool ok = true;
        for (int e = 0; e < nx * ny; ++e) {
            ok = ok && root.hice[e] == 1. + e && root.cice[e] == 2. + e; // every row current, none stale
            int r0, r1;
            DynamicsStep::splitRows(ny, world, 0, r0, r1);
            const bool own = e / nx < r1;
            // without thermodynamics the column planes do not travel (they never change): rank 0 keeps its own
            ok = ok && root.hsnow[e] == ((thermo || own) ? 3. + e : -1000.) && root.newice[e] == ((thermo || own) ? 5. + e : -1000.);
        }
        CHECK(ok);
        const std::vector<double> few(5, 0.);
        CHECK_THROWS_AS(DynamicsStep::placeRows(root, thermo, nx, 0, 3, few.data(), few.size()), std::runtime_error);
    }
    // EIGHT ranks (the node the scaling runs use), acknowledged: no sender returns before rank 0's commit (the write of the
    // restart file) has run; if the commit fails, or a rank reports a failure instead of delivering, EVERY rank throws
    for (int scenario = 0; scenario < 3; ++scenario) { // 0: all well, 1: the commit throws, 2: rank 5 reports a failure
        const int W = 8;
        RankEnvironment env[W];
        const int port = 20000 + (int)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 20000) + 70 + 3 * scenario;
        for (int r = 0; r < W; ++r)
            env[r].world = W, env[r].rank = r, env[r].masterPort = port;
        std::atomic<bool> committed(false);
        std::atomic<int> returnedBeforeCommit(0), threw(0);
        std::vector<std::thread> senders;
        for (int r = 1; r < W; ++r)
            senders.emplace_back([&, r] {
                try {
                    if (scenario == 2 && r == 5) {
                        reportFailureToRankZero(env[r], 20);
                        ++threw; // such a rank rethrows its own error
                        return;
                    }
                    const std::vector<double> rows(1000 + r, (double)r);
                    gatherToRankZero(env[r], rows.data(), rows.size() * sizeof(double), nullptr, 20);
                    if (!committed)
                        ++returnedBeforeCommit;
                } catch (const std::runtime_error&) {
                    ++threw;
                }
            });
        int delivered = 0;
        bool rootThrew = false;
        try {
            gatherToRankZero(
                env[0], nullptr, 0,
                [&](int rank, const char* data, std::size_t bytes) {
                    const double* d = reinterpret_cast<const double*>(data);
                    if (bytes == (1000u + (unsigned)rank) * sizeof(double) && d[0] == (double)rank && d[999 + rank] == (double)rank)
                        ++delivered;
                },
                20,
                [&]() {
                    std::this_thread::sleep_for(std::chrono::milliseconds(200)); // a sender that does not wait would be caught
                    if (scenario == 1)
                        throw std::runtime_error("disk full");
                    committed = true;
                });
        } catch (const std::runtime_error&) {
            rootThrew = true;
        }
        for (auto& t : senders)
            t.join();
        if (scenario == 0)
            CHECK(delivered == W - 1 && committed && !rootThrew && threw == 0 && returnedBeforeCommit == 0);
        else
            CHECK(rootThrew && !committed && threw == W - 1); // every rank of a run without a restart file fails
    }
    // a rank that died before the end never delivers: rank 0 gives up after the timeout instead of hanging
    RankEnvironment alone;
    alone.world = 2, alone.rank = 0, alone.masterPort = 20000 + (int)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 20000) + 60;
    CHECK_THROWS_AS(gatherToRankZero(alone, nullptr, 0, [](int, const char*, std::size_t) {}, 1), std::runtime_error);
    RankEnvironment orphan = alone; // ... and a sender whose rank 0 has gone
    orphan.rank = 1, orphan.masterPort += 7;
    const double x = 1.;
    CHECK_THROWS_AS(gatherToRankZero(orphan, &x, sizeof x, nullptr, 1), std::runtime_error);
}

static void fillDynamicsState(FieldStore& f, int nx, int ny, double salt)
{ // nx = slow ("x" of the file), ny = fast
    f.dyn.resize((std::size_t)nx, (std::size_t)ny);
    auto fill = [&](std::vector<double>& v, double base) {
        for (std::size_t k = 0; k < v.size(); ++k)
            v[k] = base + salt + 1e-3 * (double)k;
    };
    fill(f.dyn.hdg, 1.), fill(f.dyn.adg, 2.), fill(f.dyn.u, 3.), fill(f.dyn.v, 4.), fill(f.dyn.s11, 5.), fill(f.dyn.s12, 6.), fill(f.dyn.s22, 7.);
    f.dyn.present = true;
}

static void test_restart_with_dynamics_state()
{ // what a dynamics run carries between steps travels in the restart file, in both formats (RectGrid.hpp), and between the ranks
    for (const char* path : { "/tmp/nsdg_host_test_dyn_restart.nc", "/tmp/nsdg_host_test_dyn_restart.nsdg" }) {
        RectGrid g;
        g.resize(5, 7, 2);
        FieldStore& f = g.fields();
        for (std::size_t e = 0; e < f.n; ++e)
            f.hice[e] = 0.1 * e, f.cice[e] = 0.5, f.newice[e] = 1e-5 * e;
        fillDynamicsState(f, 5, 7, 0.25);
        g.dump(path);
        auto again = StructureFactory::generateFromFile(path);
        again->init(path);
        const FieldStore& r = again->fields();
        CHECK(r.dyn.present && r.dyn.hdg == f.dyn.hdg && r.dyn.adg == f.dyn.adg && r.dyn.u == f.dyn.u && r.dyn.v == f.dyn.v);
        CHECK(r.dyn.s11 == f.dyn.s11 && r.dyn.s12 == f.dyn.s12 && r.dyn.s22 == f.dyn.s22 && r.newice == f.newice && r.hice == f.hice);
        if (Hdf5File::isHdf5(path)) { // named dimensions as the reference names its own (core/src/DevGridIO.cpp:169-172)
            const Hdf5File h(path);
            CHECK(h.dims("/data/hice_dg") == (std::vector<std::uint64_t> { 5, 5, 7 }) && h.dims("/data/u") == (std::vector<std::uint64_t> { 11, 15 }));
            CHECK(h.dims("/data/s12") == (std::vector<std::uint64_t> { 8, 5, 7 }) && h.exists("/data/stress8") && h.exists("/data/xnode"));
        }
        // a file without the state (the reference's, a column-only run's): the dynamics start from rest
        RectGrid plain;
        plain.resize(5, 7, 2);
        plain.dump(path);
        again->init(path);
        CHECK(!again->fields().dyn.present);
        std::remove(path);
    }
    // the rows of a rank travel with their share of the state
    const int nx = 7, ny = 10; // here nx = the fast dimension of the dynamics (row length), ny = rows
    FieldStore a, b;
    a.resize((std::size_t)nx * ny, 1), b.resize((std::size_t)nx * ny, 1);
    fillDynamicsState(a, ny, nx, 0.5);
    for (const auto& rows : { std::make_pair(3, 6), std::make_pair(6, 10) }) { // an interior block and the last one (which owns the top node row)
        const std::vector<double> packed = DynamicsStep::packRows(a, false, nx, rows.first, rows.second);
        const std::size_t nodeRows = 2 * (std::size_t)(rows.second - rows.first) + (rows.second == ny ? 1 : 0);
        CHECK(packed.size() == (std::size_t)(rows.second - rows.first) * nx * (2 + 34) + 2 * nodeRows * (2 * nx + 1));
        DynamicsStep::placeRows(b, false, nx, rows.first, rows.second, packed.data(), packed.size());
        bool ok = b.dyn.present;
        for (int c = 0; c < 8 && ok; ++c)
            for (int e = rows.first * nx; e < rows.second * nx; ++e)
                ok = ok && b.dyn.s12[(std::size_t)c * a.n + e] == a.dyn.s12[(std::size_t)c * a.n + e] && (c >= 5 || b.dyn.adg[(std::size_t)c * a.n + e] == a.dyn.adg[(std::size_t)c * a.n + e]);
        for (std::size_t k = 2 * (std::size_t)rows.first * (2 * nx + 1); k < (2 * (std::size_t)rows.first + nodeRows) * (2 * nx + 1) && ok; ++k)
            ok = ok && b.dyn.u[k] == a.dyn.u[k] && b.dyn.v[k] == a.dyn.v[k];
        CHECK(ok);
        CHECK(b.dyn.u[0] == 0. && b.dyn.s11[0] == 0.); // rows nobody delivered stay as they were
        CHECK_THROWS_AS(DynamicsStep::placeRows(b, false, nx, rows.first, rows.second, packed.data(), packed.size() - 1), std::runtime_error);
    }
}

static std::vector<char> fileBytes(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void run_dynamics_to_file(const std::string& model, const std::string& extra, const std::string& finalFile)
{
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    addConfig("[Modules]\nNextsim::IModelStep = Nextsim::DynamicsStep\n[model]\nstructure = rectgrid\ntime_step = 120\n" + model + "final_file = " + finalFile
        + "\n[rectgrid]\nnx = 128\nny = 96\n[init]\nhice = 0.3\ncice = 0.9\nsst = -1.76\nhsnow = 0.05\ntice = -8\n[dynamics]\nnsub = 23\nthermodynamics = true\nforcing = winter\n" + extra);
    ConfiguredModule::parseConfigurator();
    {
        Model model;
        model.configure();
        model.run();
    } // ~Model writes the restart file
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
}

static void test_dynamics_restart_is_exact()
{ // N steps == N/2 steps + restart file + N/2 steps, byte for byte, as one block and as eight (review of round 5: the restart dropped the
  // higher DG coefficients, the velocity and the stress -- a restarted run started from rest; the reference writes every prognostic
  // field it has, core/src/DevGridIO.cpp:169-201, and reads them back, :101-138)
    for (const char* ext : { ".nsdg", ".nc" }) {
        const std::string whole = std::string("/tmp/nsdg_rst_whole") + ext, half = std::string("/tmp/nsdg_rst_half") + ext, resumed = std::string("/tmp/nsdg_rst_resumed") + ext;
        for (const char* blocks : { "", "row_blocks = 8\npasses_per_exchange = 1\n" }) {
            run_dynamics_to_file("start = 0\nstop = 720\n", blocks, whole); // 6 steps
            run_dynamics_to_file("start = 0\nstop = 360\n", blocks, half); // 3 steps, restart file ...
            run_dynamics_to_file("init_file = " + half + "\nstart = 360\nstop = 720\n", blocks, resumed); // ... and 3 more from it
            const std::vector<char> w = fileBytes(whole), r = fileBytes(resumed), h = fileBytes(half);
            CHECK(w.size() > 128 * 96 * 8 * 40 && w == r); // the file holds the whole state, and the resumed run reproduces it byte for byte
            CHECK(h != w);
        }
        // and the decomposition does not show: the file of the 8-block run above == the file of a single block
        const std::vector<char> eight = fileBytes(whole);
        run_dynamics_to_file("start = 0\nstop = 720\n", "", whole);
        CHECK(fileBytes(whole) == eight);
        // a restart from the file of the OLD kind (cell means only) starts from rest: it must differ
        auto plain = StructureFactory::generateFromFile(half);
        plain->init(half);
        plain->fields().dyn.clear();
        plain->dump(half);
        run_dynamics_to_file("init_file = " + half + "\nstart = 360\nstop = 720\n", "", resumed);
        CHECK(fileBytes(resumed) != eight);
        for (const std::string& p : { whole, half, resumed })
            std::remove(p.c_str());
    }
}

static FieldStore run_dynamics(const std::string& extra, double* umax)
{
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    addConfig("[Modules]\nNextsim::IModelStep = Nextsim::DynamicsStep\n[model]\nstructure = rectgrid\nstart = 0\nstop = 360\ntime_step = 120\n"
              "final_file = /tmp/nsdg_dyn_blocks.nsdg\n[rectgrid]\nnx = 128\nny = 96\n[init]\nhice = 0.3\ncice = 0.9\nsst = -1.76\nhsnow = 0.05\ntice = -8\n"
              "[dynamics]\nnsub = 23\nthermodynamics = true\nforcing = winter\n" + extra);
    ConfiguredModule::parseConfigurator();
    FieldStore out;
    {
        Model model;
        model.configure();
        model.run();
        DynamicsStep& dyn = dynamic_cast<DynamicsStep&>(model.step());
        dyn.stop(0);
        *umax = dyn.maxSpeed();
        out = model.structure().fields();
    }
    std::remove("/tmp/nsdg_dyn_blocks.nsdg");
    Configurator::clear();
    ModuleLoader::getLoader().setAllDefaults();
    return out;
}

static void test_dynamics_row_blocks()
{ // the multi-block step (threads + in-process transport, native row-block drivers) against the single block: bit for bit
    double u1 = 0, u3 = 0, u4 = 0, ul = 0;
    const FieldStore one = run_dynamics("", &u1);
    CHECK(u1 > 1e-6 && u1 < 1.0);
    double lo = 1e9, hi = -1e9;
    for (double t : one.tice)
        lo = std::min(lo, t), hi = std::max(hi, t);
    CHECK(lo > -40 && hi <= 0 && hi > lo); // the device-side winter forcing and the wind coupling changed the ice temperature
    const FieldStore three = run_dynamics("row_blocks = 3\npasses_per_exchange = 2\n", &u3);
    const FieldStore four = run_dynamics("row_blocks = 4\npasses_per_exchange = 1\noverlap = false\ngraph = true\n", &u4);
    for (const FieldStore* f : { &three, &four }) {
        CHECK(f->hice == one.hice);
        CHECK(f->cice == one.cice);
        CHECK(f->hsnow == one.hsnow);
        CHECK(f->tice == one.tice);
        CHECK(f->newice == one.newice);
    }
    CHECK(u3 == u1 && u4 == u1);
    // RCCL from the C++ host on one GPU: an interior block of 8 whose neighbours are the rank itself
    const FieldStore loop = run_dynamics("loopback_world = 8\npasses_per_exchange = 1\n", &ul);
    bool finite = std::isfinite(ul);
    for (double h : loop.hice)
        finite = finite && std::isfinite(h);
    CHECK(finite && ul > 0);
}

int main(int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "--gpu") == 0;
    try {
        if (!gpu) {
            test_module_loader();
            test_configured_module();
            test_configurator();
            test_command_line_parser();
            test_enum_wrapper();
            test_iterator();
            test_timer();
            test_physics_config();
            test_subcycle_policy();
            test_structure();
            test_restart_hdf5();
            test_rendezvous();
            test_restart_gather();
            test_restart_with_dynamics_state();
        } else {
            test_hipstep_melting();
            test_model_dev1();
            test_dynamics_step();
            test_dynamics_row_blocks();
            test_dynamics_restart_is_exact();
            test_dynamics_reports_a_run_that_blows_up();
        }
    } catch (const std::exception& e) {
        std::printf("FAIL: unexpected exception: %s\n", e.what());
        ++failures;
    }
    std::printf("%s: %d checks, %d failures\n", gpu ? "host GPU tests" : "host CPU tests", checks, failures);
    return failures ? 1 : 0;
}
