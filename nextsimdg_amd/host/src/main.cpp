// main.cpp -- `nextsim_amd`: same start-up sequence as the reference's main (core/src/main.cpp:14-37):
// command line -> config files -> module defaults -> [Modules] overrides -> Model::configure -> run.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <iostream>

#include "Configurator.hpp"
#include "DynamicsStep.hpp"
#include "Model.hpp"
#include "ModuleLoader.hpp"

int main(int argc, char* argv[])
{
    using namespace Nextsim;
    Configurator::setCommandLine(argc, argv);
    CommandLineParser cmdLine(argc, argv);
    if (cmdLine.helpRequested()) {
        std::cerr << CommandLineParser::helpText() << std::endl;
        return 0;
    }
    Configurator::addFiles(cmdLine.getConfigFileNames());
    try {
        ModuleLoader::getLoader().setAllDefaults();
        ConfiguredModule::parseConfigurator();
        const bool timing = Configured<Model>::getConfiguration("model.timing", false);
        Model model;
        model.configure();
        if (timing) // charge the asynchronous device work to the node that enqueued it
            Timer::main.setDeviceSync([] { (void)hipDeviceSynchronize(); });
        model.run();
        // one line per run with the state of element 0 (all elements are identical in run/dev1.cfg)
        const FieldStore& f = model.structure().fields();
        std::printf("elements=%zu launches=%ld hice=%.17g cice=%.17g hsnow=%.17g tice0=%.17g sst=%.17g\n", f.n, model.step().launches(),
            f.hice[0], f.cice[0], f.hsnow[0], f.tice[0], f.sst[0]);
        if (auto* dyn = dynamic_cast<DynamicsStep*>(&model.step()))
            std::printf("dynamics umax=%.17g sumH=%.17g sumA=%.17g\n", dyn->maxSpeed(), dyn->sumH(), dyn->sumA());
        if (timing)
            Timer::main.report(std::cout);
    } catch (const std::exception& e) {
        std::cerr << "nextsim_amd: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
