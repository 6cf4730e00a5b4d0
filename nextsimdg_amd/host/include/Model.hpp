// Model.hpp -- owns the iterator, the model step and the structure; configure() + run(), as the
// reference's Model (core/src/include/Model.hpp:25-60, core/src/Model.cpp:31-88).
#pragma once
#include <memory>
#include <string>

#include <iostream>

#include "Configured.hpp"
#include "ModuleLoader.hpp"
#include "HipStep.hpp"
#include "Iterator.hpp"
#include "RectGrid.hpp"
#include "Timer.hpp"

namespace Nextsim {

class Model : public Configured<Model> {
public:
    Model();
    ~Model() override; //!< writes the final restart file; errors are swallowed (Model.cpp:40-53)
    void configure() override;
    void run();
    void writeRestartFile();
    IStructure& structure() { return *dataStructure; }
    IModelStep& step() { return *modelStep; }

    enum { RESTARTFILE_KEY, STARTTIME_KEY, STOPTIME_KEY, RUNLENGTH_KEY, TIMESTEP_KEY, STRUCTURE_KEY, FINALFILE_KEY };

private:
    Iterator iterator;
    // "Change the model step calculation here" (core/src/include/Model.hpp:47): here the step is itself a
    // plugin, [Modules] Nextsim::IModelStep = Nextsim::HipStep (default) | Nextsim::DynamicsStep
    std::unique_ptr<IModelStep> modelStep;
    std::shared_ptr<IStructure> dataStructure;
    std::string initialFileName, finalFileName;
    bool configured = false;
};


// ---- implementation (header-only: the class is a few lines of glue)
template <>
inline const std::map<int, std::string> Configured<Model>::keyMap = {
    { Model::RESTARTFILE_KEY, "model.init_file" }, // core/src/Model.cpp:23-29
    { Model::STARTTIME_KEY, "model.start" },
    { Model::STOPTIME_KEY, "model.stop" },
    { Model::RUNLENGTH_KEY, "model.run_length" },
    { Model::TIMESTEP_KEY, "model.time_step" },
    { Model::STRUCTURE_KEY, "model.structure" }, // new: structure type when there is no init file
    { Model::FINALFILE_KEY, "model.final_file" }, // new: defaults to the reference's fixed "restart.nc" role
};

inline Model::Model()
    : modelStep(ModuleLoader::getLoader().getInstance<IModelStep>())
{
    iterator.setIterant(modelStep.get());
    finalFileName = "restart.nc"; // core/src/Model.cpp:37; written as HDF5 (Hdf5Subset.hpp)
}

inline Model::~Model()
{
    try {
        if (dataStructure && configured) // a run that failed in configure() has no state worth a restart file
            writeRestartFile();
    } catch (std::exception& e) {
        // swallowed, as in the reference destructor
    }
}

inline void Model::configure()
{
    ScopedTimer t("configure");
    const std::string startTimeStr = getConfiguration(keyMap.at(STARTTIME_KEY), std::string("0"));
    const std::string stopTimeStr = getConfiguration(keyMap.at(STOPTIME_KEY), std::string("1"));
    const std::string durationStr = getConfiguration(keyMap.at(RUNLENGTH_KEY), std::string(""));
    const std::string stepStr = getConfiguration(keyMap.at(TIMESTEP_KEY), std::string("1"));
    iterator.parseAndSet(startTimeStr, stopTimeStr, durationStr, stepStr);

    initialFileName = getConfiguration(keyMap.at(RESTARTFILE_KEY), std::string(""));
    finalFileName = getConfiguration(keyMap.at(FINALFILE_KEY), finalFileName);
    modelStep->setInitFile(initialFileName);
    if (!initialFileName.empty()) {
        // as the reference (core/src/Model.cpp:59-62): the named restart file MUST be readable; a missing, truncated
        // or unsupported file stops the run (std::runtime_error / Hdf5Error / std::invalid_argument) instead of
        // silently starting from something else
        dataStructure = StructureFactory::generateFromFile(initialFileName);
        dataStructure->init(initialFileName);
        std::cout << "Initial state read from " << initialFileName << " (structure " << dataStructure->structureType() << ", "
                  << dataStructure->nx() << " x " << dataStructure->ny() << ")" << std::endl;
    } else { // no init file configured: constants from the init.* keys (defaults = run/dev_res.py:10-20 of the reference)
        dataStructure = StructureFactory::generate(getConfiguration(keyMap.at(STRUCTURE_KEY), std::string("devgrid")));
        dataStructure->init("");
        std::cout << "No model.init_file: constant initial state from the [init] keys" << std::endl;
    }
    modelStep->setInitialData(*dataStructure);
    modelStep->init();
    DummyExternalData::setAll(*dataStructure); // core/src/Model.cpp:76
    configured = true;
}

inline void Model::run()
{
    ScopedTimer t("run");
    iterator.run();
}

inline void Model::writeRestartFile()
{
    modelStep->writeRestartFile(finalFileName);
    std::cout << "Restart file written to " << finalFileName << std::endl;
}

} // namespace Nextsim
