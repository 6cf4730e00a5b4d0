// Model.hpp -- owns the iterator, the model step and the structure; configure() + run(), as the
// reference's Model (core/src/include/Model.hpp:25-60, core/src/Model.cpp:31-88).
#pragma once
#include <memory>
#include <string>

#include "Configured.hpp"
#include "HipStep.hpp"
#include "Iterator.hpp"
#include "RectGrid.hpp"

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
};

} // namespace Nextsim
