// IModelStep.hpp -- the batched seam of the model: one object that advances the whole structure by
// one time step (reference: core/src/include/IModelStep.hpp:16-34).  HipStep is the MI355X
// implementation; Model wires it exactly where the reference wires DevStep
// ("Change the model step calculation here", core/src/include/Model.hpp:47).
#pragma once
#include <string>

#include "IStructure.hpp"
#include "Iterator.hpp"

namespace Nextsim {

class IModelStep : public Iterator::Iterant {
public:
    virtual ~IModelStep() = default;
    void setInitFile(const std::string& filePath) { initialRestartFilePath = filePath; }
    virtual void writeRestartFile(const std::string& filePath) = 0;
    virtual void setInitialData(IStructure& dataStructure) = 0;
    //! Number of model steps executed on the device so far (new; for reports and tests).
    virtual long launches() const { return 0; }

protected:
    std::string initialRestartFilePath;
};

} // namespace Nextsim
