// Iterator.hpp -- the time loop: start(t0); for (t = t0; t < stop; t += dt) iterate(dt); stop(stop)
// with integer seconds, as in the reference (core/src/include/Iterator.hpp:18-118,
// core/src/Iterator.cpp:35-62).
#pragma once
#include <string>

#include "IStructure.hpp"

namespace Nextsim {

class Iterator {
public:
    typedef int TimePoint;
    typedef int Duration;

    class Iterant {
    public:
        virtual ~Iterant() = default;
        virtual void init() = 0;
        virtual void start(const TimePoint& startTime) = 0;
        virtual void iterate(const Duration& dt) = 0;
        virtual void stop(const TimePoint& stopTime) = 0;
    };
    class NullIterant : public Iterant {
        void init() override { }
        void start(const TimePoint&) override { }
        void iterate(const Duration&) override { }
        void stop(const TimePoint&) override { }
    };
    static NullIterant nullIterant;

    Iterator();
    explicit Iterator(Iterant* iterant);
    void setIterant(Iterant* iterant);
    void setStartStopStep(TimePoint startTime, TimePoint stopTime, Duration timestep);
    void setStartDurationStep(TimePoint startTime, Duration duration, Duration timestep);
    //! run_length (if not empty) takes precedence over stop, as in the reference.
    void parseAndSet(const std::string& startTimeStr, const std::string& stopTimeStr, const std::string& durationStr,
        const std::string& stepStr);
    void run();

private:
    Iterant* iterant;
    TimePoint startTime = 0, stopTime = 0;
    Duration timestep = 1;
};


// IModelStep.hpp -- the batched seam of the model: one object that advances the whole structure by
// one time step (reference: core/src/include/IModelStep.hpp:16-34).  HipStep is the MI355X
// implementation; Model wires it exactly where the reference wires DevStep
// ("Change the model step calculation here", core/src/include/Model.hpp:47).
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
