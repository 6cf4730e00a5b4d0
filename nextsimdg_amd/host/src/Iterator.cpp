#include "Iterator.hpp"

#include <cstdlib>
#include <stdexcept>

namespace Nextsim {

Iterator::NullIterant Iterator::nullIterant;

Iterator::Iterator()
    : iterant(&nullIterant)
{
}
Iterator::Iterator(Iterant* it)
    : iterant(it)
{
}
void Iterator::setIterant(Iterant* it) { iterant = it; }

void Iterator::setStartStopStep(TimePoint start, TimePoint stop, Duration step)
{
    startTime = start;
    stopTime = stop;
    timestep = step;
}
void Iterator::setStartDurationStep(TimePoint start, Duration duration, Duration step) { setStartStopStep(start, start + duration, step); }

void Iterator::parseAndSet(const std::string& startStr, const std::string& stopStr, const std::string& durationStr,
    const std::string& stepStr)
{
    startTime = std::atoi(startStr.c_str());
    timestep = std::atoi(stepStr.c_str());
    stopTime = durationStr.empty() ? std::atoi(stopStr.c_str()) : startTime + std::atoi(durationStr.c_str());
}

void Iterator::run()
{
    if (timestep <= 0)
        throw std::invalid_argument("Iterator::run(): the time step must be positive");
    iterant->start(startTime);
    for (TimePoint t = startTime; t < stopTime; t += timestep)
        iterant->iterate(timestep);
    iterant->stop(stopTime);
}

} // namespace Nextsim
