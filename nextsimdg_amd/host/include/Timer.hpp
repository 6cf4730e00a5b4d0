// Timer.hpp -- hierarchical timers with the reference's surface (core/src/include/Timer.hpp:18-134,
// core/src/Timer.cpp:35-198): Timer::main.tick("name") descends into (or creates) a child node and starts
// it, tock() stops it and ascends; every node accumulates wall time, CPU time and an activation count;
// report() prints the tree with the share of the parent and the time per call.  ScopedTimer is the RAII
// form.  Addition for a GPU step: tock() can first drain a HIP stream (setDeviceSync), so that the wall
// time of a node includes the device work enqueued inside it -- without it the asynchronous launches
// would be charged to whichever node happens to synchronise later.
//
// In the reference the timers exist but no model code calls them; here Model and the
// model steps are instrumented and `model.timing = true` prints the report at the end of a run.
#pragma once
#include <chrono>
#include <ctime>
#include <functional>
#include <map>
#include <ostream>
#include <string>
#include <vector>

namespace Nextsim {

class Timer {
public:
    typedef std::string Key;
    Timer();
    explicit Timer(const Key& rootKey);

    void tick(const Key& timerName);
    void tock(const Key& timerName); //!< stops `timerName`, which must be the running node
    void tock();
    void reset();
    std::ostream& report(std::ostream& os) const;

    //! wall seconds / activation count of a node addressed by its path from the root, e.g. {"run", "iterate"}
    double wallSeconds(const std::vector<Key>& path) const;
    int ticks(const std::vector<Key>& path) const;

    //! called at every tock() before the clock is read (e.g. a stream synchronisation); may be empty
    void setDeviceSync(std::function<void()> sync) { m_sync = std::move(sync); }

    static Timer main;

private:
    struct Node {
        Key name;
        Node* parent = nullptr;
        std::map<Key, Node> children;
        std::vector<Key> order; // children in first-tick order
        double wall = 0, cpu = 0;
        int count = 0;
        bool running = false;
        std::chrono::steady_clock::time_point wall0;
        std::clock_t cpu0 = 0;
    };
    const Node* find(const std::vector<Key>& path) const;
    static void print(std::ostream& os, const Node& n, const std::string& prefix, double parentWall);
    Node root;
    Node* current;
    std::function<void()> m_sync;
};

//! RAII tick/tock on Timer::main (core/src/include/ScopedTimer.hpp)
class ScopedTimer {
public:
    explicit ScopedTimer(const Timer::Key& name) { Timer::main.tick(name); }
    ~ScopedTimer() { Timer::main.tock(); }
    ScopedTimer(const ScopedTimer&) = delete;
    ScopedTimer& operator=(const ScopedTimer&) = delete;
};

} // namespace Nextsim
