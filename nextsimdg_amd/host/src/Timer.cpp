#include "Timer.hpp"

#include <iomanip>
#include <stdexcept>

namespace Nextsim {

Timer Timer::main("Total");

Timer::Timer()
    : Timer("")
{
}

Timer::Timer(const Key& rootKey)
{
    root.name = rootKey;
    current = &root;
    root.running = true;
    root.count = 1;
    root.wall0 = std::chrono::steady_clock::now();
    root.cpu0 = std::clock();
}

void Timer::reset()
{
    const Key name = root.name;
    root = Node();
    root.name = name;
    current = &root;
    root.running = true;
    root.count = 1;
    root.wall0 = std::chrono::steady_clock::now();
    root.cpu0 = std::clock();
}

void Timer::tick(const Key& timerName)
{
    auto it = current->children.find(timerName);
    if (it == current->children.end()) {
        it = current->children.emplace(timerName, Node()).first;
        it->second.name = timerName;
        it->second.parent = current;
        current->order.push_back(timerName);
    }
    Node& n = it->second;
    n.running = true;
    ++n.count;
    n.wall0 = std::chrono::steady_clock::now();
    n.cpu0 = std::clock();
    current = &n;
}

void Timer::tock()
{
    if (current == &root)
        return; // nothing to stop: the root runs for the life of the timer
    if (m_sync)
        m_sync();
    Node& n = *current;
    n.wall += std::chrono::duration<double>(std::chrono::steady_clock::now() - n.wall0).count();
    n.cpu += double(std::clock() - n.cpu0) / CLOCKS_PER_SEC;
    n.running = false;
    current = n.parent;
}

void Timer::tock(const Key& timerName)
{
    if (current->name != timerName)
        throw std::logic_error("Timer::tock(\"" + timerName + "\"): the running timer is \"" + current->name + "\"");
    tock();
}

const Timer::Node* Timer::find(const std::vector<Key>& path) const
{
    const Node* n = &root;
    for (const Key& k : path) {
        const auto it = n->children.find(k);
        if (it == n->children.end())
            return nullptr;
        n = &it->second;
    }
    return n;
}

double Timer::wallSeconds(const std::vector<Key>& path) const
{
    const Node* n = find(path);
    return n ? n->wall : 0.;
}

int Timer::ticks(const std::vector<Key>& path) const
{
    const Node* n = find(path);
    return n ? n->count : 0;
}

void Timer::print(std::ostream& os, const Node& n, const std::string& prefix, double parentWall)
{
    double wall = n.wall;
    if (n.running && !n.parent) // the root is still running while the report is written
        wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - n.wall0).count();
    os << prefix << n.name << ": ticks = " << n.count << " wall time " << std::fixed << std::setprecision(6) << wall << " s";
    if (parentWall > 0)
        os << " (" << std::setprecision(1) << 100. * wall / parentWall << "% of parent)";
    os << " cpu time " << std::setprecision(6) << n.cpu << " s";
    if (n.count > 1)
        os << " " << std::setprecision(3) << 1e3 * wall / n.count << " ms/tick";
    os << "\n";
    for (std::size_t i = 0; i < n.order.size(); ++i) {
        const bool last = i + 1 == n.order.size();
        std::string childPrefix = prefix;
        for (char& c : childPrefix) // continue the vertical rules of the ancestors
            if (c == '+' || c == '`')
                c = (c == '+') ? '|' : ' ';
            else if (c == '-')
                c = ' ';
        print(os, n.children.at(n.order[i]), childPrefix + (last ? "`- " : "+- "), wall);
    }
}

std::ostream& Timer::report(std::ostream& os) const
{
    print(os, root, "", 0.);
    return os;
}

} // namespace Nextsim
