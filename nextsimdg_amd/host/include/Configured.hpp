// Configured.hpp -- typed configuration lookup for classes, same surface as the reference's
// Configured<C> (core/src/include/Configured.hpp:21-127): a class derives from Configured<itself>,
// lists its keys in the static keyMap and pulls values with getConfiguration(key, default) inside
// configure(); tryConfigure(x) configures x if (and only if) it derives from ConfiguredBase.
#pragma once
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "Configurator.hpp"

namespace Nextsim {

class ConfiguredBase {
public:
    virtual ~ConfiguredBase() = default;
    virtual void configure() = 0;
};

namespace detail {
template <typename T> inline T fromString(const std::string& s, const std::string& key)
{
    std::istringstream is(s);
    T v;
    if (!(is >> v))
        throw std::invalid_argument("configuration value \"" + s + "\" of " + key + " has the wrong type");
    return v;
}
template <> inline std::string fromString<std::string>(const std::string& s, const std::string&) { return s; }
template <> inline bool fromString<bool>(const std::string& s, const std::string& key)
{
    std::string l;
    for (char c : s)
        l += (char)std::tolower((unsigned char)c);
    if (l == "true" || l == "1" || l == "yes" || l == "on")
        return true;
    if (l == "false" || l == "0" || l == "no" || l == "off")
        return false;
    throw std::invalid_argument("configuration value \"" + s + "\" of " + key + " is not a boolean");
}
} // namespace detail

template <typename C> class Configured : public ConfiguredBase {
public:
    virtual ~Configured() = default;
    virtual void configure() = 0;

    //! The value of `name` from the command line / configuration streams, or the default.
    template <typename T> static T getConfiguration(const std::string& name, const T& defaultValue)
    {
        std::string raw;
        if (!Configurator::lookup(name, raw))
            return defaultValue;
        return detail::fromString<T>(raw, name);
    }

    //! Configures the object if its (dynamic) type derives from ConfiguredBase; otherwise a no-op.
    template <typename T> static void tryConfigure(T& ref) { tryConfigure(&ref); }
    template <typename T> static void tryConfigure(T* ptr)
    {
        if constexpr (std::is_base_of_v<ConfiguredBase, T>) {
            if (ptr)
                ptr->configure();
        } else if constexpr (std::is_polymorphic_v<T>) {
            if (auto* c = dynamic_cast<ConfiguredBase*>(ptr))
                c->configure();
        }
    }

    //! Per-class map from an enum of keys to their configuration names.
    static const std::map<int, std::string> keyMap;
};

template <typename T> void tryConfigure(T& t) { Configured<int>::tryConfigure(t); }
template <typename T> void tryConfigure(T* p) { Configured<int>::tryConfigure(p); }

} // namespace Nextsim
