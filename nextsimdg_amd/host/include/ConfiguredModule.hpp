// ConfiguredModule.hpp -- applies the [Modules] section of the configuration to the ModuleLoader
// (reference: core/src/ConfiguredModule.cpp:19-61).  Keys are "Modules.<fully qualified interface>",
// values implementation names; an unknown module key is ignored, an unknown implementation of a
// known module throws std::domain_error.
#pragma once
#include <string>

namespace Nextsim {
class ConfiguredModule {
public:
    static const std::string MODULE_PREFIX;
    static void parseConfigurator();
    static std::string addPrefix(const std::string& moduleName);
};
} // namespace Nextsim
