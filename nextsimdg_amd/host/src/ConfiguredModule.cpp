#include "ConfiguredModule.hpp"

#include <stdexcept>

#include "Configurator.hpp"
#include "ModuleLoader.hpp"

namespace Nextsim {

const std::string ConfiguredModule::MODULE_PREFIX = "Modules";

std::string ConfiguredModule::addPrefix(const std::string& moduleName) { return MODULE_PREFIX + "." + moduleName; }

void ConfiguredModule::parseConfigurator()
{
    ModuleLoader& loader = ModuleLoader::getLoader();
    for (const std::string& module : loader.listModules()) {
        std::string impl;
        if (!Configurator::lookup(addPrefix(module), impl))
            continue; // not mentioned: keep the current selection
        bool known = false;
        for (const std::string& name : loader.listImplementations(module))
            known = known || (name == impl);
        if (!known)
            throw std::domain_error("Invalid implementation \"" + impl + "\" of module " + module + ".");
        loader.setImplementation(module, impl);
    }
}

} // namespace Nextsim
