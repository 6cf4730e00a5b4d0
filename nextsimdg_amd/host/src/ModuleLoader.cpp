#include "ModuleLoader.hpp"

ModuleLoader& ModuleLoader::getLoader()
{
    static ModuleLoader instance;
    return instance;
}

void ModuleLoader::registerImplementation(const std::string& module, std::type_index iface, const std::string& impl,
    std::function<void*()> create, std::function<void*()> shared)
{
    m_modules.insert(module);
    m_names[module].push_back(impl);
    m_ifaceOf.emplace(module, iface);
    Entry& e = m_entries[iface];
    e.module = module;
    e.impls[impl] = Impl { create, shared };
    if (e.selected.empty())
        e.selected = impl; // first registered = default
}

const ModuleLoader::Entry& ModuleLoader::entry(const std::type_info& ti) const
{
    const auto it = m_entries.find(std::type_index(ti));
    if (it == m_entries.end())
        throw std::out_of_range(std::string("ModuleLoader: no module registered for interface type ") + ti.name());
    return it->second;
}

void ModuleLoader::init(const VariablesMap& map)
{
    for (const auto& kv : map)
        setImplementation(kv.first, kv.second);
}

void ModuleLoader::setImplementation(const std::string& module, const std::string& impl)
{
    const auto it = m_ifaceOf.find(module);
    if (it == m_ifaceOf.end())
        return; // unknown modules are ignored, as in the generated assignment chain of the reference
    Entry& e = m_entries.at(it->second);
    if (e.impls.find(impl) == e.impls.end())
        throw std::invalid_argument("ModuleLoader::init(): Module " + module + " does not have an implementation named " + impl);
    e.selected = impl;
}

void ModuleLoader::setDefault(const std::string& module) { setImplementation(module, listImplementations(module).front()); }

void ModuleLoader::setAllDefaults()
{
    for (const std::string& module : listModules())
        setDefault(module);
}
