// ModuleLoader.hpp -- run-time choice of the implementation of each plugin interface.
//
// Same surface and semantics as the reference's ModuleLoader (core/src/include/ModuleLoader.hpp:19-109,
// core/src/ModuleLoader.cpp:16-61): a singleton keyed by the fully qualified interface name
// ("Nextsim::IIceAlbedo") -> implementation name ("Nextsim::CCSMIceAlbedo"); getImplementation<T>()
// returns ONE shared static instance of the selected class, getInstance<T>() a fresh unique_ptr; the
// default is the first implementation registered; unknown implementation -> std::invalid_argument,
// unknown module in listImplementations -> std::out_of_range.
//
// Re-design: the reference generates the registry at build time from modules.json with a Python
// script (core/src/modules/moduleloader_builder.py:38-125); here implementations register
// themselves from static initialisers (NSDG_REGISTER_MODULE), so a new component only has to be
// linked in.  Registration order within a translation unit defines the default, as the JSON order
// does in the reference.
#pragma once
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <typeindex>
#include <typeinfo>

class ModuleLoader {
public:
    static ModuleLoader& getLoader();

    typedef std::map<std::string, std::string> VariablesMap;
    //! No-op kept for source compatibility (registration happens at static-initialisation time).
    void init() { }
    //! Selects implementations from a module -> implementation map.
    void init(const VariablesMap& map);

    const std::set<std::string>& listModules() const { return m_modules; }
    const std::list<std::string>& listImplementations(const std::string& module) const
    {
        return m_names.at(module);
    }

    template <class T> std::unique_ptr<T> getInstance() const
    {
        const Entry& e = entry(typeid(T));
        return std::unique_ptr<T>(static_cast<T*>(e.impls.at(e.selected).create()));
    }
    template <class T> T& getImplementation()
    {
        const Entry& e = entry(typeid(T));
        return *static_cast<T*>(e.impls.at(e.selected).shared());
    }

    void setImplementation(const std::string& module, const std::string& impl);
    void setDefault(const std::string& module);
    void setAllDefaults();

    //! Used by NSDG_REGISTER_MODULE.  create() returns a new object, shared() the static instance,
    //! both as pointers to the INTERFACE type.
    void registerImplementation(const std::string& module, std::type_index iface, const std::string& impl,
        std::function<void*()> create, std::function<void*()> shared);

    ModuleLoader(const ModuleLoader&) = delete;
    void operator=(const ModuleLoader&) = delete;

private:
    ModuleLoader() = default;
    struct Impl {
        std::function<void*()> create, shared;
    };
    struct Entry {
        std::string module;
        std::map<std::string, Impl> impls;
        std::string selected;
    };
    const Entry& entry(const std::type_info& ti) const;

    std::set<std::string> m_modules;
    std::map<std::string, std::list<std::string>> m_names;
    std::map<std::string, std::type_index> m_ifaceOf;
    std::map<std::type_index, Entry> m_entries;
};

namespace nsdg_host_detail {
template <class Iface, class ImplT> struct ModuleRegistrar {
    ModuleRegistrar(const char* module, const char* impl)
    {
        ModuleLoader::getLoader().registerImplementation(
            module, typeid(Iface), impl, []() -> void* { return static_cast<Iface*>(new ImplT()); },
            []() -> void* {
                static ImplT instance;
                return static_cast<Iface*>(&instance);
            });
    }
};
} // namespace nsdg_host_detail

#define NSDG_CONCAT2(a, b) a##b
#define NSDG_CONCAT(a, b) NSDG_CONCAT2(a, b)
//! Registers ImplT as an implementation of Iface under the given fully qualified names.
#define NSDG_REGISTER_MODULE(Iface, ImplT, moduleName, implName) \
    static nsdg_host_detail::ModuleRegistrar<Iface, ImplT> NSDG_CONCAT(nsdg_registrar_, __COUNTER__)(moduleName, implName)
