#include "Configurator.hpp"

#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "ModuleLoader.hpp"

namespace Nextsim {

namespace {
std::string trim(const std::string& s)
{
    const char* ws = " \t\r\n";
    const auto b = s.find_first_not_of(ws);
    if (b == std::string::npos)
        return "";
    const auto e = s.find_last_not_of(ws);
    return s.substr(b, e - b + 1);
}
} // namespace

std::vector<std::map<std::string, std::string>>& Configurator::sources()
{
    static std::vector<std::map<std::string, std::string>> s;
    return s;
}
int& Configurator::argc()
{
    static int c = 0;
    return c;
}
char**& Configurator::argv()
{
    static char** v = nullptr;
    return v;
}

std::map<std::string, std::string> Configurator::parseIni(std::istream& is)
{
    std::map<std::string, std::string> kv;
    std::string line, section;
    int lineno = 0;
    while (std::getline(is, line)) {
        ++lineno;
        const auto hash = line.find('#');
        if (hash != std::string::npos)
            line.erase(hash);
        line = trim(line);
        if (line.empty())
            continue;
        if (line.front() == '[') {
            if (line.back() != ']') {
                std::cerr << "config line " << lineno << ": unterminated section header" << std::endl;
                continue;
            }
            section = trim(line.substr(1, line.size() - 2));
            continue;
        }
        const auto eq = line.find('=');
        if (eq == std::string::npos) {
            std::cerr << "config line " << lineno << ": expected key = value" << std::endl;
            continue;
        }
        const std::string key = trim(line.substr(0, eq));
        const std::string full = section.empty() ? key : section + "." + key;
        kv.emplace(full, trim(line.substr(eq + 1))); // emplace keeps the first occurrence
    }
    return kv;
}

void Configurator::addFile(const std::string& filename)
{
    std::unique_ptr<std::istream> f(new std::ifstream(filename));
    if (!*f)
        std::cerr << "cannot open configuration file " << filename << std::endl;
    addStream(std::move(f));
}

void Configurator::addStream(std::unique_ptr<std::istream> pis) { sources().push_back(parseIni(*pis)); }

void Configurator::clearStreams() { sources().clear(); }

void Configurator::clear()
{
    clearStreams();
    setCommandLine(0, nullptr);
}

void Configurator::setCommandLine(int argc_, char* argv_[])
{
    argc() = argc_;
    argv() = argv_;
}

bool Configurator::lookup(const std::string& key, std::string& value)
{
    // command line: --key=value or --key value (unix style, unknown options ignored)
    const std::string flag = "--" + key;
    for (int i = 1; i < argc(); ++i) {
        const std::string a = argv()[i];
        if (a == flag && i + 1 < argc()) {
            value = argv()[i + 1];
            return true;
        }
        if (a.size() > flag.size() && a.compare(0, flag.size(), flag) == 0 && a[flag.size()] == '=') {
            value = a.substr(flag.size() + 1);
            return true;
        }
    }
    for (const auto& src : sources()) {
        const auto it = src.find(key);
        if (it != src.end()) {
            value = it->second;
            return true;
        }
    }
    return false;
}


// ---- ConfiguredModule
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

// ---- CommandLineParser
std::string CommandLineParser::helpText()
{
    return "neXtSIM_DG (MI355X) command line options:\n"
           "  -h [ --help ]            print help message\n"
           "  --config-file arg        specify a configuration file\n"
           "  --config-files arg...    specify a list of configuration files\n"
           "  --section.key=value      override any configuration value\n";
}

CommandLineParser::CommandLineParser(int argc, char* argv[])
{
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--help" || a == "-h") {
            m_help = true;
        } else if (a == "--config-file") {
            if (i + 1 < argc)
                m_configFilenames.push_back(argv[++i]);
        } else if (a.rfind("--config-file=", 0) == 0) {
            m_configFilenames.push_back(a.substr(14));
        } else if (a == "--config-files") { // multitoken: every following token up to the next option
            while (i + 1 < argc && std::string(argv[i + 1]).rfind("-", 0) != 0)
                m_configFilenames.push_back(argv[++i]);
        }
    }
}

} // namespace Nextsim
