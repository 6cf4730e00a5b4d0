// Configurator.hpp -- configuration sources of the host layer.
//
// Keeps the surface and the precedence rules of the reference's Configurator
// (core/src/include/Configurator.hpp:21-121, core/src/Configurator.cpp:18-60): a process-wide ordered
// list of INI streams plus the raw command line; for any key the FIRST source that defines it wins,
// the command line being consulted before the streams, so `--section.key=value` overrides files and
// an earlier file overrides a later one.  Re-designed without Boost: values are looked up by their
// fully qualified key "section.key" in sources parsed once when they are added.
#pragma once
#include <istream>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace Nextsim {

class Configurator {
public:
    //! Adds a configuration file (INI syntax) to the end of the source list.
    static void addFile(const std::string& filename);
    template <typename C> static void addFiles(const C& container)
    {
        for (const auto& f : container)
            addFile(f);
    }
    //! Adds an already open stream; it is read to its end immediately.
    static void addStream(std::unique_ptr<std::istream> pis);
    static void clearStreams();
    //! Forgets every stream and the command line.
    static void clear();
    //! argv is borrowed, not copied (as in the reference); argv[0] is the program name.
    static void setCommandLine(int argc, char* argv[]);

    //! Looks a key up: command line first, then the streams in the order they were added.
    //! Returns false if no source defines it.
    static bool lookup(const std::string& key, std::string& value);

    //! Parses INI text into key -> value (first occurrence kept); syntax errors are reported on
    //! stderr and skipped, mirroring "echo the exception, but carry on" (Configurator.cpp:49-52).
    static std::map<std::string, std::string> parseIni(std::istream& is);

private:
    static std::vector<std::map<std::string, std::string>>& sources();
    static int& argc();
    static char**& argv();
};


// ConfiguredModule.hpp -- applies the [Modules] section of the configuration to the ModuleLoader
// (reference: core/src/ConfiguredModule.cpp:19-61).  Keys are "Modules.<fully qualified interface>",
// values implementation names; an unknown module key is ignored, an unknown implementation of a
// known module throws std::domain_error.
class ConfiguredModule {
public:
    static const std::string MODULE_PREFIX;
    static void parseConfigurator();
    static std::string addPrefix(const std::string& moduleName);
};

// CommandLineParser.hpp -- the three command-line options of the executable, as in the reference
// (core/src/CommandLineParser.cpp:23-61): --help/-h, --config-file F, --config-files F1 F2 ...;
// the file names are returned in command-line order.  Anything else is left for Configurator.
class CommandLineParser {
public:
    CommandLineParser(int argc, char* argv[]);
    std::vector<std::string> getConfigFileNames() const { return m_configFilenames; }
    bool helpRequested() const { return m_help; }
    static std::string helpText();

private:
    std::vector<std::string> m_configFilenames;
    bool m_help = false;
};

} // namespace Nextsim
