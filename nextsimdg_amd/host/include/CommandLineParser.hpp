// CommandLineParser.hpp -- the three command-line options of the executable, as in the reference
// (core/src/CommandLineParser.cpp:23-61): --help/-h, --config-file F, --config-files F1 F2 ...;
// the file names are returned in command-line order.  Anything else is left for Configurator.
#pragma once
#include <string>
#include <vector>

namespace Nextsim {
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
