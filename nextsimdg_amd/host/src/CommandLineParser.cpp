#include "CommandLineParser.hpp"

namespace Nextsim {

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
