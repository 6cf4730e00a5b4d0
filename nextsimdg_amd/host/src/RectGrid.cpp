#include "RectGrid.hpp"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "Hdf5Subset.hpp"
#include "ModuleLoader.hpp"

namespace Nextsim {

void FieldStore::resize(std::size_t nElements, int nIceLayers)
{
    n = nElements;
    nLayers = nIceLayers;
    for (auto* v : { &hice, &cice, &hsnow, &sst, &sss, &tair, &tdew, &slp, &mixrat, &qsw, &qlw, &mld, &snowfall, &wind, &newice })
        v->assign(n, 0.);
    tice.assign(n * (std::size_t)nLayers, 0.);
    dyn.clear();
}

ElementData& ElementData::operator=(const PrognosticGenerator& g)
{
    iceThickness() = g.m_hice;
    iceConcentration() = g.m_cice;
    snowThickness() = g.m_hsnow;
    seaSurfaceTemperature() = g.m_sst;
    seaSurfaceSalinity() = g.m_sss;
    for (int l = 0; l < s->nLayers; ++l) // missing layers repeat the last given one (PrognosticData.cpp:82-94)
        iceTemperature(l) = g.m_tice.empty() ? 0. : g.m_tice[std::min<std::size_t>(l, g.m_tice.size() - 1)];
    return *this;
}

void DummyExternalData::setAll(IStructure& is)
{
    for (is.cursor = 0; is.cursor; ++is.cursor) {
        is.cursor->airTemperature() = -1;
        is.cursor->dewPoint2m() = -4;
        is.cursor->airPressure() = 1e5;
        is.cursor->mixingRatio() = -1.;
        is.cursor->incomingShortwave() = 0;
        is.cursor->incomingLongwave() = 311;
        is.cursor->mixedLayerDepth() = 10;
        is.cursor->snowfall() = 0;
    }
}

RectGrid::RectGrid()
    : m_nx(0)
    , m_ny(0)
    , m_current(&m_store, 0)
{
    resize(10, 10, 1);
}

void RectGrid::resize(int nx, int ny, int nLayers)
{
    if (nx <= 0 || ny <= 0 || nLayers <= 0)
        throw std::invalid_argument("RectGrid: nx, ny and nLayers must be positive");
    m_nx = nx;
    m_ny = ny;
    m_store.resize((std::size_t)nx * ny, nLayers);
    resetCursor();
}

template <> const std::map<int, std::string> Configured<RectGrid>::keyMap = { { 0, "rectgrid.nx" }, { 1, "rectgrid.ny" }, { 2, "rectgrid.nLayers" } };

void RectGrid::configure()
{
    const int nx = getConfiguration(keyMap.at(0), m_nx);
    const int ny = getConfiguration(keyMap.at(1), m_ny);
    const int nl = getConfiguration(keyMap.at(2), m_store.nLayers);
    if (nx != m_nx || ny != m_ny || nl != m_store.nLayers)
        resize(nx, ny, nl);
}

int RectGrid::resetCursor()
{
    m_cursor = 0;
    m_current = ElementData(&m_store, 0);
    return 0;
}

void RectGrid::incrCursor()
{
    ++m_cursor;
    m_current = ElementData(&m_store, m_cursor);
}

namespace {
const char* MAGIC = "NSDG-RESTART 1";
bool readHeader(std::istream& f, std::string& type, int& nx, int& ny, int& nl, bool* dynamics = nullptr)
{
    std::string line;
    if (!std::getline(f, line) || line != MAGIC)
        return false;
    type.clear();
    nx = ny = nl = 0;
    if (dynamics)
        *dynamics = false;
    while (std::getline(f, line) && line != "END-HEADER") {
        const auto eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        const std::string k = line.substr(0, eq), v = line.substr(eq + 1);
        if (k == "structure.type")
            type = v;
        else if (k == "data.x")
            nx = std::stoi(v);
        else if (k == "data.y")
            ny = std::stoi(v);
        else if (k == "data.nLayers")
            nl = std::stoi(v);
        else if (k == "data.dynamics" && dynamics)
            *dynamics = v == "1";
    }
    return nx > 0 && ny > 0 && nl > 0;
}
} // namespace

std::string RectGrid::typeInFile(const std::string& filePath, bool throwOnError)
{
    if (Hdf5File::isHdf5(filePath)) { // the reference's NetCDF-4 layout: group "structure", attribute "type" (StructureFactory.cpp:46-58)
        try {
            return Hdf5File(filePath).stringAttribute("/" + metadataNodeName(), typeNodeName());
        } catch (const Hdf5Error&) {
            if (throwOnError)
                throw; // a corrupt or unsupported file must stop the run with the reader's message
            return std::string();
        }
    }
    std::ifstream f(filePath, std::ios::binary);
    std::string type;
    int a, b, c;
    return (f && readHeader(f, type, a, b, c)) ? type : std::string();
}

namespace {
bool wantsHdf5(const std::string& path)
{
    for (const char* ext : { ".nc", ".h5", ".hdf5" }) {
        const std::size_t n = std::strlen(ext);
        if (path.size() > n && path.compare(path.size() - n, n, ext) == 0)
            return true;
    }
    return false;
}
const char* const PLANE_NAMES[5] = { "hice", "cice", "hsnow", "sst", "sss" }; // core/src/DevGridIO.cpp:35-40
} // namespace

void RectGrid::dump(const std::string& filePath) const
{
    // tice is (x, y, nLayers) in the file, layer-major planes in memory
    std::vector<double> t(m_store.tice.size());
    for (std::size_t e = 0; e < m_store.n; ++e)
        for (int l = 0; l < m_store.nLayers; ++l)
            t[e * m_store.nLayers + l] = m_store.tice[(std::size_t)l * m_store.n + e];
    if (wantsHdf5(filePath)) { // groups, names and dimensions of core/src/DevGridIO.cpp:150-201
        Hdf5Writer w;
        w.group("/" + metadataNodeName());
        w.stringAttribute("/" + metadataNodeName(), typeNodeName(), structureType());
        const std::string data = "/" + dataNodeName();
        w.group(data);
        // the named dimensions of core/src/DevGridIO.cpp:169-172 (NetCDF-4 stores them as HDF5 dimension scales)
        w.dimension(data, "x", (std::uint64_t)m_nx, 0);
        w.dimension(data, "y", (std::uint64_t)m_ny, 1);
        w.dimension(data, "nLayers", (std::uint64_t)m_store.nLayers, 2);
        const std::vector<double>* planes[5] = { &m_store.hice, &m_store.cice, &m_store.hsnow, &m_store.sst, &m_store.sss };
        for (int k = 0; k < 5; ++k) {
            w.dataset(data + "/" + PLANE_NAMES[k], { (std::uint64_t)m_nx, (std::uint64_t)m_ny }, *planes[k]);
            w.attachDimensions(data + "/" + PLANE_NAMES[k], { data + "/x", data + "/y" }); // DevGridIO.cpp:174-190
        }
        w.dataset(data + "/tice", { (std::uint64_t)m_nx, (std::uint64_t)m_ny, (std::uint64_t)m_store.nLayers }, t);
        w.attachDimensions(data + "/tice", { data + "/x", data + "/y", data + "/nLayers" }); // DevGridIO.cpp:192-201
        if (m_store.dyn.present) { // the state of the dynamics: further variables of the same group (RectGrid.hpp)
            const DynamicsState& d = m_store.dyn;
            const std::uint64_t X = (std::uint64_t)m_nx, Y = (std::uint64_t)m_ny;
            w.dimension(data, "dg2", 5, 3);
            w.dimension(data, "stress8", 8, 4);
            w.dimension(data, "xnode", 2 * X + 1, 5);
            w.dimension(data, "ynode", 2 * Y + 1, 6);
            for (const auto& v : { std::make_pair("hice_dg", &d.hdg), std::make_pair("cice_dg", &d.adg) }) {
                w.dataset(data + "/" + v.first, { 5, X, Y }, *v.second);
                w.attachDimensions(data + "/" + v.first, { data + "/dg2", data + "/x", data + "/y" });
            }
            for (const auto& v : { std::make_pair("u", &d.u), std::make_pair("v", &d.v) }) {
                w.dataset(data + "/" + v.first, { 2 * X + 1, 2 * Y + 1 }, *v.second);
                w.attachDimensions(data + "/" + v.first, { data + "/xnode", data + "/ynode" });
            }
            for (const auto& v : { std::make_pair("s11", &d.s11), std::make_pair("s12", &d.s12), std::make_pair("s22", &d.s22) }) {
                w.dataset(data + "/" + v.first, { 8, X, Y }, *v.second);
                w.attachDimensions(data + "/" + v.first, { data + "/stress8", data + "/x", data + "/y" });
            }
            w.dataset(data + "/newice", { X, Y }, m_store.newice);
            w.attachDimensions(data + "/newice", { data + "/x", data + "/y" });
        }
        w.write(filePath);
        return;
    }
    std::ofstream f(filePath, std::ios::binary);
    if (!f)
        throw std::runtime_error("cannot write restart file " + filePath);
    f << MAGIC << "\n"
      << metadataNodeName() << "." << typeNodeName() << "=" << structureType() << "\n"
      << dataNodeName() << ".x=" << m_nx << "\n"
      << dataNodeName() << ".y=" << m_ny << "\n"
      << dataNodeName() << ".nLayers=" << m_store.nLayers << "\n"
      << (m_store.dyn.present ? "data.dynamics=1\nvariables=hice,cice,hsnow,sst,sss,tice,hice_dg,cice_dg,u,v,s11,s12,s22,newice\nEND-HEADER\n"
                                : "variables=hice,cice,hsnow,sst,sss,tice\nEND-HEADER\n");
    for (const auto* v : { &m_store.hice, &m_store.cice, &m_store.hsnow, &m_store.sst, &m_store.sss })
        f.write(reinterpret_cast<const char*>(v->data()), (std::streamsize)(v->size() * sizeof(double)));
    f.write(reinterpret_cast<const char*>(t.data()), (std::streamsize)(t.size() * sizeof(double)));
    if (m_store.dyn.present) {
        const DynamicsState& d = m_store.dyn;
        for (const auto* v : { &d.hdg, &d.adg, &d.u, &d.v, &d.s11, &d.s12, &d.s22, &m_store.newice })
            f.write(reinterpret_cast<const char*>(v->data()), (std::streamsize)(v->size() * sizeof(double)));
    }
}

void RectGrid::init(const std::string& filePath)
{
    tryConfigure(*this);
    if (filePath.empty()) { // constants, defaults = run/dev_res.py:10-20 of the reference
        typedef Configured<RectGrid> C;
        const double hice = C::getConfiguration("init.hice", 0.1), cice = C::getConfiguration("init.cice", 0.5);
        const double hsnow = C::getConfiguration("init.hsnow", 0.0), sst = C::getConfiguration("init.sst", -1.0);
        const double sss = C::getConfiguration("init.sss", 32.0), tice = C::getConfiguration("init.tice", -1.0);
        for (cursor = 0; cursor; ++cursor)
            *cursor = PrognosticGenerator().hice(hice).cice(cice).hsnow(hsnow).sst(sst).sss(sss).tice({ tice });
        return;
    }
    if (Hdf5File::isHdf5(filePath)) {
        // NetCDF-4 restart file of the reference (core/src/DevGridIO.cpp:93-147): group "data", variables
        // hice, cice, hsnow, sst, sss (x, y) and tice (x, y, nLayers); the number of layers is taken from
        // the third dimension of tice (DevGridIO.cpp:97-99), element (i, j) has the linear index i*ny + j
        const Hdf5File h(filePath);
        const std::string g = "/" + dataNodeName() + "/";
        const std::vector<std::uint64_t> d = h.dims(g + "tice");
        if (d.size() != 3)
            throw std::runtime_error("restart file " + filePath + ": tice must have the dimensions (x, y, nLayers)");
        if (structureType() == "devgrid" && (d[0] != 10 || d[1] != 10))
            throw std::runtime_error("devgrid restart files must be 10x10");
        resize((int)d[0], (int)d[1], (int)d[2]);
        std::vector<double>* planes[5] = { &m_store.hice, &m_store.cice, &m_store.hsnow, &m_store.sst, &m_store.sss };
        for (int k = 0; k < 5; ++k) {
            std::vector<double> v = h.readDoubles(g + PLANE_NAMES[k]);
            if (v.size() != m_store.n)
                throw std::runtime_error("restart file " + filePath + ": " + PLANE_NAMES[k] + " does not have x*y elements");
            planes[k]->swap(v);
        }
        const std::vector<double> t = h.readDoubles(g + "tice");
        for (std::size_t e = 0; e < m_store.n; ++e)
            for (int l = 0; l < m_store.nLayers; ++l)
                m_store.tice[(std::size_t)l * m_store.n + e] = t[e * m_store.nLayers + l];
        if (h.exists(g + "s11")) { // the state of a dynamics run (all of it or none of it)
            DynamicsState& dy = m_store.dyn;
            dy.resize((std::size_t)m_nx, (std::size_t)m_ny);
            for (const auto& v : { std::make_pair("hice_dg", &dy.hdg), std::make_pair("cice_dg", &dy.adg), std::make_pair("u", &dy.u), std::make_pair("v", &dy.v),
                     std::make_pair("s11", &dy.s11), std::make_pair("s12", &dy.s12), std::make_pair("s22", &dy.s22), std::make_pair("newice", &m_store.newice) }) {
                std::vector<double> a = h.readDoubles(g + v.first);
                if (a.size() != v.second->size())
                    throw std::runtime_error("restart file " + filePath + ": " + v.first + " does not have the size the grid asks for");
                v.second->swap(a);
            }
            dy.present = true;
        }
        resetCursor();
        return;
    }
    std::ifstream f(filePath, std::ios::binary);
    std::string type;
    int nx, ny, nl;
    bool dynamics = false;
    if (!f || !readHeader(f, type, nx, ny, nl, &dynamics))
        throw std::runtime_error("cannot read restart file " + filePath);
    if (structureType() == "devgrid" && (nx != 10 || ny != 10))
        throw std::runtime_error("devgrid restart files must be 10x10");
    resize(nx, ny, nl);
    for (auto* v : { &m_store.hice, &m_store.cice, &m_store.hsnow, &m_store.sst, &m_store.sss })
        f.read(reinterpret_cast<char*>(v->data()), (std::streamsize)(v->size() * sizeof(double)));
    std::vector<double> t(m_store.tice.size());
    f.read(reinterpret_cast<char*>(t.data()), (std::streamsize)(t.size() * sizeof(double)));
    if (dynamics) {
        DynamicsState& d = m_store.dyn;
        d.resize((std::size_t)nx, (std::size_t)ny);
        for (auto* v : { &d.hdg, &d.adg, &d.u, &d.v, &d.s11, &d.s12, &d.s22, &m_store.newice })
            f.read(reinterpret_cast<char*>(v->data()), (std::streamsize)(v->size() * sizeof(double)));
        d.present = true;
    }
    if (!f)
        throw std::runtime_error("restart file " + filePath + " is truncated");
    for (std::size_t e = 0; e < m_store.n; ++e)
        for (int l = 0; l < nl; ++l)
            m_store.tice[(std::size_t)l * m_store.n + e] = t[e * nl + l];
}

// registration order = default order: devgrid first, as in core/src/modules/modules.json:9-14
NSDG_REGISTER_MODULE(IStructure, DevGrid, "Nextsim::IStructure", "Nextsim::DevGrid");
NSDG_REGISTER_MODULE(IStructure, RectGrid, "Nextsim::IStructure", "Nextsim::RectGrid");

std::shared_ptr<IStructure> StructureFactory::generate(const std::string& structureName)
{
    ModuleLoader& loader = ModuleLoader::getLoader();
    for (const std::string& impl : loader.listImplementations("Nextsim::IStructure")) {
        loader.setImplementation("Nextsim::IStructure", impl);
        std::shared_ptr<IStructure> s(loader.getInstance<IStructure>().release());
        if (s->structureTypeCheck(structureName))
            return s;
    }
    throw std::invalid_argument("StructureFactory::generate: no structure type named " + structureName);
}

std::shared_ptr<IStructure> StructureFactory::generateFromFile(const std::string& filePath)
{
    {
        std::ifstream probe(filePath, std::ios::binary);
        if (!probe)
            throw std::runtime_error("StructureFactory::generateFromFile: cannot open " + filePath);
    }
    const std::string type = RectGrid::typeInFile(filePath, true);
    if (type.empty())
        throw std::invalid_argument("StructureFactory::generateFromFile: no structure type in " + filePath);
    return generate(type);
}

} // namespace Nextsim
