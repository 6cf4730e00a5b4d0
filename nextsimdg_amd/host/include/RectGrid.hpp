// RectGrid.hpp -- size-configurable rectangular structure ("rectgrid") and the reference-compatible
// fixed 10x10 "devgrid" (core/src/modules/DevGrid.cpp:16-48, DevGrid.hpp:24-73).
//
// Restart files: the reference reads and writes NetCDF-4 (core/src/DevGridIO.cpp:65-210).  Neither netCDF
// nor the HDF5 development files exist in this image, so init() reads such files -- the reference's own
// run/dev1.res.nc included -- with the dependency-free HDF5-subset reader of Hdf5Subset.hpp, and dump()
// writes HDF5 with the same groups / variable names / dimensions when the path ends in .nc, .h5 or .hdf5.
// Any other path uses a raw sidecar with the same logical content: a text header
//     NSDG-RESTART 1 / structure.type=<name> / data.x=<nx> / data.y=<ny> / data.nLayers=<n>
// followed by the float64 variables hice, cice, hsnow, sst, sss (x, y) and tice (x, y, nLayers) in
// x-major order, little endian.
//
// The state of a DYNAMICS run (FieldStore::dyn, round 6) travels in both formats as further variables of group `data` -- the
// reference writes every prognostic field it has (core/src/DevGridIO.cpp:169-201), its snapshot has no dynamics, a reader that does
// not know these names never asks for them:
//     hice_dg, cice_dg (dg2 = 5, x, y)    DG2 coefficients 1..5 of the advected thickness / concentration (coefficient 0 = hice / cice)
//     u, v             (xnode = 2x+1, ynode = 2y+1)    CG2 nodal velocity
//     s11, s12, s22    (stress8 = 8, x, y)             stress coefficients of the 8-function DG space
//     newice           (x, y)              the column model's persistent new-ice volume (NextsimPhysics::m_newice)
// (sidecar: header line data.dynamics=1, the arrays in this order after tice).  A file without them starts the dynamics from rest.
#pragma once
#include "Configured.hpp"
#include "IStructure.hpp"

namespace Nextsim {

class RectGrid : public IStructure, public Configured<RectGrid> {
public:
    RectGrid();
    void configure() override; //!< rectgrid.nx / rectgrid.ny / rectgrid.nLayers
    void init(const std::string& filePath) override;
    void dump(const std::string& filePath) const override;
    std::string structureType() const override { return "rectgrid"; }
    int nIceLayers() const override { return m_store.nLayers; }
    int nx() const override { return m_nx; }
    int ny() const override { return m_ny; }
    FieldStore& fields() override { return m_store; }
    const FieldStore& fields() const override { return m_store; }
    void resize(int nx, int ny, int nLayers);

    int resetCursor() override;
    bool validCursor() const override { return m_cursor < m_store.n; }
    ElementData& cursorData() override { return m_current; }
    void incrCursor() override;

    //! Reads the structure type recorded in a restart file ("" if unreadable; throwOnError: a corrupt or
    //! unsupported HDF5 file raises Hdf5Error instead).
    static std::string typeInFile(const std::string& filePath, bool throwOnError = false);

protected:
    int m_nx, m_ny;
    FieldStore m_store;
    std::size_t m_cursor = 0;
    ElementData m_current;
};

class DevGrid : public RectGrid {
public:
    DevGrid() { resize(10, 10, 1); } // core/src/modules/DevGrid.cpp:20, DevGrid.hpp:49
    void configure() override { } // the reference's grid is a compile-time 10x10
    std::string structureType() const override { return "devgrid"; }
};

class StructureFactory {
public:
    //! std::invalid_argument for an unknown name (core/src/StructureFactory.cpp:20-44)
    static std::shared_ptr<IStructure> generate(const std::string& structureName);
    static std::shared_ptr<IStructure> generateFromFile(const std::string& filePath);
};

} // namespace Nextsim
