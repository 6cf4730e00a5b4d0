// ElementData.hpp -- per-element view of the structure's fields.
//
// The reference stores one heap-allocated ElementData object per element (array of structs:
// core/src/include/ElementData.hpp:30-68 = PrognosticData + PhysicsData + ExternalData, ~0.5 KB each).
// Here the fields live in flat struct-of-arrays planes (FieldStore) that are copied to HBM as they
// are; ElementData is a light proxy (store pointer + index) that offers the reference's accessor
// names, so host code written against the reference (e.g. DummyExternalData::setAll,
// core/src/include/DummyExternalData.hpp:22-34, or the tests' PrognosticGenerator chains) reads the
// same.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

namespace Nextsim {

//! What a dynamics run carries from one step to the next beyond the cell means hice / cice of the FieldStore: the higher DG2 coefficients of
//! the advected thickness and concentration, the CG2 velocity and the 24 stress coefficients.  The reference writes every prognostic field
//! it has into its restart file (core/src/DevGridIO.cpp:169-201, read back at :101-138); its snapshot has no dynamics, so these are
//! additional variables of the same group (names and dimensions: RectGrid.hpp).  x = the slow index of the file, y the fast one.
struct DynamicsState {
    bool present = false; //!< false: a run starts from rest with piecewise-constant fields (a file of the reference, or of a column-only run)
    std::vector<double> hdg, adg; //!< coefficients 1..5 of H and A: [5][x][y]
    std::vector<double> u, v; //!< nodal velocity on the (2x+1) x (2y+1) lattice
    std::vector<double> s11, s12, s22; //!< stress coefficients: [8][x][y]
    void resize(std::size_t nx, std::size_t ny)
    {
        hdg.assign(5 * nx * ny, 0.), adg.assign(5 * nx * ny, 0.);
        u.assign((2 * nx + 1) * (2 * ny + 1), 0.), v.assign((2 * nx + 1) * (2 * ny + 1), 0.);
        s11.assign(8 * nx * ny, 0.), s12.assign(8 * nx * ny, 0.), s22.assign(8 * nx * ny, 0.);
    }
    void clear()
    {
        present = false;
        for (auto* a : { &hdg, &adg, &u, &v, &s11, &s12, &s22 })
            a->clear();
    }
};

//! Struct-of-arrays field container: every plane has n = nx*ny doubles, element index i*ny' ... see IStructure.
struct FieldStore {
    std::size_t n = 0;
    int nLayers = 1;
    // prognostic (core/src/include/PrognosticData.hpp:86-92)
    std::vector<double> hice, cice, hsnow, sst, sss;
    std::vector<double> tice; // nLayers planes: tice[l*n + e]
    // external forcing (core/src/include/ExternalData.hpp:66-74)
    std::vector<double> tair, tdew, slp, mixrat, qsw, qlw, mld, snowfall;
    // physics input / persistent state (physics/src/include/PhysicsData.hpp:26; NextsimPhysics.hpp m_newice)
    std::vector<double> wind, newice;
    // state of the dynamics between two steps (absent until a dynamics run has stopped, or a restart file that holds it was read)
    DynamicsState dyn;

    void resize(std::size_t nElements, int nIceLayers);
};

//! Builder with the reference's method names (core/src/include/PrognosticGenerator.hpp:17-90).
class PrognosticGenerator {
public:
    PrognosticGenerator& hice(double v) { m_hice = v; return *this; }
    PrognosticGenerator& cice(double v) { m_cice = v; return *this; }
    PrognosticGenerator& hsnow(double v) { m_hsnow = v; return *this; }
    PrognosticGenerator& sst(double v) { m_sst = v; return *this; }
    PrognosticGenerator& sss(double v) { m_sss = v; return *this; }
    PrognosticGenerator& tice(const std::vector<double>& v) { m_tice = v; return *this; }
    double m_hice = 0, m_cice = 0, m_hsnow = 0, m_sst = 0, m_sss = 0;
    std::vector<double> m_tice;
};

class ElementData {
public:
    ElementData(FieldStore* store, std::size_t index)
        : s(store)
        , e(index)
    {
    }
    ElementData& operator=(const PrognosticGenerator& g);

    // PrognosticData surface
    double& iceThickness() { return s->hice[e]; }
    double& iceConcentration() { return s->cice[e]; }
    double& snowThickness() { return s->hsnow[e]; }
    double& seaSurfaceTemperature() { return s->sst[e]; }
    double& seaSurfaceSalinity() { return s->sss[e]; }
    double& iceTemperature(int layer) { return s->tice[(std::size_t)layer * s->n + e]; }
    double iceTrueThickness() const { return s->cice[e] != 0 ? s->hice[e] / s->cice[e] : 0; } // PrognosticData.hpp:56
    double snowTrueThickness() const { return s->cice[e] != 0 ? s->hsnow[e] / s->cice[e] : 0; } // :75
    int nIceLayers() const { return s->nLayers; }
    // ExternalData surface
    double& airTemperature() { return s->tair[e]; }
    double& dewPoint2m() { return s->tdew[e]; }
    double& airPressure() { return s->slp[e]; }
    double& mixingRatio() { return s->mixrat[e]; }
    double& incomingShortwave() { return s->qsw[e]; }
    double& incomingLongwave() { return s->qlw[e]; }
    double& mixedLayerDepth() { return s->mld[e]; }
    double& snowfall() { return s->snowfall[e]; }
    double mixedLayerBulkHeatCapacity() const { return s->mld[e] * 1025. * 4186.84; } // ExternalData.hpp:60
    // PhysicsData surface (input only; derived quantities live in registers on the GPU)
    double& windSpeed() { return s->wind[e]; }
    double& newIce() { return s->newice[e]; }

    std::size_t index() const { return e; }

private:
    FieldStore* s;
    std::size_t e;
};

} // namespace Nextsim
