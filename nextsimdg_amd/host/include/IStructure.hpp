// IStructure.hpp -- the field container handed to the model step.  Keeps the reference's surface
// (core/src/modules/include/IStructure.hpp:32-137): init/dump by file path, structureType() and its
// case-insensitive check, nIceLayers(), and the element cursor protocol
//     for (s.cursor = 0; s.cursor; ++s.cursor) { auto& d = *s.cursor; ... }
// and adds what a batched GPU step needs: the grid shape and direct access to the SoA planes.
// Linear element index = i*ny + j with i the slow ("x" in the restart file, dims (x, y)) index, the
// same x-major order as core/src/DevGridIO.cpp:107-109 (i*nx + j on the reference's square grid).
#pragma once
#include <algorithm>
#include <string>

#include "ElementData.hpp"

namespace Nextsim {

//! The element cursor of a structure S (protocol of core/src/modules/include/IStructure.hpp:96-122), kept
//! outside the interface so that any container with resetCursor / validCursor / cursorData / incrCursor can use it.
template <class S, class E> class StructureCursor {
public:
    explicit StructureCursor(S& structure)
        : m_s(structure)
    {
    }
    S& operator=(const int position) const
    {
        if (position == 0)
            m_s.resetCursor();
        return m_s;
    }
    operator bool() const { return m_s.validCursor(); }
    E& operator*() const { return m_s.cursorData(); }
    E* operator->() const { return &m_s.cursorData(); }
    S& operator++() const
    {
        m_s.incrCursor();
        return m_s;
    }

private:
    S& m_s;
};

class IStructure {
public:
    IStructure()
        : cursor(*this)
    {
    }
    virtual ~IStructure() = default;

    //! Initialises the structure; an empty path means "from configuration constants" (init.* keys).
    virtual void init(const std::string& filePath) = 0;
    virtual std::string structureType() const { return "none"; }
    bool structureTypeCheck(const std::string& str) const
    {
        std::string a = structureType(), b = str;
        auto lower = [](std::string& x) { std::transform(x.begin(), x.end(), x.begin(), [](unsigned char c) { return (char)std::tolower(c); }); };
        lower(a);
        lower(b);
        return a == b;
    }
    virtual int nIceLayers() const = 0;
    virtual void dump(const std::string& filePath) const = 0;

    // grid + SoA access (new)
    virtual int nx() const = 0; //!< slow dimension ("x" of the restart file)
    virtual int ny() const = 0; //!< fast dimension ("y")
    virtual FieldStore& fields() = 0;
    virtual const FieldStore& fields() const = 0;
    std::size_t size() const { return fields().n; }

    // cursor protocol
    virtual int resetCursor() = 0;
    virtual bool validCursor() const = 0;
    virtual ElementData& cursorData() = 0;
    virtual void incrCursor() = 0;

    //! `cursor = 0` rewinds, `bool(cursor)` tests validity, `*cursor` / `cursor->` give the element, `++cursor` advances
    typedef StructureCursor<IStructure, ElementData> Cursor;
    const Cursor cursor;

    // node names of the restart layout (core/src/modules/include/IStructure.hpp:127-132)
    static std::string metadataNodeName() { return "structure"; }
    static std::string dataNodeName() { return "data"; }
    static std::string typeNodeName() { return "type"; }
};

//! Constant forcing for every element, values of core/src/include/DummyExternalData.hpp:22-34.
struct DummyExternalData {
    static void setAll(IStructure& is);
};

} // namespace Nextsim
