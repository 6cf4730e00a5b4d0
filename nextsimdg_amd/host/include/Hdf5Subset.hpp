// Hdf5Subset.hpp -- dependency-free reader and writer for the subset of HDF5 that the reference's
// NetCDF-4 restart files use (core/src/DevGridIO.cpp:65-210 writes them through netCDF-cxx4; neither
// netCDF nor the HDF5 development files exist in this image).
//
// What is understood (HDF5 File Format Specification version 3):
//   * superblock version 2 / 3, 8-byte (or 4-byte) offsets and lengths
//   * version-2 object headers ("OHDR") with continuation chunks ("OCHK"), checksums verified
//   * groups with compact link storage (link messages in the header) and with dense link storage whose
//     fractal heap consists of a root direct block or a root indirect block of direct blocks (the name
//     index B-tree is not needed: the heap blocks are scanned)
//   * hard links only
//   * datasets: simple dataspaces (version 1 / 2), fixed-point and IEEE floating-point datatypes of 1-8
//     bytes in either byte order, contiguous or compact layout (layout message version 3 / 4), no filters
//   * attributes (message version 1-3) of fixed-length string type and of the numeric types above
// Anything else (old-style groups of superblock 0/1 files, chunked or filtered datasets, variable-length
// types, soft/external links) raises Hdf5Error naming the unsupported feature.
//
// The writer produces files of the same subset (superblock 2, version-2 object headers, compact links,
// contiguous little-endian float64 datasets, fixed-length string attributes) that `h5dump` and the HDF5
// library read; it does not add the dimension-scale attributes NetCDF-4 uses to name dimensions.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace Nextsim {

class Hdf5Error : public std::runtime_error {
public:
    explicit Hdf5Error(const std::string& what)
        : std::runtime_error("Hdf5Subset: " + what)
    {
    }
};

//! Jenkins lookup3 (hashlittle, initval 0): the checksum of version-2 HDF5 metadata.
std::uint32_t hdf5Checksum(const unsigned char* data, std::size_t length);

class Hdf5File {
public:
    //! true if the file starts with the HDF5 signature
    static bool isHdf5(const std::string& filePath);

    //! Reads the whole file into memory and parses the superblock.  Throws Hdf5Error.
    explicit Hdf5File(const std::string& filePath);

    //! names of the links of a group, e.g. listGroup("/data")
    std::vector<std::string> listGroup(const std::string& groupPath) const;
    bool exists(const std::string& path) const;

    //! dimensions of a dataset (slowest first)
    std::vector<std::uint64_t> dims(const std::string& datasetPath) const;
    //! all elements of a numeric dataset converted to double, in file (row-major) order
    std::vector<double> readDoubles(const std::string& datasetPath) const;

    //! a fixed-length string attribute of a group or dataset (trailing NULs removed)
    std::string stringAttribute(const std::string& objectPath, const std::string& name) const;
    bool hasAttribute(const std::string& objectPath, const std::string& name) const;

private:
    struct Message {
        int type;
        std::size_t offset, size; // body
    };
    struct NumericType {
        int cls = -1; // 0 fixed point, 1 floating point, 3 string
        std::size_t size = 0;
        bool bigEndian = false, isSigned = false;
    };
    std::vector<unsigned char> m_data;
    int m_so = 8, m_sl = 8; // size of offsets / lengths
    std::uint64_t m_base = 0, m_root = 0;

    std::uint64_t u(std::size_t off, int n) const;
    void need(std::size_t off, std::size_t n, const char* what) const;
    std::vector<Message> messages(std::uint64_t headerAddress) const;
    void chunkMessages(std::size_t begin, std::size_t end, bool creationOrder, std::vector<Message>& out,
        std::vector<std::pair<std::uint64_t, std::uint64_t>>& continuations) const;
    std::map<std::string, std::uint64_t> links(std::uint64_t groupHeader) const;
    bool parseLink(std::size_t& p, std::size_t end, std::map<std::string, std::uint64_t>& out) const;
    void heapLinks(std::uint64_t heapAddress, std::map<std::string, std::uint64_t>& out) const;
    void scanDirectBlock(std::uint64_t address, std::uint64_t size, int blockOffsetBytes, bool checksummed,
        std::map<std::string, std::uint64_t>& out) const;
    std::uint64_t resolve(const std::string& path) const;
    NumericType parseType(std::size_t off) const;
    std::vector<std::uint64_t> parseSpace(std::size_t off) const;
    double element(const unsigned char* p, const NumericType& t) const;
    bool findAttribute(std::uint64_t header, const std::string& name, NumericType& type, std::vector<std::uint64_t>& dims,
        std::size_t& dataOff) const;
};

//! Writer of the same subset: groups, float64 datasets, string and int32 attributes, and NetCDF-4 named dimensions
//! (HDF5 dimension scales), collected in memory and laid out by write().
class Hdf5Writer {
public:
    //! creates the group (and its parents) if it does not exist yet
    void group(const std::string& path);
    void stringAttribute(const std::string& objectPath, const std::string& name, const std::string& value);
    void intAttribute(const std::string& objectPath, const std::string& name, const std::vector<std::int32_t>& values, bool scalar);
    void dataset(const std::string& path, const std::vector<std::uint64_t>& dims, const std::vector<double>& values);
    //! A NetCDF-4 dimension `name` of length n in `groupPath` that is not a variable: the dataset netCDF creates for it
    //! (big-endian float32, no values written) with CLASS = "DIMENSION_SCALE", the NAME netCDF gives such dimensions and
    //! _Netcdf4Dimid (core/src/DevGridIO.cpp:169-172 creates x, y, nLayers this way through netCDF-cxx4).
    void dimension(const std::string& groupPath, const std::string& name, std::uint64_t n, int dimid);
    //! Names the axes of a dataset: DIMENSION_LIST (one object reference per axis, through the global heap) and
    //! _Netcdf4Coordinates on the variable, an entry in REFERENCE_LIST of every dimension (core/src/DevGridIO.cpp:174-201).
    void attachDimensions(const std::string& datasetPath, const std::vector<std::string>& dimensionPaths);
    void write(const std::string& filePath) const;

private:
    struct Attribute {
        std::string name;
        std::vector<unsigned char> type, space, data; // message parts; `data` is patched with addresses by write()
        std::vector<std::pair<std::size_t, std::string>> objectRefs; // offset in data -> path whose header address goes there
        std::vector<std::pair<std::size_t, int>> heapRefs; // offset in data -> global heap object index (address + index go there)
    };
    struct Node {
        bool isDataset = false, isDimension = false;
        int dimid = -1;
        std::vector<std::string> children; // insertion order
        std::vector<Attribute> attributes;
        std::vector<std::uint64_t> dims;
        std::vector<double> values;
        std::vector<std::string> axes; // attached dimensions (variables)
        std::vector<std::pair<std::string, int>> referencedBy; // (variable, axis) pairs (dimensions)
    };
    std::map<std::string, Node> m_nodes = { { "/", Node() } };
    Node& ensureGroup(const std::string& path);
    Node& find(const std::string& path);
};

} // namespace Nextsim
