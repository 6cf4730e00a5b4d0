// Hdf5Subset.cpp -- see Hdf5Subset.hpp.  Field layouts follow the HDF5 File Format Specification
// version 3 (sections II.A superblock, III.D fractal heap, IV.A.1.b version-2 object header prefix,
// IV.A.2.* header messages).
#include "Hdf5Subset.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>

namespace Nextsim {

namespace {
const unsigned char SIGNATURE[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
const std::uint64_t UNDEF = ~std::uint64_t(0);

enum MessageType {
    MSG_NIL = 0x00,
    MSG_DATASPACE = 0x01,
    MSG_LINK_INFO = 0x02,
    MSG_DATATYPE = 0x03,
    MSG_FILL_VALUE = 0x05,
    MSG_LINK = 0x06,
    MSG_LAYOUT = 0x08,
    MSG_GROUP_INFO = 0x0A,
    MSG_FILTERS = 0x0B,
    MSG_ATTRIBUTE = 0x0C,
    MSG_CONTINUATION = 0x10,
    MSG_SYMBOL_TABLE = 0x11,
    MSG_ATTRIBUTE_INFO = 0x15
};

inline std::uint32_t rot(std::uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

std::vector<std::string> splitPath(const std::string& path)
{
    std::vector<std::string> parts;
    std::string item;
    std::stringstream ss(path);
    while (std::getline(ss, item, '/'))
        if (!item.empty())
            parts.push_back(item);
    return parts;
}
} // namespace

std::uint32_t hdf5Checksum(const unsigned char* k, std::size_t length)
{
    std::uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + (std::uint32_t)length;
    auto word = [&](int i) { return (std::uint32_t)k[i] | ((std::uint32_t)k[i + 1] << 8) | ((std::uint32_t)k[i + 2] << 16) | ((std::uint32_t)k[i + 3] << 24); };
    while (length > 12) {
        a += word(0), b += word(4), c += word(8);
        a -= c, a ^= rot(c, 4), c += b;
        b -= a, b ^= rot(a, 6), a += c;
        c -= b, c ^= rot(b, 8), b += a;
        a -= c, a ^= rot(c, 16), c += b;
        b -= a, b ^= rot(a, 19), a += c;
        c -= b, c ^= rot(b, 4), b += a;
        length -= 12;
        k += 12;
    }
    if (length == 0)
        return c;
    std::uint32_t t[3] = { 0, 0, 0 };
    for (std::size_t i = 0; i < length; ++i)
        t[i / 4] += (std::uint32_t)k[i] << (8 * (i % 4));
    a += t[0], b += t[1], c += t[2];
    c ^= b, c -= rot(b, 14);
    a ^= c, a -= rot(c, 11);
    b ^= a, b -= rot(a, 25);
    c ^= b, c -= rot(b, 16);
    a ^= c, a -= rot(c, 4);
    b ^= a, b -= rot(a, 14);
    c ^= b, c -= rot(b, 24);
    return c;
}

// ------------------------------------------------------------------------------------------------ reader

bool Hdf5File::isHdf5(const std::string& filePath)
{
    std::ifstream f(filePath, std::ios::binary);
    unsigned char sig[8];
    return f && f.read(reinterpret_cast<char*>(sig), 8) && std::memcmp(sig, SIGNATURE, 8) == 0;
}

void Hdf5File::need(std::size_t off, std::size_t n, const char* what) const
{
    if (off > m_data.size() || n > m_data.size() - off)
        throw Hdf5Error(std::string("truncated file while reading ") + what);
}

std::uint64_t Hdf5File::u(std::size_t off, int n) const
{
    need(off, (std::size_t)n, "an integer field");
    std::uint64_t v = 0;
    bool allOnes = true;
    for (int i = n - 1; i >= 0; --i) {
        v = (v << 8) | m_data[off + i];
        allOnes = allOnes && m_data[off + i] == 0xff;
    }
    return (allOnes && n < 8 && n >= 4) ? UNDEF : v; // undefined address in a 4-byte-offset file
}

Hdf5File::Hdf5File(const std::string& filePath)
{
    std::ifstream f(filePath, std::ios::binary);
    if (!f)
        throw Hdf5Error("cannot open " + filePath);
    m_data.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    need(0, 12, "the superblock");
    if (std::memcmp(m_data.data(), SIGNATURE, 8) != 0)
        throw Hdf5Error(filePath + " is not an HDF5 file");
    const int version = m_data[8];
    if (version != 2 && version != 3)
        throw Hdf5Error("superblock version " + std::to_string(version) + " (old-style symbol-table groups) is not supported");
    m_so = m_data[9], m_sl = m_data[10];
    if ((m_so != 4 && m_so != 8) || (m_sl != 4 && m_sl != 8))
        throw Hdf5Error("unsupported size of offsets / lengths");
    const std::size_t end = 12 + 4 * (std::size_t)m_so;
    need(0, end + 4, "the superblock");
    if (hdf5Checksum(m_data.data(), end) != (std::uint32_t)u(end, 4))
        throw Hdf5Error("superblock checksum mismatch");
    m_base = u(12, m_so);
    m_root = u(12 + 3 * (std::size_t)m_so, m_so);
}

void Hdf5File::chunkMessages(std::size_t begin, std::size_t end, bool creationOrder, std::vector<Message>& out,
    std::vector<std::pair<std::uint64_t, std::uint64_t>>& continuations) const
{
    const std::size_t header = 4 + (creationOrder ? 2 : 0);
    std::size_t p = begin;
    while (p + header <= end) {
        const int type = m_data[p];
        const std::size_t size = (std::size_t)u(p + 1, 2);
        const std::size_t body = p + header;
        if (body + size > end)
            break; // gap at the end of the chunk
        if (type == MSG_CONTINUATION)
            continuations.emplace_back(u(body, m_so), u(body + m_so, m_sl));
        else if (type != MSG_NIL)
            out.push_back({ type, body, size });
        p = body + size;
    }
}

std::vector<Hdf5File::Message> Hdf5File::messages(std::uint64_t headerAddress) const
{
    const std::size_t off = (std::size_t)(m_base + headerAddress);
    need(off, 7, "an object header");
    if (std::memcmp(&m_data[off], "OHDR", 4) != 0)
        throw Hdf5Error("version-1 object headers are not supported (no OHDR signature)");
    if (m_data[off + 4] != 2)
        throw Hdf5Error("unsupported object header version");
    const int flags = m_data[off + 5];
    std::size_t p = off + 6;
    if (flags & 0x20)
        p += 16; // access, modification, change and birth times
    if (flags & 0x10)
        p += 4; // attribute storage phase-change values
    const int sizeBytes = 1 << (flags & 3);
    const std::size_t chunk0 = (std::size_t)u(p, sizeBytes);
    p += sizeBytes;
    need(p, chunk0 + 4, "an object header chunk");
    if (hdf5Checksum(&m_data[off], p + chunk0 - off) != (std::uint32_t)u(p + chunk0, 4))
        throw Hdf5Error("object header checksum mismatch");
    std::vector<Message> out;
    std::vector<std::pair<std::uint64_t, std::uint64_t>> cont;
    chunkMessages(p, p + chunk0, flags & 0x04, out, cont);
    for (std::size_t i = 0; i < cont.size(); ++i) { // the list grows while it is walked
        const std::size_t coff = (std::size_t)(m_base + cont[i].first), len = (std::size_t)cont[i].second;
        need(coff, len, "an object header continuation");
        if (len < 8 || std::memcmp(&m_data[coff], "OCHK", 4) != 0)
            throw Hdf5Error("bad object header continuation block");
        if (hdf5Checksum(&m_data[coff], len - 4) != (std::uint32_t)u(coff + len - 4, 4))
            throw Hdf5Error("object header continuation checksum mismatch");
        chunkMessages(coff + 4, coff + len - 4, flags & 0x04, out, cont);
    }
    return out;
}

bool Hdf5File::parseLink(std::size_t& p, std::size_t end, std::map<std::string, std::uint64_t>& out) const
{
    if (p + 3 > end || m_data[p] != 1)
        return false;
    const int flags = m_data[p + 1];
    std::size_t q = p + 2;
    int linkType = 0;
    if (flags & 0x08)
        linkType = m_data[q++];
    if (flags & 0x04)
        q += 8; // creation order
    if (flags & 0x10)
        q += 1; // character set of the name
    const int lenSize = 1 << (flags & 3);
    if (q + lenSize > end)
        return false;
    const std::size_t nameLen = (std::size_t)u(q, lenSize);
    q += lenSize;
    if (nameLen == 0 || q + nameLen > end)
        return false;
    const std::string name(reinterpret_cast<const char*>(&m_data[q]), nameLen);
    q += nameLen;
    if (linkType == 0) {
        if (q + m_so > end)
            return false;
        out[name] = u(q, m_so);
        q += m_so;
    } else { // soft (1) or external (64) link: length-prefixed value, skipped
        if (q + 2 > end)
            return false;
        q += 2 + (std::size_t)u(q, 2);
    }
    p = q;
    return true;
}

void Hdf5File::scanDirectBlock(std::uint64_t address, std::uint64_t size, int blockOffsetBytes, bool checksummed,
    std::map<std::string, std::uint64_t>& out) const
{
    const std::size_t off = (std::size_t)(m_base + address);
    need(off, (std::size_t)size, "a fractal heap direct block");
    if (std::memcmp(&m_data[off], "FHDB", 4) != 0)
        throw Hdf5Error("bad fractal heap direct block");
    std::size_t p = off + 5 + m_so + blockOffsetBytes + (checksummed ? 4 : 0);
    const std::size_t end = off + (std::size_t)size;
    while (p < end) {
        if (m_data[p] == 0) { // free space
            ++p;
            continue;
        }
        if (!parseLink(p, end, out))
            break;
    }
}

void Hdf5File::heapLinks(std::uint64_t heapAddress, std::map<std::string, std::uint64_t>& out) const
{
    const std::size_t off = (std::size_t)(m_base + heapAddress);
    need(off, 5, "a fractal heap header");
    if (std::memcmp(&m_data[off], "FRHP", 4) != 0 || m_data[off + 4] != 0)
        throw Hdf5Error("bad fractal heap header");
    std::size_t q = off + 5;
    q += 2; // heap ID length
    const std::uint64_t filterLen = u(q, 2);
    q += 2;
    const int flags = m_data[q];
    q += 1;
    q += 4; // maximum size of managed objects
    q += m_sl + m_so; // next huge object ID, B-tree of huge objects
    q += m_sl + m_so; // free space in managed blocks, free-space manager
    q += 4 * (std::size_t)m_sl; // managed space, allocated managed space, allocation iterator, number of managed objects
    q += 4 * (std::size_t)m_sl; // huge / tiny object sizes and counts
    const std::uint64_t tableWidth = u(q, 2);
    q += 2;
    const std::uint64_t startBlock = u(q, m_sl);
    q += m_sl;
    const std::uint64_t maxDirect = u(q, m_sl);
    q += m_sl;
    const int maxHeapBits = (int)u(q, 2);
    q += 2;
    q += 2; // starting number of rows of the root indirect block
    const std::uint64_t rootAddr = u(q, m_so);
    q += m_so;
    const std::uint64_t curRows = u(q, 2);
    if (filterLen > 0)
        throw Hdf5Error("filtered fractal heaps are not supported");
    if (rootAddr == UNDEF)
        return;
    const int blockOffsetBytes = (maxHeapBits + 7) / 8;
    const bool checksummed = flags & 0x02;
    if (curRows == 0) {
        scanDirectBlock(rootAddr, startBlock, blockOffsetBytes, checksummed, out);
        return;
    }
    const std::size_t ioff = (std::size_t)(m_base + rootAddr);
    need(ioff, 5, "a fractal heap indirect block");
    if (std::memcmp(&m_data[ioff], "FHIB", 4) != 0)
        throw Hdf5Error("bad fractal heap indirect block");
    std::uint64_t maxDirectRows = 2;
    for (std::uint64_t s = startBlock; s < maxDirect; s <<= 1)
        ++maxDirectRows;
    std::size_t p = ioff + 5 + m_so + blockOffsetBytes;
    for (std::uint64_t row = 0; row < curRows; ++row) {
        if (row >= maxDirectRows)
            throw Hdf5Error("fractal heaps with nested indirect blocks are not supported");
        const std::uint64_t size = row < 2 ? startBlock : startBlock << (row - 1);
        for (std::uint64_t col = 0; col < tableWidth; ++col, p += m_so) {
            const std::uint64_t addr = u(p, m_so);
            if (addr != UNDEF)
                scanDirectBlock(addr, size, blockOffsetBytes, checksummed, out);
        }
    }
}

std::map<std::string, std::uint64_t> Hdf5File::links(std::uint64_t groupHeader) const
{
    std::map<std::string, std::uint64_t> out;
    for (const Message& m : messages(groupHeader)) {
        if (m.type == MSG_LINK) {
            std::size_t p = m.offset;
            parseLink(p, m.offset + m.size, out);
        } else if (m.type == MSG_LINK_INFO) {
            const int flags = m_data[m.offset + 1];
            const std::size_t q = m.offset + 2 + ((flags & 1) ? 8 : 0);
            const std::uint64_t heap = u(q, m_so);
            if (heap != UNDEF)
                heapLinks(heap, out);
        } else if (m.type == MSG_SYMBOL_TABLE) {
            throw Hdf5Error("old-style (symbol table) groups are not supported");
        }
    }
    return out;
}

std::uint64_t Hdf5File::resolve(const std::string& path) const
{
    std::uint64_t at = m_root;
    for (const std::string& part : splitPath(path)) {
        const auto l = links(at);
        const auto it = l.find(part);
        if (it == l.end())
            throw Hdf5Error("no object named " + path);
        at = it->second;
    }
    return at;
}

bool Hdf5File::exists(const std::string& path) const
{
    try {
        resolve(path);
        return true;
    } catch (const Hdf5Error&) {
        return false;
    }
}

std::vector<std::string> Hdf5File::listGroup(const std::string& groupPath) const
{
    std::vector<std::string> names;
    for (const auto& kv : links(resolve(groupPath)))
        names.push_back(kv.first);
    return names;
}

Hdf5File::NumericType Hdf5File::parseType(std::size_t off) const
{
    need(off, 8, "a datatype message");
    NumericType t;
    t.cls = m_data[off] & 0x0f;
    t.size = (std::size_t)u(off + 4, 4);
    const int bits0 = m_data[off + 1];
    if (t.cls == 0) {
        t.bigEndian = bits0 & 0x01;
        t.isSigned = bits0 & 0x08;
    } else if (t.cls == 1) {
        if (bits0 & 0x40)
            throw Hdf5Error("VAX floating-point byte order is not supported");
        t.bigEndian = bits0 & 0x01;
    }
    return t;
}

std::vector<std::uint64_t> Hdf5File::parseSpace(std::size_t off) const
{
    need(off, 4, "a dataspace message");
    const int version = m_data[off], rank = m_data[off + 1];
    if (version != 1 && version != 2)
        throw Hdf5Error("unsupported dataspace message version");
    if (version == 2 && m_data[off + 3] == 2)
        throw Hdf5Error("null dataspaces are not supported");
    std::size_t q = off + (version == 1 ? 8 : 4);
    std::vector<std::uint64_t> d((std::size_t)rank);
    for (int i = 0; i < rank; ++i, q += m_sl)
        d[i] = u(q, m_sl);
    return d;
}

double Hdf5File::element(const unsigned char* p, const NumericType& t) const
{
    unsigned char b[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // little-endian image
    for (std::size_t i = 0; i < t.size; ++i)
        b[i] = t.bigEndian ? p[t.size - 1 - i] : p[i];
    if (t.cls == 1) {
        if (t.size == 8) {
            double v;
            std::memcpy(&v, b, 8);
            return v;
        }
        if (t.size == 4) {
            float v;
            std::memcpy(&v, b, 4);
            return v;
        }
        throw Hdf5Error("unsupported floating-point size");
    }
    std::uint64_t v = 0;
    for (int i = (int)t.size - 1; i >= 0; --i)
        v = (v << 8) | b[i];
    if (t.isSigned && t.size < 8 && (v >> (8 * t.size - 1)))
        v |= ~std::uint64_t(0) << (8 * t.size);
    return t.isSigned ? (double)(std::int64_t)v : (double)v;
}

std::vector<std::uint64_t> Hdf5File::dims(const std::string& datasetPath) const
{
    for (const Message& m : messages(resolve(datasetPath)))
        if (m.type == MSG_DATASPACE)
            return parseSpace(m.offset);
    throw Hdf5Error(datasetPath + " has no dataspace");
}

std::vector<double> Hdf5File::readDoubles(const std::string& datasetPath) const
{
    const Message *space = nullptr, *type = nullptr, *layout = nullptr;
    const std::vector<Message> msgs = messages(resolve(datasetPath));
    for (const Message& m : msgs) {
        if (m.type == MSG_DATASPACE)
            space = &m;
        else if (m.type == MSG_DATATYPE)
            type = &m;
        else if (m.type == MSG_LAYOUT)
            layout = &m;
        else if (m.type == MSG_FILTERS)
            throw Hdf5Error(datasetPath + ": filtered (compressed) datasets are not supported");
    }
    if (!space || !type || !layout)
        throw Hdf5Error(datasetPath + " is not a dataset");
    const NumericType t = parseType(type->offset);
    if ((t.cls != 0 && t.cls != 1) || t.size == 0 || t.size > 8)
        throw Hdf5Error(datasetPath + ": only fixed-point and floating-point datasets are supported");
    std::uint64_t n = 1;
    for (std::uint64_t d : parseSpace(space->offset))
        n *= d;
    const int version = m_data[layout->offset], cls = m_data[layout->offset + 1];
    if (version != 3 && version != 4)
        throw Hdf5Error(datasetPath + ": unsupported data layout message version");
    std::size_t dataOff, dataSize;
    if (cls == 1) {
        const std::uint64_t addr = u(layout->offset + 2, m_so);
        if (addr == UNDEF)
            throw Hdf5Error(datasetPath + " has no allocated storage");
        dataOff = (std::size_t)(m_base + addr);
        dataSize = (std::size_t)u(layout->offset + 2 + m_so, m_sl);
    } else if (cls == 0) {
        dataSize = (std::size_t)u(layout->offset + 2, 2);
        dataOff = layout->offset + 4;
    } else {
        throw Hdf5Error(datasetPath + ": chunked datasets are not supported");
    }
    if (n * t.size > dataSize)
        throw Hdf5Error(datasetPath + ": storage smaller than the dataspace");
    need(dataOff, (std::size_t)n * t.size, "dataset storage");
    std::vector<double> out((std::size_t)n);
    for (std::size_t i = 0; i < out.size(); ++i)
        out[i] = element(&m_data[dataOff + i * t.size], t);
    return out;
}

bool Hdf5File::findAttribute(std::uint64_t header, const std::string& name, NumericType& type, std::vector<std::uint64_t>& dims,
    std::size_t& dataOff) const
{
    bool dense = false;
    for (const Message& m : messages(header)) {
        if (m.type == MSG_ATTRIBUTE_INFO) {
            const int flags = m_data[m.offset + 1];
            const std::size_t q = m.offset + 2 + ((flags & 1) ? 2 : 0);
            dense = dense || u(q, m_so) != UNDEF;
        }
        if (m.type != MSG_ATTRIBUTE)
            continue;
        const int version = m_data[m.offset];
        if (version < 1 || version > 3)
            throw Hdf5Error("unsupported attribute message version");
        if (version > 1 && (m_data[m.offset + 1] & 0x03))
            throw Hdf5Error("shared attribute datatypes / dataspaces are not supported");
        const std::size_t nameSize = (std::size_t)u(m.offset + 2, 2), typeSize = (std::size_t)u(m.offset + 4, 2),
                          spaceSize = (std::size_t)u(m.offset + 6, 2);
        auto pad = [version](std::size_t s) { return version == 1 ? (s + 7) / 8 * 8 : s; };
        std::size_t q = m.offset + (version == 3 ? 9 : 8);
        need(q, pad(nameSize) + pad(typeSize) + pad(spaceSize), "an attribute");
        const std::string attrName(reinterpret_cast<const char*>(&m_data[q]));
        q += pad(nameSize);
        if (attrName != name)
            continue;
        type = parseType(q);
        q += pad(typeSize);
        dims = parseSpace(q);
        dataOff = q + pad(spaceSize);
        return true;
    }
    if (dense)
        throw Hdf5Error("dense attribute storage is not supported (attribute " + name + ")");
    return false;
}

bool Hdf5File::hasAttribute(const std::string& objectPath, const std::string& name) const
{
    NumericType t;
    std::vector<std::uint64_t> d;
    std::size_t off;
    return findAttribute(resolve(objectPath), name, t, d, off);
}

std::string Hdf5File::stringAttribute(const std::string& objectPath, const std::string& name) const
{
    NumericType t;
    std::vector<std::uint64_t> d;
    std::size_t off;
    if (!findAttribute(resolve(objectPath), name, t, d, off))
        throw Hdf5Error("no attribute " + name + " on " + objectPath);
    if (t.cls != 3)
        throw Hdf5Error("attribute " + name + " is not a fixed-length string");
    need(off, t.size, "attribute data");
    std::string s(reinterpret_cast<const char*>(&m_data[off]), t.size);
    while (!s.empty() && (s.back() == '\0' || s.back() == ' '))
        s.pop_back();
    return s;
}

// ------------------------------------------------------------------------------------------------ writer

namespace {
void put(std::vector<unsigned char>& b, std::uint64_t v, int n)
{
    for (int i = 0; i < n; ++i)
        b.push_back((unsigned char)(v >> (8 * i)));
}
void message(std::vector<unsigned char>& b, int type, const std::vector<unsigned char>& body)
{
    b.push_back((unsigned char)type);
    put(b, body.size(), 2);
    b.push_back(0);
    b.insert(b.end(), body.begin(), body.end());
}
std::string parentOf(const std::string& path)
{
    const auto pos = path.find_last_of('/');
    return pos == 0 ? "/" : path.substr(0, pos);
}
std::string leafOf(const std::string& path) { return path.substr(path.find_last_of('/') + 1); }
std::string normalise(const std::string& path)
{
    std::string out;
    for (const std::string& p : splitPath(path))
        out += "/" + p;
    return out.empty() ? "/" : out;
}
} // namespace

Hdf5Writer::Node& Hdf5Writer::ensureGroup(const std::string& pathIn)
{
    const std::string path = normalise(pathIn);
    auto it = m_nodes.find(path);
    if (it != m_nodes.end())
        return it->second;
    Node& parent = ensureGroup(parentOf(path));
    if (parent.isDataset)
        throw Hdf5Error("a dataset cannot have children: " + path);
    parent.children.push_back(leafOf(path));
    return m_nodes[path];
}

void Hdf5Writer::group(const std::string& path) { ensureGroup(path); }

Hdf5Writer::Node& Hdf5Writer::find(const std::string& path)
{
    const auto it = m_nodes.find(normalise(path));
    if (it == m_nodes.end())
        throw Hdf5Error("no object " + path);
    return it->second;
}

namespace {
// message parts of the datatypes / dataspaces the writer knows (HDF5 File Format Specification IV.A.2.b, IV.A.2.d)
std::vector<unsigned char> stringType(std::size_t size)
{
    std::vector<unsigned char> b = { 0x13, 0, 0, 0 }; // class 3 string, null-terminated, ASCII
    put(b, size, 4);
    return b;
}
std::vector<unsigned char> int32Type()
{
    std::vector<unsigned char> b = { 0x10, 0x08, 0, 0 }; // class 0 fixed point, little endian, signed
    put(b, 4, 4);
    put(b, 0, 2), put(b, 32, 2); // bit offset, precision
    return b;
}
std::vector<unsigned char> float64Type()
{
    std::vector<unsigned char> b = { 0x11, 0x20, 0x3f, 0x00 }; // IEEE float, little endian, implied mantissa msb, sign bit 63
    put(b, 8, 4);
    put(b, 0, 2), put(b, 64, 2); // bit offset, precision
    b.push_back(52), b.push_back(11), b.push_back(0), b.push_back(52); // exponent location/size, mantissa location/size
    put(b, 1023, 4);
    return b;
}
std::vector<unsigned char> float32BigEndianType()
{
    std::vector<unsigned char> b = { 0x11, 0x21, 0x1f, 0x00 }; // IEEE float, big endian, implied mantissa msb, sign bit 31
    put(b, 4, 4);
    put(b, 0, 2), put(b, 32, 2);
    b.push_back(23), b.push_back(8), b.push_back(0), b.push_back(23);
    put(b, 127, 4);
    return b;
}
std::vector<unsigned char> objectReferenceType()
{
    std::vector<unsigned char> b = { 0x17, 0, 0, 0 }; // class 7 reference, type 0 = object reference
    put(b, 8, 4);
    return b;
}
std::vector<unsigned char> vlenOfObjectReferencesType()
{
    std::vector<unsigned char> b = { 0x19, 0, 0, 0 }; // class 9 variable length, type 0 = sequence
    put(b, 16, 4); // in the file: 4-byte length + global heap id (8-byte address + 4-byte index)
    const auto base = objectReferenceType();
    b.insert(b.end(), base.begin(), base.end());
    return b;
}
std::vector<unsigned char> referenceListType()
{ // H5DS: compound { hobj_ref_t dataset; int dimension; } with the in-memory layout of ds_list_t (16 bytes)
    std::vector<unsigned char> b = { 0x36, 2, 0, 0 }; // class 6 compound, version 3, two members
    put(b, 16, 4);
    for (const char c : std::string("dataset"))
        b.push_back((unsigned char)c);
    b.push_back(0);
    b.push_back(0); // byte offset of the member (one byte: the compound is smaller than 256 bytes)
    auto t = objectReferenceType();
    b.insert(b.end(), t.begin(), t.end());
    for (const char c : std::string("dimension"))
        b.push_back((unsigned char)c);
    b.push_back(0);
    b.push_back(8);
    t = int32Type();
    b.insert(b.end(), t.begin(), t.end());
    return b;
}
std::vector<unsigned char> scalarSpace() { return { 2, 0, 0, 0 }; }
std::vector<unsigned char> simpleSpace(const std::vector<std::uint64_t>& dims)
{
    std::vector<unsigned char> b = { 2, (unsigned char)dims.size(), 0, 1 };
    for (std::uint64_t d : dims)
        put(b, d, 8);
    return b;
}
} // namespace

void Hdf5Writer::stringAttribute(const std::string& objectPath, const std::string& name, const std::string& value)
{
    Attribute a;
    a.name = name;
    a.type = stringType(value.size() + 1);
    a.space = scalarSpace();
    a.data.assign(value.begin(), value.end());
    a.data.push_back(0);
    find(objectPath).attributes.push_back(a);
}

void Hdf5Writer::intAttribute(const std::string& objectPath, const std::string& name, const std::vector<std::int32_t>& values, bool scalar)
{
    if (scalar && values.size() != 1)
        throw Hdf5Error("a scalar attribute has one value: " + name);
    Attribute a;
    a.name = name;
    a.type = int32Type();
    a.space = scalar ? scalarSpace() : simpleSpace({ (std::uint64_t)values.size() });
    for (std::int32_t v : values)
        put(a.data, (std::uint32_t)v, 4);
    find(objectPath).attributes.push_back(a);
}

void Hdf5Writer::dataset(const std::string& pathIn, const std::vector<std::uint64_t>& dims, const std::vector<double>& values)
{
    std::uint64_t n = 1;
    for (std::uint64_t d : dims)
        n *= d;
    if (n != values.size())
        throw Hdf5Error("dataset " + pathIn + ": dimensions and number of values disagree");
    const std::string path = normalise(pathIn);
    if (m_nodes.count(path))
        throw Hdf5Error("object exists: " + path);
    Node& node = ensureGroup(path); // creates the parents and the link
    node.isDataset = true;
    node.dims = dims;
    node.values = values;
}

void Hdf5Writer::dimension(const std::string& groupPath, const std::string& name, std::uint64_t n, int dimid)
{
    const std::string path = normalise(groupPath + "/" + name);
    if (m_nodes.count(path))
        throw Hdf5Error("object exists: " + path);
    Node& node = ensureGroup(path);
    node.isDataset = node.isDimension = true;
    node.dimid = dimid;
    node.dims = { n };
    stringAttribute(path, "CLASS", "DIMENSION_SCALE");
    char text[80];
    std::snprintf(text, sizeof text, "This is a netCDF dimension but not a netCDF variable.%10llu", (unsigned long long)n);
    stringAttribute(path, "NAME", text);
    intAttribute(path, "_Netcdf4Dimid", { dimid }, true);
}

void Hdf5Writer::attachDimensions(const std::string& datasetPath, const std::vector<std::string>& dimensionPaths)
{
    Node& var = find(datasetPath);
    if (!var.isDataset || var.isDimension || var.dims.size() != dimensionPaths.size())
        throw Hdf5Error("attachDimensions: " + datasetPath + " needs one dimension per axis");
    if (!var.axes.empty())
        throw Hdf5Error("attachDimensions: " + datasetPath + " has its dimensions already");
    std::vector<std::int32_t> ids;
    for (std::size_t k = 0; k < dimensionPaths.size(); ++k) {
        Node& d = find(dimensionPaths[k]);
        if (!d.isDimension || d.dims[0] != var.dims[k])
            throw Hdf5Error("attachDimensions: " + dimensionPaths[k] + " is not a dimension of the right length");
        d.referencedBy.emplace_back(normalise(datasetPath), (int)k);
        var.axes.push_back(normalise(dimensionPaths[k]));
        ids.push_back(d.dimid);
    }
    intAttribute(datasetPath, "_Netcdf4Coordinates", ids, false);
}

void Hdf5Writer::write(const std::string& filePath) const
{
    // ---- attributes that refer to other objects are generated here, when all attachments are known
    std::map<std::string, std::vector<Attribute>> attrs;
    int heapObjects = 0; // global heap: one 8-byte object reference per (variable, axis)
    std::vector<std::string> heapTargets; // heap object i+1 holds the address of this path
    for (const auto& kv : m_nodes) {
        std::vector<Attribute> list = kv.second.attributes;
        const Node& n = kv.second;
        if (!n.axes.empty()) {
            Attribute a;
            a.name = "DIMENSION_LIST";
            a.type = vlenOfObjectReferencesType();
            a.space = simpleSpace({ (std::uint64_t)n.axes.size() });
            for (const std::string& axis : n.axes) {
                put(a.data, 1, 4); // sequence of one reference
                a.heapRefs.emplace_back(a.data.size(), ++heapObjects);
                put(a.data, 0, 8), put(a.data, 0, 4); // heap address + object index, patched below
                heapTargets.push_back(axis);
            }
            list.insert(list.begin(), a);
        }
        if (n.isDimension && !n.referencedBy.empty()) {
            Attribute a;
            a.name = "REFERENCE_LIST";
            a.type = referenceListType();
            a.space = simpleSpace({ (std::uint64_t)n.referencedBy.size() });
            for (const auto& ref : n.referencedBy) {
                a.objectRefs.emplace_back(a.data.size(), ref.first);
                put(a.data, 0, 8);
                put(a.data, (std::uint32_t)ref.second, 4);
                put(a.data, 0, 4); // padding of ds_list_t
            }
            list.push_back(a);
        }
        attrs[kv.first] = list;
    }
    auto attributeBody = [](const Attribute& a) {
        std::vector<unsigned char> b = { 3, 0 };
        put(b, a.name.size() + 1, 2);
        put(b, a.type.size(), 2);
        put(b, a.space.size(), 2);
        b.push_back(0); // ASCII name
        b.insert(b.end(), a.name.begin(), a.name.end());
        b.push_back(0);
        b.insert(b.end(), a.type.begin(), a.type.end());
        b.insert(b.end(), a.space.begin(), a.space.end());
        b.insert(b.end(), a.data.begin(), a.data.end());
        return b;
    };
    // header size per node: messages only (the prefix and the checksum are added by the caller)
    auto headerSize = [&](const std::string& path, const Node& n) {
        std::size_t s = 0;
        if (n.isDataset) {
            s += 4 + 4 + 8 * n.dims.size(); // dataspace
            s += 4 + (n.isDimension ? float32BigEndianType().size() : float64Type().size()); // datatype
            s += 4 + 2; // fill value
            s += 4 + 18; // layout
        } else {
            s += 4 + 18; // link info
            s += 4 + 2; // group info
            for (const std::string& c : n.children)
                s += 4 + 3 + c.size() + 8;
        }
        for (const auto& a : attrs.at(path))
            s += 4 + attributeBody(a).size();
        return s;
    };
    auto storageBytes = [](const Node& n) { return n.isDimension ? 4 * n.dims[0] : 8 * (std::uint64_t)n.values.size(); };
    // addresses: superblock, all object headers (map order), the global heap collection, then the raw data, 8-byte aligned
    std::map<std::string, std::uint64_t> headerAt, dataAt;
    std::uint64_t at = 48;
    for (const auto& kv : m_nodes) {
        headerAt[kv.first] = at;
        const std::size_t hs = headerSize(kv.first, kv.second);
        if (hs > 0xffffffffu)
            throw Hdf5Error("object header too large");
        at += 10 + hs + 4;
    }
    std::uint64_t heapAt = 0, heapSize = 0;
    if (heapObjects > 0) {
        at = (at + 7) / 8 * 8;
        heapAt = at;
        heapSize = std::max<std::uint64_t>(4096, 16 + 24 * (std::uint64_t)heapObjects + 16);
        heapSize = (heapSize + 7) / 8 * 8;
        at += heapSize;
    }
    for (const auto& kv : m_nodes)
        if (kv.second.isDataset) {
            at = (at + 7) / 8 * 8;
            dataAt[kv.first] = at;
            at += storageBytes(kv.second);
        }
    const std::uint64_t eof = at;

    std::vector<unsigned char> f(SIGNATURE, SIGNATURE + 8);
    f.push_back(2), f.push_back(8), f.push_back(8), f.push_back(0);
    put(f, 0, 8), put(f, UNDEF, 8), put(f, eof, 8), put(f, headerAt.at("/"), 8);
    put(f, hdf5Checksum(f.data(), f.size()), 4);

    auto patch = [](std::vector<unsigned char>& b, std::size_t off, std::uint64_t v, int n) {
        for (int i = 0; i < n; ++i)
            b[off + i] = (unsigned char)(v >> (8 * i));
    };
    for (const auto& kv : m_nodes) {
        const Node& n = kv.second;
        const std::size_t hs = headerSize(kv.first, n);
        // chunk-0 size field: 1, 2 or 4 bytes (flags bits 0-1); 4 bytes always keeps the layout arithmetic simple
        std::vector<unsigned char> h = { 'O', 'H', 'D', 'R', 2, 0x02 };
        put(h, hs, 4);
        if (n.isDataset) {
            message(h, MSG_DATASPACE, simpleSpace(n.dims));
            message(h, MSG_DATATYPE, n.isDimension ? float32BigEndianType() : float64Type());
            message(h, MSG_FILL_VALUE, { 3, 0x09 }); // early allocation, fill written if set, no fill value defined
            std::vector<unsigned char> b = { 3, 1 };
            put(b, dataAt.at(kv.first), 8), put(b, storageBytes(n), 8);
            message(h, MSG_LAYOUT, b);
        } else {
            std::vector<unsigned char> b = { 0, 0 };
            put(b, UNDEF, 8), put(b, UNDEF, 8); // compact link storage: no heap, no name index
            message(h, MSG_LINK_INFO, b);
            message(h, MSG_GROUP_INFO, { 0, 0 });
            for (const std::string& c : n.children) {
                if (c.size() > 255)
                    throw Hdf5Error("link name too long: " + c);
                b = { 1, 0, (unsigned char)c.size() };
                b.insert(b.end(), c.begin(), c.end());
                put(b, headerAt.at(kv.first == "/" ? "/" + c : kv.first + "/" + c), 8);
                message(h, MSG_LINK, b);
            }
        }
        for (Attribute a : attrs.at(kv.first)) {
            for (const auto& r : a.objectRefs)
                patch(a.data, r.first, headerAt.at(r.second), 8);
            for (const auto& r : a.heapRefs) {
                patch(a.data, r.first, heapAt, 8);
                patch(a.data, r.first + 8, (std::uint64_t)r.second, 4);
            }
            message(h, MSG_ATTRIBUTE, attributeBody(a));
        }
        put(h, hdf5Checksum(h.data(), h.size()), 4);
        if (f.size() != headerAt.at(kv.first) || h.size() != 10 + hs + 4)
            throw Hdf5Error("internal layout error");
        f.insert(f.end(), h.begin(), h.end());
    }
    if (heapObjects > 0) { // global heap collection (III.E): the targets of the DIMENSION_LIST references
        f.resize((std::size_t)heapAt, 0);
        std::vector<unsigned char> g = { 'G', 'C', 'O', 'L', 1, 0, 0, 0 };
        put(g, heapSize, 8);
        for (int i = 0; i < heapObjects; ++i) {
            put(g, (std::uint64_t)(i + 1), 2), put(g, 1, 2), put(g, 0, 4); // index, reference count, reserved
            put(g, 8, 8); // object size
            put(g, headerAt.at(heapTargets[(std::size_t)i]), 8);
        }
        const std::uint64_t free = heapSize - g.size();
        put(g, 0, 2), put(g, 0, 2), put(g, 0, 4), put(g, free, 8); // object 0: the free space, its size includes this header
        g.resize((std::size_t)heapSize, 0);
        f.insert(f.end(), g.begin(), g.end());
    }
    for (const auto& kv : m_nodes)
        if (kv.second.isDataset) {
            f.resize((std::size_t)dataAt.at(kv.first), 0);
            const std::size_t at0 = f.size();
            f.resize(at0 + (std::size_t)storageBytes(kv.second), 0);
            if (!kv.second.isDimension && !kv.second.values.empty())
                std::memcpy(&f[at0], kv.second.values.data(), 8 * kv.second.values.size()); // host is little endian (x86-64)
        }
    std::ofstream out(filePath, std::ios::binary);
    if (!out || !out.write(reinterpret_cast<const char*>(f.data()), (std::streamsize)f.size()))
        throw Hdf5Error("cannot write " + filePath);
}

} // namespace Nextsim
