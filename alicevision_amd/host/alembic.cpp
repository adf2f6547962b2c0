// alembic.cpp — see alembic.hpp.  Part 1: the Ogawa / Alembic container (reader, writer); part 2: SfMData <-> Alembic, restating
// sfmDataIO/AlembicImporter.cpp and AlembicExporter.cpp of the reference for the data sfmData.hpp holds.
#include "alembic.hpp"

#include "log.hpp"
#include "sfmData.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace avdm_host {
namespace abc {

namespace {

constexpr uint64_t kDataBit = 1ull << 63;

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("Alembic: " + what); }

template <class T>
T loadLE(const uint8_t* p)
{
    T v;
    std::memcpy(&v, p, sizeof(T)); // the file is little-endian, and so is every host this program runs on
    return v;
}

// MurmurHash3_x64_128 (Austin Appleby, public domain), seed 0: the 16-byte digest in front of every sample.  The Alembic library keys
// its read cache on it, so two different samples must not carry the same digest — the writer computes it like the library does.
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
void murmur3_x64_128(const uint8_t* data, size_t len, uint8_t out[16])
{
    const uint64_t c1 = 0x87c37b91114253d5ull, c2 = 0x4cf5ad432745937full;
    uint64_t h1 = 0, h2 = 0;
    const size_t nblocks = len / 16;
    for(size_t i = 0; i < nblocks; ++i)
    {
        uint64_t k1 = loadLE<uint64_t>(data + 16 * i), k2 = loadLE<uint64_t>(data + 16 * i + 8);
        k1 *= c1, k1 = rotl64(k1, 31), k1 *= c2, h1 ^= k1;
        h1 = rotl64(h1, 27), h1 += h2, h1 = h1 * 5 + 0x52dce729;
        k2 *= c2, k2 = rotl64(k2, 33), k2 *= c1, h2 ^= k2;
        h2 = rotl64(h2, 31), h2 += h1, h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t* tail = data + 16 * nblocks;
    const size_t t = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for(size_t i = t; i > 8; --i)
        k2 ^= (uint64_t)tail[i - 1] << (8 * (i - 9));
    if(t > 8)
        k2 *= c2, k2 = rotl64(k2, 33), k2 *= c1, h2 ^= k2;
    for(size_t i = std::min<size_t>(t, 8); i > 0; --i)
        k1 ^= (uint64_t)tail[i - 1] << (8 * (i - 1));
    if(t > 0)
        k1 *= c1, k1 = rotl64(k1, 31), k1 *= c2, h1 ^= k1;
    h1 ^= (uint64_t)len, h2 ^= (uint64_t)len;
    h1 += h2, h2 += h1;
    h1 = fmix64(h1), h2 = fmix64(h2);
    h1 += h2, h2 += h1;
    std::memcpy(out, &h1, 8);
    std::memcpy(out + 8, &h2, 8);
}

float halfToFloat(uint16_t h)
{
    const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float v;
    if(e == 0)
        v = std::ldexp((float)m, -24);
    else if(e == 31)
        v = m ? NAN : INFINITY;
    else
        v = std::ldexp((float)(m | 0x400u), (int)e - 25);
    return s ? -v : v;
}

} // namespace

size_t podBytes(Pod p)
{
    switch(p)
    {
        case Pod::Bool:
        case Pod::UInt8:
        case Pod::Int8: return 1;
        case Pod::UInt16:
        case Pod::Int16:
        case Pod::Float16: return 2;
        case Pod::UInt32:
        case Pod::Int32:
        case Pod::Float32: return 4;
        case Pod::UInt64:
        case Pod::Int64:
        case Pod::Float64: return 8;
        default: return 0;
    }
}

std::string metadataValue(const std::string& metadata, const std::string& key)
{
    size_t i = 0;
    while(i < metadata.size())
    {
        size_t e = metadata.find(';', i);
        if(e == std::string::npos)
            e = metadata.size();
        const size_t eq = metadata.find('=', i);
        if(eq != std::string::npos && eq < e && metadata.compare(i, eq - i, key) == 0)
            return metadata.substr(eq + 1, e - eq - 1);
        i = e + 1;
    }
    return "";
}
std::string Object::meta(const std::string& key) const { return metadataValue(metadata, key); }

// ---- reader -------------------------------------------------------------------------------------------------------------------------
Archive::Archive(const std::string& filename)
{
    std::ifstream f(filename, std::ios::binary);
    if(!f)
        fail("cannot open '" + filename + "'");
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    _bytes.resize((size_t)std::max<std::streamoff>(n, 0));
    if(!_bytes.empty())
        f.read(reinterpret_cast<char*>(_bytes.data()), n);
    if(_bytes.size() >= 8 && std::memcmp(_bytes.data(), "\x89HDF\r\n\x1a\n", 8) == 0)
        fail("'" + filename + "' is an HDF5 Alembic archive (written before Alembic 1.5 / with the HDF5 back end); only Ogawa archives are read");
    if(_bytes.size() < 16 || std::memcmp(_bytes.data(), "Ogawa", 5) != 0)
        fail("'" + filename + "' is not an Ogawa archive");
    if(_bytes[5] != 0xff)
        fail("'" + filename + "' was not closed by its writer (frozen flag not set)");
    const std::vector<Node> root = group(u64At(8));
    if(root.size() < 6 || !root[1].isData || root[2].isData || !root[5].isData)
        fail("unexpected root group");
    const auto ver = data(root[1].pos);
    _libraryVersion = ver.second >= 4 ? loadLE<int32_t>(ver.first) : 0;
    _topPos = root[2].pos;
    const auto im = data(root[5].pos);
    _indexedMetadata.push_back("");
    for(size_t i = 0; i < im.second;)
    {
        const size_t len = im.first[i];
        if(i + 1 + len > im.second)
            fail("truncated indexed metadata");
        _indexedMetadata.emplace_back(reinterpret_cast<const char*>(im.first + i + 1), len);
        i += 1 + len;
    }
}

uint64_t Archive::u64At(uint64_t pos) const
{
    if(pos + 8 > _bytes.size())
        fail("position past the end of the file");
    return loadLE<uint64_t>(_bytes.data() + pos);
}

std::vector<Node> Archive::group(uint64_t pos) const
{
    std::vector<Node> out;
    if(pos == 0)
        return out;
    const uint64_t n = u64At(pos);
    if(n > (_bytes.size() - pos) / 8)
        fail("group larger than the file");
    out.resize((size_t)n);
    for(uint64_t i = 0; i < n; ++i)
    {
        const uint64_t c = u64At(pos + 8 + 8 * i);
        out[(size_t)i].isData = (c & kDataBit) != 0;
        out[(size_t)i].pos = c & ~kDataBit;
    }
    return out;
}

std::pair<const uint8_t*, size_t> Archive::data(uint64_t pos) const
{
    if(pos == 0)
        return {_bytes.data(), 0};
    const uint64_t n = u64At(pos);
    if(n > _bytes.size() - pos - 8)
        fail("data larger than the file");
    return {_bytes.data() + pos + 8, (size_t)n};
}

Object Archive::top() const
{
    Object o;
    o.name = "ABC";
    o.pos = _topPos;
    return o;
}

std::vector<Object> Archive::children(const Object& o) const
{
    std::vector<Object> out;
    const std::vector<Node> k = group(o.pos);
    if(k.size() < 2 || !k.back().isData)
        return out;
    auto hdr = data(k.back().pos);
    if(hdr.second < 32)
        return out;
    const size_t end = hdr.second - 32; // the two hashes
    size_t i = 0, ci = 1;
    while(i < end)
    {
        if(i + 4 > end)
            fail("truncated object header");
        const uint32_t n = loadLE<uint32_t>(hdr.first + i);
        i += 4;
        if(i + n + 1 > end)
            fail("truncated object header");
        Object c;
        c.name.assign(reinterpret_cast<const char*>(hdr.first + i), n);
        i += n;
        const uint8_t mi = hdr.first[i++];
        if(mi == 0xff)
        {
            if(i + 4 > end)
                fail("truncated object header");
            const uint32_t m = loadLE<uint32_t>(hdr.first + i);
            i += 4;
            if(i + m > end)
                fail("truncated object header");
            c.metadata.assign(reinterpret_cast<const char*>(hdr.first + i), m);
            i += m;
        }
        else
        {
            if(mi >= _indexedMetadata.size())
                fail("metadata index out of range");
            c.metadata = _indexedMetadata[mi];
        }
        if(ci + 1 >= k.size() || k[ci].isData)
            fail("object header lists more children than the group holds");
        c.pos = k[ci++].pos;
        out.push_back(std::move(c));
    }
    return out;
}

bool Archive::child(const Object& o, const std::string& name, Object& out) const
{
    for(Object& c : children(o))
        if(c.name == name)
        {
            out = std::move(c);
            return true;
        }
    return false;
}

std::vector<PropertyHeader> Archive::propertiesAt(uint64_t compoundPos) const
{
    std::vector<PropertyHeader> out;
    const std::vector<Node> k = group(compoundPos);
    if(k.empty() || !k.back().isData)
        return out;
    const auto hdr = data(k.back().pos);
    size_t i = 0, ci = 0;
    auto need = [&](size_t n) {
        if(i + n > hdr.second)
            fail("truncated property header");
    };
    auto rd = [&](int sizeHint) -> uint32_t {
        uint32_t v;
        if(sizeHint == 0)
        {
            need(1);
            v = hdr.first[i];
            i += 1;
        }
        else if(sizeHint == 1)
        {
            need(2);
            v = loadLE<uint16_t>(hdr.first + i);
            i += 2;
        }
        else
        {
            need(4);
            v = loadLE<uint32_t>(hdr.first + i);
            i += 4;
        }
        return v;
    };
    while(i < hdr.second)
    {
        need(4);
        const uint32_t info = loadLE<uint32_t>(hdr.first + i);
        i += 4;
        PropertyHeader h;
        const int kind = info & 3, sizeHint = (info >> 2) & 3;
        h.kind = kind == 0 ? PropertyHeader::Compound : (kind == 1 ? PropertyHeader::Scalar : PropertyHeader::Array);
        if(kind != 0)
        {
            const int pod = (info >> 4) & 0xf;
            h.pod = pod <= (int)Pod::WString ? (Pod)pod : Pod::Unknown;
            h.extent = (info >> 12) & 0xff;
            h.nextSampleIndex = rd(sizeHint);
            if(info & 0x200)
            {
                h.firstChangedIndex = rd(sizeHint);
                h.lastChangedIndex = rd(sizeHint);
            }
            else if(info & 0x800)
                h.firstChangedIndex = h.lastChangedIndex = 0;
            else
            {
                h.firstChangedIndex = 1;
                h.lastChangedIndex = h.nextSampleIndex ? h.nextSampleIndex - 1 : 0;
            }
            if(info & 0x100)
                h.timeSamplingIndex = rd(sizeHint);
        }
        const uint32_t n = rd(sizeHint);
        need(n);
        h.name.assign(reinterpret_cast<const char*>(hdr.first + i), n);
        i += n;
        const uint32_t mi = (info >> 20) & 0xff;
        if(mi == 0xff)
        {
            const uint32_t m = rd(sizeHint);
            need(m);
            h.metadata.assign(reinterpret_cast<const char*>(hdr.first + i), m);
            i += m;
        }
        else
        {
            if(mi >= _indexedMetadata.size())
                fail("metadata index out of range");
            h.metadata = _indexedMetadata[mi];
        }
        if(ci + 1 >= k.size())
            fail("property header lists more properties than the group holds");
        h.node = k[ci++];
        out.push_back(std::move(h));
    }
    return out;
}

std::vector<PropertyHeader> Archive::properties(const Object& o) const
{
    const std::vector<Node> k = group(o.pos);
    if(k.empty() || k[0].isData)
        return {};
    return propertiesAt(k[0].pos);
}

std::vector<PropertyHeader> Archive::properties(const PropertyHeader& compound) const
{
    if(compound.kind != PropertyHeader::Compound || compound.node.isData)
        return {};
    return propertiesAt(compound.node.pos);
}

const PropertyHeader* Archive::find(const std::vector<PropertyHeader>& props, const std::string& name)
{
    for(const PropertyHeader& p : props)
        if(p.name == name)
            return &p;
    return nullptr;
}

std::pair<const uint8_t*, size_t> Archive::sampleBytes(const PropertyHeader& p, size_t sample) const
{
    if(p.kind == PropertyHeader::Compound)
        fail("'" + p.name + "' is a compound property");
    if(sample >= p.nextSampleIndex)
        fail("'" + p.name + "': sample " + std::to_string(sample) + " of " + std::to_string(p.nextSampleIndex));
    // samples before the first change share sample 0, samples after the last change share the last stored one
    size_t stored;
    if(sample < p.firstChangedIndex || (p.firstChangedIndex == 0 && p.lastChangedIndex == 0))
        stored = 0;
    else
        stored = std::min<size_t>(sample, p.lastChangedIndex) - p.firstChangedIndex + 1;
    const std::vector<Node> k = group(p.node.pos);
    const size_t ci = p.kind == PropertyHeader::Array ? 2 * stored : stored;
    if(p.node.isData || ci >= k.size() || !k[ci].isData)
        fail("'" + p.name + "': sample not stored");
    auto d = data(k[ci].pos);
    if(d.second == 0)
        return d;
    if(d.second < 16)
        fail("'" + p.name + "': sample shorter than its digest");
    return {d.first + 16, d.second - 16};
}

std::vector<double> Archive::readDoubles(const PropertyHeader& p, size_t sample) const
{
    const auto d = sampleBytes(p, sample);
    const size_t sz = podBytes(p.pod);
    if(sz == 0)
        fail("'" + p.name + "' does not hold numbers");
    const size_t n = d.second / sz;
    std::vector<double> out(n);
    for(size_t i = 0; i < n; ++i)
    {
        const uint8_t* q = d.first + i * sz;
        switch(p.pod)
        {
            case Pod::Bool: out[i] = *q ? 1.0 : 0.0; break;
            case Pod::UInt8: out[i] = *q; break;
            case Pod::Int8: out[i] = (int8_t)*q; break;
            case Pod::UInt16: out[i] = loadLE<uint16_t>(q); break;
            case Pod::Int16: out[i] = loadLE<int16_t>(q); break;
            case Pod::UInt32: out[i] = loadLE<uint32_t>(q); break;
            case Pod::Int32: out[i] = loadLE<int32_t>(q); break;
            case Pod::UInt64: out[i] = (double)loadLE<uint64_t>(q); break;
            case Pod::Int64: out[i] = (double)loadLE<int64_t>(q); break;
            case Pod::Float16: out[i] = halfToFloat(loadLE<uint16_t>(q)); break;
            case Pod::Float32: out[i] = loadLE<float>(q); break;
            default: out[i] = loadLE<double>(q); break;
        }
    }
    return out;
}

std::vector<float> Archive::readFloats(const PropertyHeader& p, size_t sample) const
{
    if(p.pod == Pod::Float32)
    {
        const auto d = sampleBytes(p, sample);
        std::vector<float> out(d.second / 4);
        if(!out.empty())
            std::memcpy(out.data(), d.first, out.size() * 4);
        return out;
    }
    const std::vector<double> v = readDoubles(p, sample);
    return std::vector<float>(v.begin(), v.end());
}

std::vector<uint64_t> Archive::readUInts(const PropertyHeader& p, size_t sample) const
{
    const auto d = sampleBytes(p, sample);
    const size_t sz = podBytes(p.pod);
    if(sz == 0 || p.pod == Pod::Float16 || p.pod == Pod::Float32 || p.pod == Pod::Float64)
        fail("'" + p.name + "' does not hold integers");
    const size_t n = d.second / sz;
    std::vector<uint64_t> out(n);
    for(size_t i = 0; i < n; ++i)
    {
        const uint8_t* q = d.first + i * sz;
        switch(p.pod)
        {
            case Pod::Bool: out[i] = *q ? 1 : 0; break;
            case Pod::UInt8: out[i] = *q; break;
            case Pod::Int8: out[i] = (uint64_t)(int64_t)(int8_t)*q; break;
            case Pod::UInt16: out[i] = loadLE<uint16_t>(q); break;
            case Pod::Int16: out[i] = (uint64_t)(int64_t)loadLE<int16_t>(q); break;
            case Pod::UInt32: out[i] = loadLE<uint32_t>(q); break;
            case Pod::Int32: out[i] = (uint64_t)(int64_t)loadLE<int32_t>(q); break;
            case Pod::UInt64: out[i] = loadLE<uint64_t>(q); break;
            default: out[i] = (uint64_t)loadLE<int64_t>(q); break;
        }
    }
    return out;
}

std::vector<std::string> Archive::readStrings(const PropertyHeader& p, size_t sample) const
{
    if(p.pod != Pod::String)
        fail("'" + p.name + "' does not hold strings");
    const auto d = sampleBytes(p, sample);
    std::vector<std::string> out;
    size_t b = 0;
    for(size_t i = 0; i < d.second; ++i)
        if(d.first[i] == 0)
        {
            out.emplace_back(reinterpret_cast<const char*>(d.first + b), i - b);
            b = i + 1;
        }
    if(b < d.second)
        out.emplace_back(reinterpret_cast<const char*>(d.first + b), d.second - b);
    return out;
}

// ---- writer -------------------------------------------------------------------------------------------------------------------------
namespace {

template <class T>
void append(std::vector<uint8_t>& v, T x)
{
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&x);
    v.insert(v.end(), p, p + sizeof(T));
}
template <class T>
std::vector<uint8_t> rawBytes(const std::vector<T>& v)
{
    std::vector<uint8_t> out(v.size() * sizeof(T));
    if(!out.empty())
        std::memcpy(out.data(), v.data(), out.size());
    return out;
}

OutProperty leaf(PropertyHeader::Kind kind, Pod pod, int extent, const std::string& name, std::vector<uint8_t> bytes, size_t count,
                 const std::string& metadata = "")
{
    OutProperty p;
    p.kind = kind;
    p.pod = pod;
    p.extent = extent;
    p.name = name;
    p.metadata = metadata;
    p.sample = std::move(bytes);
    p.count = count;
    return p;
}

class Writer
{
public:
    std::vector<uint8_t> out;
    std::vector<std::string> indexed; // metadata table (index 0 = "")
    std::map<std::string, uint64_t> sampleAt; // digest -> position: equal samples are stored once, like the library does

    Writer()
    {
        out.assign({'O', 'g', 'a', 'w', 'a', 0x00, 0x00, 0x01});
        append<uint64_t>(out, 0); // root group position, patched by finish()
        indexed.push_back("");
    }
    uint64_t writeData(const uint8_t* p, size_t n)
    {
        if(n == 0)
            return kDataBit; // the empty data
        const uint64_t pos = out.size();
        append<uint64_t>(out, (uint64_t)n);
        out.insert(out.end(), p, p + n);
        return pos | kDataBit;
    }
    uint64_t writeData(const std::vector<uint8_t>& v) { return writeData(v.data(), v.size()); }
    uint64_t writeGroup(const std::vector<uint64_t>& children)
    {
        if(children.empty())
            return 0; // the empty group
        const uint64_t pos = out.size();
        append<uint64_t>(out, (uint64_t)children.size());
        for(uint64_t c : children)
            append<uint64_t>(out, c);
        return pos;
    }
    uint64_t writeSample(const std::vector<uint8_t>& raw)
    {
        if(raw.empty())
            return kDataBit;
        std::vector<uint8_t> blob(16 + raw.size());
        murmur3_x64_128(raw.data(), raw.size(), blob.data());
        std::memcpy(blob.data() + 16, raw.data(), raw.size());
        const std::string key(reinterpret_cast<const char*>(blob.data()), 16);
        const auto it = sampleAt.find(key);
        if(it != sampleAt.end() && data_equal(it->second, blob))
            return it->second;
        const uint64_t pos = writeData(blob);
        sampleAt[key] = pos;
        return pos;
    }
    int metadataIndex(const std::string& m)
    {
        if(m.empty())
            return 0;
        for(size_t i = 1; i < indexed.size(); ++i)
            if(indexed[i] == m)
                return (int)i;
        if(m.size() > 255 || indexed.size() >= 254)
            return 0xff; // written inline
        indexed.push_back(m);
        return (int)indexed.size() - 1;
    }
    // a property: returns its child entry (group position, or data for nothing) and appends its header to `headers`
    uint64_t writeProperty(const OutProperty& p, std::vector<uint8_t>& headers)
    {
        uint64_t node;
        uint32_t info;
        const int mi = metadataIndex(p.metadata);
        const int sizeHint = (p.name.size() > 255 || p.metadata.size() > 255) ? 2 : 0;
        if(p.kind == PropertyHeader::Compound)
        {
            node = writeCompound(p);
            info = 0;
        }
        else
        {
            std::vector<uint64_t> kids;
            kids.push_back(writeSample(p.sample));
            if(p.kind == PropertyHeader::Array)
            {
                // rank-1 dimensions are implied by the size of the data; an EMPTY string array states its dimension (0) like the
                // library does (the data alone cannot tell "no strings" from "one empty string")
                if(p.pod == Pod::String && p.count == 0)
                {
                    std::vector<uint8_t> dims;
                    append<uint64_t>(dims, 0);
                    kids.push_back(writeData(dims));
                }
                else
                    kids.push_back(kDataBit);
            }
            node = writeGroup(kids);
            info = (p.kind == PropertyHeader::Scalar ? 1u : 2u) | ((uint32_t)p.pod << 4) | 0x800u /* one sample: constant */ |
                   (((uint32_t)p.extent & 0xffu) << 12);
            if(p.kind == PropertyHeader::Scalar || p.extent == 1)
                info |= 0x400u; // homogeneous, as the library marks these
        }
        info |= (uint32_t)sizeHint << 2;
        info |= (uint32_t)mi << 20;
        append<uint32_t>(headers, info);
        auto put = [&](uint32_t v) {
            if(sizeHint == 0)
                headers.push_back((uint8_t)v);
            else
                append<uint32_t>(headers, v);
        };
        if(p.kind != PropertyHeader::Compound)
            put(1); // next sample index
        put((uint32_t)p.name.size());
        headers.insert(headers.end(), p.name.begin(), p.name.end());
        if(mi == 0xff)
        {
            put((uint32_t)p.metadata.size());
            headers.insert(headers.end(), p.metadata.begin(), p.metadata.end());
        }
        return node;
    }
    uint64_t writeCompound(const OutProperty& c)
    {
        if(c.children.empty())
            return 0;
        std::vector<uint64_t> kids;
        std::vector<uint8_t> headers;
        for(const OutProperty& p : c.children)
            kids.push_back(writeProperty(p, headers));
        kids.push_back(writeData(headers));
        return writeGroup(kids);
    }
    uint64_t writeObject(const OutObject& o)
    {
        std::vector<uint64_t> kids;
        kids.push_back(writeCompound(o.properties));
        std::vector<uint8_t> hdr;
        for(const OutObject& c : o.children)
        {
            kids.push_back(writeObject(c));
            append<uint32_t>(hdr, (uint32_t)c.name.size());
            hdr.insert(hdr.end(), c.name.begin(), c.name.end());
            const int mi = metadataIndex(c.metadata);
            hdr.push_back((uint8_t)mi);
            if(mi == 0xff)
            {
                append<uint32_t>(hdr, (uint32_t)c.metadata.size());
                hdr.insert(hdr.end(), c.metadata.begin(), c.metadata.end());
            }
        }
        // the two hashes (of the properties and of the children): readers do not check them; a digest of the header text keeps the
        // field from being all zero for two different objects
        uint8_t h[16];
        murmur3_x64_128(hdr.data(), hdr.size(), h);
        hdr.insert(hdr.end(), h, h + 16);
        hdr.insert(hdr.end(), 16, 0);
        kids.push_back(writeData(hdr));
        return writeGroup(kids);
    }

private:
    bool data_equal(uint64_t child, const std::vector<uint8_t>& blob) const
    {
        const uint64_t pos = child & ~kDataBit;
        if(pos + 8 + blob.size() > out.size() || loadLE<uint64_t>(out.data() + pos) != blob.size())
            return false;
        return std::memcmp(out.data() + pos + 8, blob.data(), blob.size()) == 0;
    }
};

} // namespace

OutProperty OutProperty::compound(const std::string& name, const std::string& metadata)
{
    OutProperty p;
    p.name = name;
    p.metadata = metadata;
    return p;
}
OutProperty OutProperty::scalarBool(const std::string& name, bool v) { return leaf(PropertyHeader::Scalar, Pod::Bool, 1, name, {(uint8_t)(v ? 1 : 0)}, 1); }
OutProperty OutProperty::scalarUInt32(const std::string& name, uint32_t v)
{
    return leaf(PropertyHeader::Scalar, Pod::UInt32, 1, name, rawBytes(std::vector<uint32_t>{v}), 1);
}
OutProperty OutProperty::scalarUInt16(const std::string& name, uint16_t v)
{
    return leaf(PropertyHeader::Scalar, Pod::UInt16, 1, name, rawBytes(std::vector<uint16_t>{v}), 1);
}
OutProperty OutProperty::scalarDouble(const std::string& name, double v)
{
    return leaf(PropertyHeader::Scalar, Pod::Float64, 1, name, rawBytes(std::vector<double>{v}), 1);
}
OutProperty OutProperty::scalarString(const std::string& name, const std::string& v)
{
    std::vector<uint8_t> b(v.begin(), v.end());
    b.push_back(0);
    return leaf(PropertyHeader::Scalar, Pod::String, 1, name, std::move(b), 1);
}
OutProperty OutProperty::scalarDoubles(const std::string& name, const std::vector<double>& v, const std::string& metadata)
{
    return leaf(PropertyHeader::Scalar, Pod::Float64, (int)v.size(), name, rawBytes(v), 1, metadata);
}
OutProperty OutProperty::scalarBytes(const std::string& name, const std::vector<uint8_t>& v)
{
    return leaf(PropertyHeader::Scalar, Pod::UInt8, (int)v.size(), name, v, 1);
}
OutProperty OutProperty::arrayUInt32(const std::string& name, const std::vector<uint32_t>& v)
{
    return leaf(PropertyHeader::Array, Pod::UInt32, 1, name, rawBytes(v), v.size());
}
OutProperty OutProperty::arrayUInt64(const std::string& name, const std::vector<uint64_t>& v)
{
    return leaf(PropertyHeader::Array, Pod::UInt64, 1, name, rawBytes(v), v.size());
}
OutProperty OutProperty::arrayDouble(const std::string& name, const std::vector<double>& v)
{
    return leaf(PropertyHeader::Array, Pod::Float64, 1, name, rawBytes(v), v.size());
}
OutProperty OutProperty::arrayFloat(const std::string& name, const std::vector<float>& v, int extent, const std::string& metadata)
{
    return leaf(PropertyHeader::Array, Pod::Float32, extent, name, rawBytes(v), v.size() / (size_t)std::max(extent, 1), metadata);
}
OutProperty OutProperty::arrayString(const std::string& name, const std::vector<std::string>& v)
{
    std::vector<uint8_t> b;
    for(const std::string& s : v)
    {
        b.insert(b.end(), s.begin(), s.end());
        b.push_back(0);
    }
    return leaf(PropertyHeader::Array, Pod::String, 1, name, std::move(b), v.size());
}

void save(const OutObject& top, const std::string& filename, const std::string& archiveMetadata)
{
    Writer w;
    const uint64_t topPos = w.writeObject(top);
    std::vector<uint8_t> v0, v1, ts, im;
    append<int32_t>(v0, 0);
    append<int32_t>(v1, 10804); // the layout written here is that of Alembic 1.8.4's Ogawa back end
    // time samplings: the one default sampling { max samples 1, time per cycle 1.0, 1 stored time: 0.0 }
    append<uint32_t>(ts, 1);
    append<double>(ts, 1.0);
    append<uint32_t>(ts, 1);
    append<double>(ts, 0.0);
    for(size_t i = 1; i < w.indexed.size(); ++i)
    {
        im.push_back((uint8_t)w.indexed[i].size());
        im.insert(im.end(), w.indexed[i].begin(), w.indexed[i].end());
    }
    std::vector<uint64_t> root;
    root.push_back(w.writeData(v0));
    root.push_back(w.writeData(v1));
    root.push_back(topPos);
    root.push_back(w.writeData(reinterpret_cast<const uint8_t*>(archiveMetadata.data()), archiveMetadata.size()));
    root.push_back(w.writeData(ts));
    root.push_back(w.writeData(im));
    const uint64_t rootPos = w.writeGroup(root);
    std::memcpy(w.out.data() + 8, &rootPos, 8);
    w.out[5] = 0xff; // frozen: the archive is complete
    std::ofstream f(filename, std::ios::binary | std::ios::trunc);
    if(!f)
        fail("cannot write '" + filename + "'");
    f.write(reinterpret_cast<const char*>(w.out.data()), (std::streamsize)w.out.size());
    if(!f)
        fail("write to '" + filename + "' failed");
}

} // namespace abc

// =====================================================================================================================================
// Part 2: SfMData <-> Alembic
// =====================================================================================================================================
namespace {

using abc::Archive;
using abc::Object;
using abc::PropertyHeader;

struct Version3
{
    int v[3] = {0, 0, 0};
    bool operator<(const Version3& o) const { return std::lexicographical_compare(v, v + 3, o.v, o.v + 3); }
    bool operator>=(const Version3& o) const { return !(*this < o); }
};
constexpr Version3 kIoVersion{{1, 2, 11}}; // sfmDataIO/sfmDataIO.hpp:13-15

// Imath::M44d as the importer uses it: x[row][col], row vectors (a point transforms as p * M)
struct M44
{
    double x[4][4];
    M44()
    {
        for(int i = 0; i < 4; ++i)
            for(int j = 0; j < 4; ++j)
                x[i][j] = i == j ? 1.0 : 0.0;
    }
    M44 operator*(const M44& o) const
    {
        M44 r;
        for(int i = 0; i < 4; ++i)
            for(int j = 0; j < 4; ++j)
            {
                double s = 0.0;
                for(int k = 0; k < 4; ++k)
                    s += x[i][k] * o.x[k][j];
                r.x[i][j] = s;
            }
        return r;
    }
};

// general 4 x 4 inverse by cofactors (the importer calls Eigen's Matrix4d::inverse(), which is this expansion for fixed size 4)
bool invert4(const double m[16], double inv[16])
{
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if(det == 0.0)
        return false;
    const double idet = 1.0 / det;
    for(int i = 0; i < 16; ++i)
        inv[i] *= idet;
    return true;
}

// The camera pose of an accumulated Alembic matrix (AlembicImporter.cpp:759-781 for cameras, :938-961 for rig / pose nodes): transpose
// to the column-vector convention, flip Y and Z (computer graphics -> computer vision), invert, and read (R, C) off [R | -R C].
Pose poseFromMatrix(const M44& mat, const Version3& abcVersion, bool cameraNode)
{
    double T[16];
    for(int i = 0; i < 4; ++i)
        for(int j = 0; j < 4; ++j)
            T[4 * i + j] = mat.x[j][i];
    const double sgn[4] = {1.0, -1.0, -1.0, 1.0};
    double A[16];
    for(int i = 0; i < 4; ++i)
        for(int j = 0; j < 4; ++j)
        {
            if(!(abcVersion < Version3{{1, 2, 3}}))
                A[4 * i + j] = sgn[i] * T[4 * i + j] * sgn[j]; // M * T * M
            else
                A[4 * i + j] = cameraNode ? T[4 * i + j] * sgn[j] : T[4 * i + j]; // T * M (camera) / T (pose node)
        }
    double T2[16];
    if(!invert4(A, T2))
        throw std::runtime_error("Alembic: singular camera transform");
    Pose p;
    for(int i = 0; i < 3; ++i)
        for(int j = 0; j < 3; ++j)
            p.rotation(i, j) = T2[4 * i + j];
    const Point3d t(T2[3], T2[7], T2[11]);
    // geometry::Pose3::center() = -R^T t
    p.center = Point3d(-(p.rotation(0, 0) * t.x + p.rotation(1, 0) * t.y + p.rotation(2, 0) * t.z),
                       -(p.rotation(0, 1) * t.x + p.rotation(1, 1) * t.y + p.rotation(2, 1) * t.z),
                       -(p.rotation(0, 2) * t.x + p.rotation(1, 2) * t.y + p.rotation(2, 2) * t.z));
    return p;
}

// camera/cameraCommon.hpp: number of parameters of each distortion model (Distortion*.hpp constructors)
int distortionParamCount(const std::string& type)
{
    static const std::map<std::string, int> counts = {{"none", 0},       {"radialk1", 1},      {"radialk3", 3},     {"radialk3pt", 3}, {"brown", 5},
                                                      {"fisheye", 4},    {"fisheye1", 1},      {"3deradial4", 6},   {"3declassicld", 5},
                                                      {"3deanamorphic4", 14}};
    const auto it = counts.find(type);
    return it == counts.end() ? -1 : it->second;
}

class Importer
{
public:
    Importer(const std::string& filename, SfMData& out) : _a(filename), _out(out) {}

    void run()
    {
        Object root;
        if(!_a.child(_a.top(), "mvgRoot", root))
            throw std::runtime_error("Alembic: no 'mvgRoot' object: not an AliceVision SfMData archive");
        const std::vector<PropertyHeader> props = _a.properties(root);
        if(const PropertyHeader* p = Archive::find(props, "mvg_ABC_version"))
        {
            const std::vector<uint64_t> v = _a.readUInts(*p);
            for(size_t i = 0; i < 3 && i < v.size(); ++i) // old files: major, minor only
                _version.v[i] = (int)v[i];
        }
        if(kIoVersion < _version)
            throw std::runtime_error("Alembic: file has a version more recent than this reader (1.2.11)");
        visit(_a.top(), M44(), true, 0);
    }

private:
    Archive _a;
    SfMData& _out;
    Version3 _version;

    // schema.getUserProperties() when it holds anything, else the arbitrary geometry parameters ("Maya always use ArbGeomParams
    // instead of user properties", AlembicImporter.cpp:93-104)
    std::vector<PropertyHeader> userProperties(const std::vector<PropertyHeader>& schemaProps) const
    {
        if(const PropertyHeader* up = Archive::find(schemaProps, ".userProperties"))
        {
            std::vector<PropertyHeader> u = _a.properties(*up);
            if(!u.empty())
                return u;
        }
        if(const PropertyHeader* ap = Archive::find(schemaProps, ".arbGeomParams"))
            return _a.properties(*ap);
        return {};
    }

    // getAbcProp: the element of a scalar property, or the first element of an array ("Maya transforms everything into arrays")
    bool uintProp(const std::vector<PropertyHeader>& props, const char* name, size_t frame, uint64_t& out) const
    {
        const PropertyHeader* p = Archive::find(props, name);
        if(!p)
            return false;
        const std::vector<uint64_t> v = _a.readUInts(*p, std::min(frame, p->numSamples() ? p->numSamples() - 1 : 0));
        if(v.empty())
            throw std::runtime_error(std::string("Alembic: property '") + name + "' is empty");
        out = v[0];
        return true;
    }
    bool indexProp(const std::vector<PropertyHeader>& props, const char* name, size_t frame, IndexT& out) const
    {
        uint64_t v;
        if(!uintProp(props, name, frame, v))
            return false;
        out = (IndexT)v;
        return true;
    }
    bool boolProp(const std::vector<PropertyHeader>& props, const char* name, size_t frame, bool& out) const
    {
        uint64_t v;
        if(!uintProp(props, name, frame, v))
            return false;
        out = v != 0;
        return true;
    }
    bool stringProp(const std::vector<PropertyHeader>& props, const char* name, size_t frame, std::string& out) const
    {
        const PropertyHeader* p = Archive::find(props, name);
        if(!p)
            return false;
        const std::vector<std::string> v = _a.readStrings(*p, std::min(frame, p->numSamples() ? p->numSamples() - 1 : 0));
        out = v.empty() ? std::string() : v[0];
        return true;
    }
    std::vector<double> doublesProp(const std::vector<PropertyHeader>& props, const char* name, size_t frame) const
    {
        const PropertyHeader* p = Archive::find(props, name);
        if(!p)
            return {};
        return _a.readDoubles(*p, std::min(frame, p->numSamples() ? p->numSamples() - 1 : 0));
    }

    // Alembic::AbcGeom::XformSample::getMatrix of one sample of an ".xform" compound: the operations in order, ret = op * ret
    bool xformMatrix(const std::vector<PropertyHeader>& xf, size_t sample, M44& out, size_t& numSamples) const
    {
        out = M44();
        numSamples = 1;
        const PropertyHeader* ops = Archive::find(xf, ".ops");
        const PropertyHeader* vals = Archive::find(xf, ".vals");
        if(const PropertyHeader* inh = Archive::find(xf, ".inherits"))
            numSamples = std::max<size_t>(inh->numSamples(), 1);
        if(!ops || !vals)
            return true; // constant identity: nothing stored
        numSamples = std::max<size_t>(vals->numSamples(), numSamples);
        const std::vector<uint64_t> op = _a.readUInts(*ops, std::min(sample, ops->numSamples() ? ops->numSamples() - 1 : 0));
        const std::vector<double> v = _a.readDoubles(*vals, std::min(sample, vals->numSamples() ? vals->numSamples() - 1 : 0));
        size_t at = 0;
        auto take = [&](size_t n) -> const double* {
            if(at + n > v.size())
                throw std::runtime_error("Alembic: xform sample holds fewer values than its operations need");
            const double* p = v.data() + at;
            at += n;
            return p;
        };
        for(uint64_t code : op)
        {
            M44 m;
            const int type = (int)(code >> 4) & 0xf;
            switch(type)
            {
                case 0: // scale
                {
                    const double* s = take(3);
                    m.x[0][0] = s[0], m.x[1][1] = s[1], m.x[2][2] = s[2];
                    break;
                }
                case 1: // translate
                {
                    const double* t = take(3);
                    m.x[3][0] = t[0], m.x[3][1] = t[1], m.x[3][2] = t[2];
                    break;
                }
                case 2: // rotate about an axis, degrees
                case 4:
                case 5:
                case 6:
                {
                    double ax[3] = {type == 4 ? 1.0 : 0.0, type == 5 ? 1.0 : 0.0, type == 6 ? 1.0 : 0.0}, deg;
                    if(type == 2)
                    {
                        const double* r = take(4);
                        ax[0] = r[0], ax[1] = r[1], ax[2] = r[2], deg = r[3];
                    }
                    else
                        deg = *take(1);
                    const double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
                    if(n > 0.0)
                    {
                        const double a = deg * M_PI / 180.0, c = std::cos(a), s = std::sin(a), X = ax[0] / n, Y = ax[1] / n, Z = ax[2] / n;
                        // Imath::Matrix44::setAxisAngle (row-vector convention)
                        m.x[0][0] = X * X * (1 - c) + c, m.x[0][1] = X * Y * (1 - c) + Z * s, m.x[0][2] = X * Z * (1 - c) - Y * s;
                        m.x[1][0] = X * Y * (1 - c) - Z * s, m.x[1][1] = Y * Y * (1 - c) + c, m.x[1][2] = Y * Z * (1 - c) + X * s;
                        m.x[2][0] = X * Z * (1 - c) + Y * s, m.x[2][1] = Y * Z * (1 - c) - X * s, m.x[2][2] = Z * Z * (1 - c) + c;
                    }
                    break;
                }
                case 3: // matrix: what AliceVision's exporter writes (XformSample::setMatrix)
                {
                    const double* q = take(16);
                    for(int i = 0; i < 4; ++i)
                        for(int j = 0; j < 4; ++j)
                            m.x[i][j] = q[4 * i + j];
                    break;
                }
                default: throw std::runtime_error("Alembic: unknown xform operation " + std::to_string(type));
            }
            out = m * out;
        }
        return true;
    }

    void visit(const Object& o, M44 mat, bool isReconstructed, int depth)
    {
        // AlembicImporter.cpp:982-1027 visitObject
        if(depth > 64)
            throw std::runtime_error("Alembic: object hierarchy deeper than 64 levels (a cycle in a damaged file?)");
        if(o.name == "mvgCamerasUndefined")
            isReconstructed = false;
        const std::string schema = o.meta("schema");
        const std::vector<Object> kids = _a.children(o);
        if(schema == "AbcGeom_Points_v1")
            readPointCloud(o);
        else if(schema == "AbcGeom_Xform_v3")
            readXform(o, kids, mat, isReconstructed);
        else if(schema == "AbcGeom_Camera_v1")
        {
            const std::vector<PropertyHeader> props = _a.properties(o);
            const PropertyHeader* geom = Archive::find(props, ".geom");
            const std::vector<PropertyHeader> g = geom ? _a.properties(*geom) : std::vector<PropertyHeader>();
            const PropertyHeader* core = Archive::find(g, ".core");
            if((core ? core->numSamples() : 1) == 1 && o.name.find("camera_ancestor") == std::string::npos) // ancestors: not held by SfMData here
                readCamera(g, mat, 0, isReconstructed);
        }
        for(const Object& c : kids)
            visit(c, mat, isReconstructed, depth + 1);
    }

    // AlembicImporter.cpp:860-980 readXform: accumulates the transform; a node with mvg_rigId / mvg_poseId is a pose (rig) node
    void readXform(const Object& o, const std::vector<Object>& kids, M44& mat, bool isReconstructed)
    {
        const std::vector<PropertyHeader> props = _a.properties(o);
        const PropertyHeader* xfp = Archive::find(props, ".xform");
        const std::vector<PropertyHeader> xf = xfp ? _a.properties(*xfp) : std::vector<PropertyHeader>();
        M44 X;
        size_t numSamples = 1;
        xformMatrix(xf, 0, X, numSamples);
        if(numSamples != 1)
        {
            // an animated camera: one view per sample, read from the first child
            if(kids.empty())
                return;
            const std::vector<PropertyHeader> cp = _a.properties(kids[0]);
            const PropertyHeader* geom = Archive::find(cp, ".geom");
            if(!geom)
                return;
            const std::vector<PropertyHeader> g = _a.properties(*geom);
            for(size_t frame = 0; frame < numSamples; ++frame)
            {
                size_t ns;
                xformMatrix(xf, frame, X, ns);
                readCamera(g, mat * X, frame, isReconstructed);
            }
            return;
        }
        mat = mat * X;
        const std::vector<PropertyHeader> up = userProperties(xf);
        IndexT rigId = UndefinedIndexT, poseId = UndefinedIndexT;
        indexProp(up, "mvg_rigId", 0, rigId);
        indexProp(up, "mvg_poseId", 0, poseId);
        if(rigId == UndefinedIndexT && poseId == UndefinedIndexT)
            return; // not a rig
        if(isReconstructed && !_out.poses.count(poseId))
            _out.poses[poseId] = poseFromMatrix(mat, _version, false);
        if(rigId != UndefinedIndexT && !_out.rigs.count(rigId))
        {
            uint64_t nbSubPoses = 0;
            uintProp(up, "mvg_nbSubPoses", 0, nbSubPoses);
            _out.rigs[rigId].subPoses.resize((size_t)nbSubPoses);
        }
        mat = M44(); // the cameras below a rig node carry their sub-poses
    }

    // AlembicImporter.cpp:375-812 readCamera (`g` = the properties of the camera's ".geom" compound)
    void readCamera(const std::vector<PropertyHeader>& g, const M44& mat, size_t frame, bool isReconstructed)
    {
        const std::vector<PropertyHeader> up = userProperties(g);
        View v;
        v.viewId = (IndexT)_out.views.size();
        v.poseId = (IndexT)_out.views.size();
        v.intrinsicId = (IndexT)_out.intrinsics.size();
        bool poseIndependant = true;
        std::string intrinsicType = "pinhole", distortionType = "none", undistortionType = "none";
        std::vector<double> sensorPix = {0, 0}, sensorMm = {0, 0};

        stringProp(up, "mvg_imagePath", frame, v.path);
        indexProp(up, "mvg_viewId", frame, v.viewId);
        indexProp(up, "mvg_poseId", frame, v.poseId);
        indexProp(up, "mvg_intrinsicId", frame, v.intrinsicId);
        indexProp(up, "mvg_rigId", frame, v.rigId);
        indexProp(up, "mvg_subPoseId", frame, v.subPoseId);
        boolProp(up, "mvg_poseIndependant", frame, poseIndependant);
        v.independantPose = poseIndependant;
        if(const PropertyHeader* p = Archive::find(up, "mvg_metadata"))
        {
            const std::vector<std::string> raw = _a.readStrings(*p, std::min(frame, p->numSamples() ? p->numSamples() - 1 : 0));
            if(raw.size() % 2)
                throw std::runtime_error("Alembic: 'mvg_metadata' holds an odd number of strings");
            for(size_t i = 0; i + 1 < raw.size(); i += 2)
                v.metadata[raw[i]] = raw[i + 1];
        }
        if(Archive::find(up, "mvg_sensorSizePix"))
        {
            sensorPix = doublesProp(up, "mvg_sensorSizePix", frame);
            if(sensorPix.size() != 2)
                throw std::runtime_error("Alembic: 'mvg_sensorSizePix' does not hold two values");
        }
        if(Archive::find(up, "mvg_sensorSizeMm"))
        {
            sensorMm = doublesProp(up, "mvg_sensorSizeMm", frame);
            if(sensorMm.size() != 2)
                throw std::runtime_error("Alembic: 'mvg_sensorSizeMm' does not hold two values");
        }
        stringProp(up, "mvg_intrinsicType", frame, intrinsicType);
        stringProp(up, "mvg_distortionType", frame, distortionType);
        stringProp(up, "mvg_undistortionType", frame, undistortionType);
        const std::vector<double> params = doublesProp(up, "mvg_intrinsicParams", frame);
        const std::vector<double> distortionParams = doublesProp(up, "mvg_distortionParams", frame);
        v.width = (int)sensorPix[0];
        v.height = (int)sensorPix[1];

        // the intrinsic (createIntrinsic + IntrinsicScaleOffset::importFromParams, camera/IntrinsicScaleOffset.cpp:126-158)
        {
            Intrinsic I;
            I.intrinsicId = v.intrinsicId;
            std::string type = intrinsicType;
            std::transform(type.begin(), type.end(), type.begin(), ::tolower);
            if(_version < Version3{{1, 2, 8}})
            { // camera/cameraCommon.hpp:206-270 compatibilityStringToEnums
                static const std::map<std::string, std::pair<const char*, const char*>> compat = {
                    {"pinhole", {"pinhole", "none"}},       {"radial1", {"pinhole", "radialk1"}},   {"radial3", {"pinhole", "radialk3"}},
                    {"brown", {"pinhole", "brown"}},        {"fisheye4", {"pinhole", "fisheye"}},   {"fisheye1", {"pinhole", "fisheye1"}},
                    {"3deanamorphic4", {"pinhole", "none"}}, {"equidistant", {"equidistant", "none"}}, {"equidistant_r3", {"equidistant", "radialk3pt"}}};
                const auto it = compat.find(type);
                if(it == compat.end())
                    throw std::runtime_error("Alembic: unknown intrinsic type '" + intrinsicType + "'");
                I.type = it->second.first;
                I.distortionType = it->second.second;
            }
            else
            {
                I.type = type;
                I.distortionType = distortionType;
                std::transform(I.distortionType.begin(), I.distortionType.end(), I.distortionType.begin(), ::tolower);
            }
            I.isPinhole = I.type == "pinhole";
            I.width = v.width;
            I.height = v.height;
            I.sensorWidth = sensorMm[0];
            I.sensorHeight = sensorMm[1];
            std::vector<double> pl = params;
            if(_version < Version3{{1, 2, 0}} && !params.empty())
            { // one focal length for both axes
                pl.assign(params.size() + 1, 0.0);
                pl[0] = pl[1] = params[0];
                for(size_t i = 1; i < params.size(); ++i)
                    pl[i + 1] = params[i];
            }
            const int nDisto = distortionParamCount(I.distortionType);
            if(nDisto > 0)
                I.distortionParams.assign((size_t)nDisto, 0.0);
            if(pl.size() >= 4)
            {
                I.scaleX = pl[0], I.scaleY = pl[1], I.offsetX = pl[2], I.offsetY = pl[3];
                if(nDisto > 0 && pl.size() == 4 + (size_t)nDisto) // old files keep the distortion behind the four pinhole parameters
                    I.distortionParams.assign(pl.begin() + 4, pl.end());
                if(_version < Version3{{1, 2, 1}})
                {
                    I.offsetX -= I.width / 2.0;
                    I.offsetY -= I.height / 2.0;
                }
            }
            if(nDisto > 0 && distortionParams.size() == (size_t)nDisto) // Distortion::setParameters ignores another size
                I.distortionParams = distortionParams;
            if(nDisto < 0)
                I.distortionParams = distortionParams;
            std::string undisto = undistortionType;
            std::transform(undisto.begin(), undisto.end(), undisto.begin(), ::tolower);
            const bool anyDisto = std::any_of(I.distortionParams.begin(), I.distortionParams.end(), [](double d) { return d != 0.0; });
            if((anyDisto && I.distortionType != "none" && I.distortionType != "radialk1" && I.distortionType != "radialk3") || undisto != "none")
                std::cerr << "[warning] intrinsic " << I.intrinsicId << ": distortion model '" << (undisto != "none" ? undisto : I.distortionType)
                          << "' is not restated; observations are used as undistorted for the view-angle tests." << std::endl;
            _out.intrinsics.emplace(I.intrinsicId, std::move(I)); // the first definition of an id stays (std::map::emplace in the importer)
        }

        const View key = v; // (ids for the pose bookkeeping below)
        _out.views.emplace(v.viewId, std::move(v));
        if(isReconstructed)
        {
            const Pose pose = poseFromMatrix(mat, _version, true);
            if(key.isPartOfRig() && !key.isPoseIndependant())
            {
                // AlembicImporter.cpp:783-795: below a rig node the camera's transform is its sub-pose; the first one read for an index stays
                Rig& rig = _out.rigs[key.rigId];
                if(key.subPoseId == UndefinedIndexT)
                    throw std::runtime_error("Alembic: rig camera of view " + std::to_string(key.viewId) + " has no sub-pose index");
                if(key.subPoseId >= rig.subPoses.size())
                    rig.subPoses.resize((size_t)key.subPoseId + 1);
                RigSubPose& sp = rig.subPoses[key.subPoseId];
                if(!sp.initialized)
                {
                    sp.initialized = true;
                    sp.pose = pose;
                }
            }
            else
                _out.poses[key.poseId] = pose; // SfMData::setPose: a view's own pose replaces what is there
        }
    }

    // AlembicImporter.cpp:153-373 readPointCloud
    void readPointCloud(const Object& o)
    {
        const std::vector<PropertyHeader> props = _a.properties(o);
        const PropertyHeader* geom = Archive::find(props, ".geom");
        if(!geom)
            return;
        const std::vector<PropertyHeader> g = _a.properties(*geom);
        const PropertyHeader* P = Archive::find(g, "P");
        if(!P)
            return;
        const std::vector<float> pos = _a.readFloats(*P);
        const size_t n = pos.size() / 3;
        const std::vector<PropertyHeader> up = userProperties(g);
        const IndexT first = (IndexT)_out.landmarks.size();
        const bool flip = !(_version < Version3{{1, 2, 3}});
        for(size_t i = 0; i < n; ++i)
        {
            Landmark& L = _out.landmarks[first + (IndexT)i];
            L = Landmark();
            L.X = flip ? Point3d(pos[3 * i], -pos[3 * i + 1], -pos[3 * i + 2]) : Point3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        }
        if(const PropertyHeader* arb = Archive::find(g, ".arbGeomParams"))
        {
            const std::vector<PropertyHeader> ap = _a.properties(*arb);
            if(const PropertyHeader* col = Archive::find(ap, "color"))
            {
                const std::vector<float> c = _a.readFloats(*col);
                if(c.size() == pos.size()) // "Colors will be ignored" otherwise (AlembicImporter.cpp:165-176)
                    for(size_t i = 0; i < n; ++i)
                        for(int k = 0; k < 3; ++k)
                            _out.landmarks[first + (IndexT)i].rgb[k] = (unsigned char)(c[3 * i + k] * 255.0f);
            }
        }
        const PropertyHeader* visSize = Archive::find(up, "mvg_visibilitySize");
        // files written before the per-observation arrays were split: (view id, feature id) pairs and (x, y) pairs
        if(visSize && Archive::find(up, "mvg_visibilityIds") && Archive::find(up, "mvg_visibilityFeatPos"))
        {
            const std::vector<uint64_t> sizes = _a.readUInts(*visSize), ids = _a.readUInts(*Archive::find(up, "mvg_visibilityIds"));
            const std::vector<float> xy = _a.readFloats(*Archive::find(up, "mvg_visibilityFeatPos"));
            if(sizes.size() != n)
                throw std::runtime_error("Alembic: number of observations per 3D point should be identical to the number of 2D features");
            if(ids.size() != xy.size())
                throw std::runtime_error("Alembic: visibility Ids and features 2D pos should have the same size");
            size_t at = 0;
            for(size_t i = 0; i < n; ++i)
                for(uint64_t k = 0; k < sizes[i]; ++k, at += 2)
                {
                    if(at + 1 >= ids.size())
                        throw std::runtime_error("Alembic: visibility arrays shorter than the visibility sizes say");
                    Observation& ob = _out.landmarks[first + (IndexT)i].observations[(IndexT)ids[at]];
                    ob.x = xy[at];
                    ob.y = xy[at + 1];
                }
        }
        if(visSize && Archive::find(up, "mvg_visibilityViewId"))
        {
            const std::vector<uint64_t> sizes = _a.readUInts(*visSize), viewIds = _a.readUInts(*Archive::find(up, "mvg_visibilityViewId"));
            if(sizes.size() != n)
                throw std::runtime_error("Alembic: number of observations per 3D point should be identical to the number of 2D features");
            std::vector<uint64_t> featIds;
            std::vector<float> xy;
            bool hasFeatures = false;
            if(Archive::find(up, "mvg_visibilityFeatId") && Archive::find(up, "mvg_visibilityFeatPos"))
            {
                featIds = _a.readUInts(*Archive::find(up, "mvg_visibilityFeatId"));
                xy = _a.readFloats(*Archive::find(up, "mvg_visibilityFeatPos"));
                if(viewIds.size() != featIds.size() || 2 * viewIds.size() != xy.size())
                    throw std::runtime_error("Alembic: visibility Ids and features id / 2D pos should have the same size");
                hasFeatures = !featIds.empty();
            }
            size_t at = 0;
            for(size_t i = 0; i < n; ++i)
                for(uint64_t k = 0; k < sizes[i]; ++k, ++at)
                {
                    if(at >= viewIds.size())
                        throw std::runtime_error("Alembic: visibility arrays shorter than the visibility sizes say");
                    Observation& ob = _out.landmarks[first + (IndexT)i].observations[(IndexT)viewIds[at]];
                    ob = Observation();
                    if(hasFeatures)
                    {
                        ob.x = xy[2 * at];
                        ob.y = xy[2 * at + 1];
                    }
                }
        }
    }
};

} // namespace

void loadSfMDataAlembic(SfMData& out, const std::string& filename)
{
    Importer(filename, out).run();
}

// ---- exporter -----------------------------------------------------------------------------------------------------------------------
namespace {

using abc::OutObject;
using abc::OutProperty;

const char* kXformMeta = "schema=AbcGeom_Xform_v3;schemaObjTitle=AbcGeom_Xform_v3:.xform";
const char* kCameraMeta = "schema=AbcGeom_Camera_v1;schemaObjTitle=AbcGeom_Camera_v1:.geom";
const char* kPointsMeta = "schema=AbcGeom_Points_v1;schemaBaseType=AbcGeom_GeomBase_v1;schemaObjTitle=AbcGeom_Points_v1:.geom";

OutObject xformObject(const std::string& name, const double* matrix /* 16 values, x[row][col], or nullptr = identity */)
{
    OutObject o(name, kXformMeta);
    OutProperty xf = OutProperty::compound(".xform", "schema=AbcGeom_Xform_v3");
    if(matrix)
    {
        xf.add(OutProperty::scalarBool(".inherits", true));
        xf.add(OutProperty::scalarBytes(".ops", {0x30})); // one matrix operation
        xf.add(OutProperty::scalarDoubles(".vals", std::vector<double>(matrix, matrix + 16)));
        bool identity = true;
        for(int i = 0; i < 16; ++i)
            identity = identity && matrix[i] == (i % 5 == 0 ? 1.0 : 0.0);
        if(!identity) // the library leaves the flag out of a transform whose every sample is the identity
            xf.add(OutProperty::scalarBool("isNotConstantIdentity", true));
    }
    o.properties.add(std::move(xf));
    return o;
}

OutProperty hidden()
{
    // Alembic::AbcGeom::CreateVisibilityProperty(..).set(kVisibilityHidden): an int8 scalar named "visible" holding 0
    OutProperty p;
    p.kind = PropertyHeader::Scalar;
    p.pod = abc::Pod::Int8;
    p.extent = 1;
    p.name = "visible";
    p.sample = {0};
    p.count = 1;
    return p;
}

} // namespace

void saveSfMDataAlembic(const SfMData& in, const std::string& filename, bool withViews, bool withObservations)
{
    // AlembicExporter.cpp:30-52: the hierarchy and the version properties
    OutObject top("ABC", "");
    OutObject root = xformObject("mvgRoot", nullptr);
    root.properties.add(OutProperty::arrayUInt32("mvg_ABC_version", {1, 2, 11}));
    root.properties.add(OutProperty::arrayUInt32("mvg_aliceVision_version", {3, 3, 0}));
    root.properties.add(OutProperty::arrayString("mvg_featuresFolders", {}));
    root.properties.add(OutProperty::arrayString("mvg_matchesFolders", {}));
    OutObject cameras = xformObject("mvgCameras", nullptr), undefined = xformObject("mvgCamerasUndefined", nullptr);
    undefined.properties.add(hidden());

    // the 4 x 4 an exporter node stores for a pose: T = [R | -R C], T2 = (M T M)^-1, transposed to Imath's row-vector convention
    auto poseMatrix = [](const Pose& p, IndexT viewId, double xm[16]) {
        const Point3d t = p.rotation * p.center * -1.0;
        const double sgn[4] = {1.0, -1.0, -1.0, 1.0};
        double A[16] = {0}, T2[16];
        for(int i = 0; i < 3; ++i)
        {
            for(int j = 0; j < 3; ++j)
                A[4 * i + j] = sgn[i] * p.rotation(i, j) * sgn[j];
            A[4 * i + 3] = sgn[i] * (i == 0 ? t.x : (i == 1 ? t.y : t.z));
        }
        A[15] = 1.0;
        if(!invert4(A, T2))
            throw std::runtime_error("Alembic: singular pose of view " + std::to_string(viewId));
        for(int i = 0; i < 4; ++i)
            for(int j = 0; j < 4; ++j)
                xm[4 * j + i] = T2[4 * i + j];
    };
    // AlembicExporter.cpp:93-312 addCamera: the transform node of a view with its camera below; `pose` = what the node's matrix states (the
    // view's own pose, or its sub-pose below a rig node), nullptr = none
    auto cameraNode = [&](const View& v, const Pose* pose, bool& complete) -> OutObject {
        const bool hasPose = pose != nullptr;
        const bool hasIntrinsic = v.intrinsicId != UndefinedIndexT && in.intrinsics.count(v.intrinsicId);
        complete = hasPose && hasIntrinsic;
        std::string stem = v.path;
        const size_t slash = stem.find_last_of("/\\");
        if(slash != std::string::npos)
            stem = stem.substr(slash + 1);
        const size_t dot = stem.rfind('.');
        if(dot != std::string::npos && dot > 0)
            stem = stem.substr(0, dot);
        std::ostringstream label;
        label << "camxform_" << std::setfill('0') << std::setw(5) << UndefinedIndexT << "_" << v.poseId << "_" << stem << "_" << v.viewId;
        double xm[16];
        if(hasPose)
            poseMatrix(*pose, v.viewId, xm);
        OutObject xf = xformObject(label.str(), hasPose ? xm : nullptr);
        OutObject cam("camera_" + label.str(), kCameraMeta);
        OutProperty geom = OutProperty::compound(".geom", "schema=AbcGeom_Camera_v1");
        OutProperty up = OutProperty::compound(".userProperties");
        if(hasPose)
            up.add(OutProperty::scalarBool("mvg_poseLocked", false));
        if(!v.path.empty())
            up.add(OutProperty::scalarString("mvg_imagePath", v.path));
        up.add(OutProperty::scalarUInt32("mvg_viewId", v.viewId));
        up.add(OutProperty::scalarUInt32("mvg_poseId", v.poseId));
        up.add(OutProperty::scalarUInt32("mvg_intrinsicId", v.intrinsicId));
        up.add(OutProperty::scalarUInt32("mvg_resectionId", UndefinedIndexT));
        if(v.isPartOfRig())
        {
            up.add(OutProperty::scalarUInt32("mvg_rigId", v.rigId));
            up.add(OutProperty::scalarUInt32("mvg_subPoseId", v.subPoseId));
        }
        if(!v.isPoseIndependant())
            up.add(OutProperty::scalarBool("mvg_poseIndependant", false));
        {
            std::vector<std::string> raw;
            for(const auto& m : v.metadata)
            {
                raw.push_back(m.first);
                raw.push_back(m.second);
            }
            up.add(OutProperty::arrayString("mvg_metadata", raw));
        }
        up.add(OutProperty::arrayUInt32("mvg_ancestorsParams", {}));
        // CameraSample (".core": 16 doubles): focal length [mm], apertures and film offsets [cm], lens squeeze ratio; the rest are
        // Alembic's defaults (overscan 0 x 4, f-stop 5.6, focus distance 5, shutter 0 / 0.020833.., clipping 0.1 / 100000)
        std::vector<double> core = {35.0, 3.6, 0.0, 2.4, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 5.6, 5.0, 0.0, 1.0 / 48.0, 0.1, 100000.0};
        if(hasIntrinsic)
        {
            const Intrinsic& I = in.intrinsics.at(v.intrinsicId);
            const float imgW = (float)I.width, imgH = (float)I.height, sensorW = (float)I.sensorWidth;
            const float sensorWpix = std::max(imgW, imgH), fx = (float)I.scaleX, fy = (float)I.scaleY, pix2mm = sensorW / sensorWpix;
            core[0] = sensorW * fx / sensorWpix;
            core[1] = (float)(0.1 * imgW * pix2mm);
            core[2] = (float)(0.1 * I.offsetX * pix2mm);
            core[3] = (float)(0.1 * imgH * pix2mm);
            core[4] = (float)(0.1 * (-I.offsetY) * pix2mm);
            core[5] = fx / fy;
            up.add(OutProperty::arrayUInt32("mvg_sensorSizePix", {(uint32_t)I.width, (uint32_t)I.height}));
            up.add(OutProperty::arrayDouble("mvg_sensorSizeMm", {I.sensorWidth, I.sensorHeight}));
            up.add(OutProperty::scalarString("mvg_intrinsicType", I.type));
            up.add(OutProperty::scalarString("mvg_intrinsicInitializationMode", "none"));
            up.add(OutProperty::scalarDouble("mvg_initialFocalLength", -1.0));
            up.add(OutProperty::scalarString("mvg_intrinsicSerialNumber", ""));
            up.add(OutProperty::scalarBool("mvg_intrinsicLocked", false));
            up.add(OutProperty::scalarBool("mvg_intrinsicPixelRatioLocked", true));
            up.add(OutProperty::scalarBool("mvg_intrinsicOffsetLocked", false));
            up.add(OutProperty::scalarBool("mvg_intrinsicScaleLocked", false));
            up.add(OutProperty::scalarString("mvg_intrinsicDistortionInitializationMode", "none"));
            up.add(OutProperty::arrayDouble("mvg_intrinsicParams", {I.scaleX, I.scaleY, I.offsetX, I.offsetY}));
            if(I.distortionType != "none" && !I.distortionType.empty())
            {
                up.add(OutProperty::arrayDouble("mvg_distortionParams", I.distortionParams));
                up.add(OutProperty::scalarBool("mvg_distortionLocked", false));
            }
            up.add(OutProperty::scalarString("mvg_distortionType", I.distortionType.empty() ? "none" : I.distortionType));
            up.add(OutProperty::scalarString("mvg_undistortionType", "none"));
        }
        geom.add(OutProperty::scalarDoubles(".core", core));
        geom.add(std::move(up));
        cam.properties.add(std::move(geom));
        if(!hasPose || !hasIntrinsic)
            xf.properties.add(hidden());
        xf.add(std::move(cam));
        return xf;
    };

    // AlembicExporter.cpp:364-393 addSfM: single cameras, then one rig node per (rig, rig pose)
    std::map<IndexT, std::map<IndexT, std::vector<IndexT>>> rigsViewIds;
    for(const auto& kv : in.views)
    {
        if(!withViews)
            break;
        const View& v = kv.second;
        if(v.isPartOfRig() && !v.isPoseIndependant())
        {
            rigsViewIds[v.rigId][v.poseId].push_back(v.viewId);
            continue;
        }
        // AlembicExporter.cpp:395-409 addSfMSingleCamera
        const bool posed = v.poseId != UndefinedIndexT && in.poses.count(v.poseId);
        const Pose own = posed ? in.poses.at(v.poseId) : Pose();
        bool complete = false;
        OutObject xf = cameraNode(v, posed ? &own : nullptr, complete);
        (complete && in.isPoseAndIntrinsicDefined(v) ? cameras : undefined).add(std::move(xf));
    }
    // AlembicExporter.cpp:411-486 addSfMCameraRig
    for(const auto& rigPair : rigsViewIds)
        for(const auto& poseViews : rigPair.second)
        {
            const IndexT rigId = rigPair.first, rigPoseId = poseViews.first;
            const auto rigIt = in.rigs.find(rigId);
            if(rigIt == in.rigs.end())
                throw std::runtime_error("Alembic: view of rig " + std::to_string(rigId) + " but no such rig");
            double xm[16];
            const bool rigPosed = in.poses.count(rigPoseId) != 0;
            if(rigPosed)
                poseMatrix(in.poses.at(rigPoseId), poseViews.second.front(), xm);
            std::ostringstream label;
            label << "rigxform_" << std::setfill('0') << std::setw(5) << rigId << "_" << rigPoseId;
            std::map<bool, OutObject> rigObj; // one node among the reconstructed cameras, one among the undefined ones
            for(const IndexT viewId : poseViews.second)
            {
                const View& v = in.views.at(viewId);
                const RigSubPose* sp = in.rigSubPose(v);
                const bool isReconstructed = sp != nullptr && sp->initialized;
                if(!rigObj.count(isReconstructed))
                {
                    OutObject node = xformObject(label.str(), rigPosed ? xm : nullptr);
                    OutProperty up = OutProperty::compound(".userProperties");
                    up.add(OutProperty::scalarUInt32("mvg_rigId", rigId));
                    up.add(OutProperty::scalarUInt32("mvg_poseId", rigPoseId));
                    up.add(OutProperty::scalarUInt16("mvg_nbSubPoses", (uint16_t)rigIt->second.subPoses.size()));
                    up.add(OutProperty::scalarBool("mvg_rigPoseLocked", false));
                    node.properties.children.front().add(std::move(up)); // under ".xform"
                    rigObj.emplace(isReconstructed, std::move(node));
                }
                bool complete = false;
                rigObj.at(isReconstructed).add(cameraNode(v, isReconstructed ? &sp->pose : nullptr, complete));
            }
            for(auto& kv : rigObj)
                (kv.first ? cameras : undefined).add(std::move(kv.second));
        }

    // AlembicExporter.cpp:488-600 addLandmarks
    OutObject cloud = xformObject("mvgCloud", nullptr), pointCloud = xformObject("mvgPointCloud", nullptr);
    if(!in.landmarks.empty())
    {
        std::vector<float> positions, colors, featPos, featScale;
        std::vector<uint64_t> ids;
        std::vector<uint32_t> descTypes, visSize, visView, visFeat;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for(const auto& kv : in.landmarks)
        {
            const Landmark& L = kv.second;
            const float p[3] = {(float)L.X.x, (float)-L.X.y, (float)-L.X.z}; // computer vision -> computer graphics
            for(int k = 0; k < 3; ++k)
            {
                positions.push_back(p[k]);
                colors.push_back(L.rgb[k] / 255.f);
                lo[k] = std::min<double>(lo[k], p[k]);
                hi[k] = std::max<double>(hi[k], p[k]);
            }
            ids.push_back(ids.size());
            descTypes.push_back(0); // feature::EImageDescriberType::UNKNOWN
            visSize.push_back((uint32_t)L.observations.size());
            for(const auto& ob : L.observations)
            {
                visView.push_back(ob.first);
                visFeat.push_back(UndefinedIndexT);
                featPos.push_back((float)ob.second.x);
                featPos.push_back((float)ob.second.y);
                featScale.push_back(0.0f);
            }
        }
        OutObject points("particleShape1", kPointsMeta);
        OutProperty geom = OutProperty::compound(".geom", "schema=AbcGeom_Points_v1;schemaBaseType=AbcGeom_GeomBase_v1");
        geom.add(OutProperty::scalarDoubles(".selfBnds", {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]}, "interpretation=box"));
        geom.add(OutProperty::arrayFloat("P", positions, 3, "geoScope=var;interpretation=point"));
        {
            OutProperty pid = OutProperty::arrayUInt64(".pointIds", ids);
            pid.metadata = "geoScope=var";
            geom.add(std::move(pid));
        }
        OutProperty arb = OutProperty::compound(".arbGeomParams");
        arb.add(OutProperty::arrayFloat("color", colors, 3, "arrayExtent=1;geoScope=vtx;interpretation=rgb;isGeomParam=true;podExtent=3;podName=float32_t"));
        geom.add(std::move(arb));
        OutProperty up = OutProperty::compound(".userProperties");
        up.add(OutProperty::arrayUInt32("mvg_describerType", descTypes));
        if(withObservations)
        {
            up.add(OutProperty::arrayUInt32("mvg_visibilitySize", visSize));
            up.add(OutProperty::arrayUInt32("mvg_visibilityViewId", visView));
            up.add(OutProperty::arrayUInt32("mvg_visibilityFeatId", visFeat));
            up.add(OutProperty::arrayFloat("mvg_visibilityFeatPos", featPos));
            up.add(OutProperty::arrayFloat("mvg_visibilityFeatScale", featScale));
        }
        geom.add(std::move(up));
        points.properties.add(std::move(geom));
        pointCloud.add(std::move(points));
    }
    cloud.add(std::move(pointCloud));
    root.add(std::move(cameras));
    root.add(std::move(undefined));
    root.add(std::move(cloud));
    root.add(xformObject("mvgAncestors", nullptr));
    top.add(std::move(root));
    abc::save(top, filename, "_ai_AlembicVersion=Alembic 1.8.4 layout (written by alicevision_amd)");
}

} // namespace avdm_host
