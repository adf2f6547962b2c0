// alembic.hpp — a reader and a writer for the subset of Alembic (Ogawa container) that AliceVision's SfMData files use, written from
// the published file layout (no Alembic library in this build): Meshroom's StructureFromMotion node hands `sfm.abc` to the depth-map
// stage, so `-i sfm.abc` has to work for this program to take the reference node's place.
//
// Container ("Ogawa"): 16-byte header { "Ogawa", frozen flag 0xff, u16 version, u64 position of the root group }; a GROUP is
// { u64 count, count x u64 child }, a child with bit 63 set is DATA { u64 size, bytes } at (child & ~bit63), otherwise a group; position 0
// is the empty group / empty data.  Alembic on top of it (AbcCoreOgawa):
//   root group  = [ data: archive version, data: library version (i32, e.g. 10804), group: top object, data: archive metadata,
//                   data: time samplings, data: indexed metadata (u8 size + text, repeated) ]
//   object group = [ group: its compound property, group per child object ..., data: child headers { u32 size, name, u8 metadata index
//                   (0xff: u32 size + text inline) } ... followed by 32 bytes of hashes ]
//   compound property group = [ one child per property ..., data: property headers ]; a header is a u32 bit field (bits 0-1 kind:
//                   0 compound / 1 scalar / 2 array, 2-3 width of the integers that follow (u8 / u16 / u32), 4-7 POD, 8 has a time-sampling
//                   index, 9 has first / last changed index, 10 homogeneous, 11 all samples equal, 12-19 extent, 20-27 metadata index),
//                   then { next sample index, [first changed, last changed], [time sampling index] } (non-compound), the name and,
//                   for metadata index 0xff, the metadata text
//   scalar property group = [ data per stored sample ]; array property group = [ data, dimensions ] per stored sample; every sample
//                   blob starts with a 16-byte digest; strings are NUL-terminated and concatenated.
// Checked against the nine scene_v1.2.*.abc files of the reference's own compatibility tests (sfmDataIO/compatibilityData, written
// by Alembic 1.7.16 and 1.8.4), whose .json twins the JSON reader loads: tests/test_host_cpu.py::test_alembic_*.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace avdm_host {
namespace abc {

enum class Pod : int
{
    Bool = 0,
    UInt8,
    Int8,
    UInt16,
    Int16,
    UInt32,
    Int32,
    UInt64,
    Int64,
    Float16,
    Float32,
    Float64,
    String,
    WString,
    Unknown = 127
};
size_t podBytes(Pod p);

struct Node
{
    bool isData = false;
    uint64_t pos = 0;
};

struct PropertyHeader
{
    enum Kind
    {
        Compound = 0,
        Scalar = 1,
        Array = 2
    };
    Kind kind = Compound;
    Pod pod = Pod::Unknown;
    int extent = 0;
    uint32_t nextSampleIndex = 0, firstChangedIndex = 0, lastChangedIndex = 0, timeSamplingIndex = 0;
    std::string name, metadata;
    Node node;
    bool isArray() const { return kind == Array; }
    size_t numSamples() const { return nextSampleIndex; }
};

struct Object
{
    std::string name, metadata;
    uint64_t pos = 0;
    // value of `key` in the "k=v;k=v" metadata text, "" when absent
    std::string meta(const std::string& key) const;
};

std::string metadataValue(const std::string& metadata, const std::string& key);

class Archive
{
public:
    explicit Archive(const std::string& filename);

    int libraryVersion() const { return _libraryVersion; }
    Object top() const;
    std::vector<Object> children(const Object& o) const;
    bool child(const Object& o, const std::string& name, Object& out) const;
    // the properties of an object's top-level compound / of a compound property
    std::vector<PropertyHeader> properties(const Object& o) const;
    std::vector<PropertyHeader> properties(const PropertyHeader& compound) const;
    static const PropertyHeader* find(const std::vector<PropertyHeader>& props, const std::string& name);

    // one sample as doubles / unsigned / strings; numeric PODs convert (the reference reads uint32 properties that old files hold as
    // int32, AlembicImporter.cpp:34-44,88-98), a string property read as numbers (or the reverse) throws
    std::vector<double> readDoubles(const PropertyHeader& p, size_t sample = 0) const;
    std::vector<uint64_t> readUInts(const PropertyHeader& p, size_t sample = 0) const;
    std::vector<std::string> readStrings(const PropertyHeader& p, size_t sample = 0) const;
    std::vector<float> readFloats(const PropertyHeader& p, size_t sample = 0) const;

private:
    std::vector<uint8_t> _bytes;
    std::vector<std::string> _indexedMetadata;
    uint64_t _topPos = 0;
    int _libraryVersion = 0;

    uint64_t u64At(uint64_t pos) const;
    std::vector<Node> group(uint64_t pos) const;
    std::pair<const uint8_t*, size_t> data(uint64_t pos) const;
    std::vector<PropertyHeader> propertiesAt(uint64_t compoundPos) const;
    std::pair<const uint8_t*, size_t> sampleBytes(const PropertyHeader& p, size_t sample) const;
};

// ---- writer -------------------------------------------------------------------------------------------------------------------------
// Builds the same layout: objects and properties are described in memory, save() lays them out bottom-up (a child is written before
// the group that points at it, like the Ogawa streams of the library); every sample carries its MurmurHash3 digest like the library's
// (its read cache is keyed on it) and equal samples are stored once.
class OutProperty
{
public:
    PropertyHeader::Kind kind = PropertyHeader::Compound;
    Pod pod = Pod::Unknown;
    int extent = 1;
    std::string name, metadata;
    std::vector<uint8_t> sample;                // scalar / array: the one (static) sample
    size_t count = 0;                           // array: number of elements (extent-sized)
    std::vector<OutProperty> children;          // compound

    static OutProperty compound(const std::string& name, const std::string& metadata = "");
    OutProperty& add(OutProperty p)
    {
        children.push_back(std::move(p));
        return children.back();
    }
    static OutProperty scalarBool(const std::string& name, bool v);
    static OutProperty scalarUInt32(const std::string& name, uint32_t v);
    static OutProperty scalarUInt16(const std::string& name, uint16_t v);
    static OutProperty scalarDouble(const std::string& name, double v);
    static OutProperty scalarString(const std::string& name, const std::string& v);
    static OutProperty scalarDoubles(const std::string& name, const std::vector<double>& v, const std::string& metadata = "");  // extent = size
    static OutProperty scalarBytes(const std::string& name, const std::vector<uint8_t>& v);                                      // u8, extent = size
    static OutProperty arrayUInt32(const std::string& name, const std::vector<uint32_t>& v);
    static OutProperty arrayUInt64(const std::string& name, const std::vector<uint64_t>& v);
    static OutProperty arrayDouble(const std::string& name, const std::vector<double>& v);
    static OutProperty arrayFloat(const std::string& name, const std::vector<float>& v, int extent = 1, const std::string& metadata = "");
    static OutProperty arrayString(const std::string& name, const std::vector<std::string>& v);
};

class OutObject
{
public:
    std::string name, metadata;
    OutProperty properties = OutProperty::compound("");
    std::vector<OutObject> children;
    OutObject() = default;
    OutObject(std::string n, std::string m) : name(std::move(n)), metadata(std::move(m)) {}
    OutObject& add(OutObject o)
    {
        children.push_back(std::move(o));
        return children.back();
    }
};

void save(const OutObject& top, const std::string& filename, const std::string& archiveMetadata);

} // namespace abc

struct SfMData;
// sfmDataIO::AlembicImporter::populateSfM (sfmDataIO/AlembicImporter.cpp:1046-1108) for the parts SfMData (sfmData.hpp) holds
void loadSfMDataAlembic(SfMData& out, const std::string& filename);
// sfmDataIO::AlembicExporter (sfmDataIO/AlembicExporter.cpp): views with their intrinsics and poses, landmarks with their observations.
// withViews = false, withObservations = false is sfmDataIO::save(.., ESfMData::STRUCTURE): the point cloud alone (the debug volume exports)
void saveSfMDataAlembic(const SfMData& in, const std::string& filename, bool withViews = true, bool withObservations = true);

} // namespace avdm_host
