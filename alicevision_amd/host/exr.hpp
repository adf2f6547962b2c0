// exr.hpp — minimal OpenEXR scan-line reader / writer (the reference goes through OpenImageIO: image/io.cpp readImage /
// writeImage / readImageMetadata, called from mvsUtils/mapIO.cpp:402-540 and mvsUtils/fileIO.cpp:389-443).
// File layout per the published OpenEXR file-format specification ("OpenEXR File Layout", openexr.com): magic 20000630,
// version 2, attribute list, line-offset table, chunks.  Supported: single-part scan-line files, pixel types HALF / FLOAT /
// UINT, compression NONE / ZIPS / ZIP, x/y sampling 1.  Written: ZIP (16-line blocks), increasing-Y line order.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace avdm_host {

struct ExrAttribute
{
    std::string name, type;
    std::vector<uint8_t> data;
};

struct ExrAttributes
{
    std::vector<ExrAttribute> list;
    void set(const std::string& name, const std::string& type, const void* data, size_t bytes);
    void setInt(const std::string& name, int v) { set(name, "int", &v, 4); }
    void setFloat(const std::string& name, float v) { set(name, "float", &v, 4); }
    void setString(const std::string& name, const std::string& v) { set(name, "string", v.data(), v.size()); }
    void setM44d(const std::string& name, const double v[16]) { set(name, "m44d", v, 128); }
    void setM33d(const std::string& name, const double v[9]) { set(name, "m33d", v, 72); }
    void setV3d(const std::string& name, const double v[3]) { set(name, "v3d", v, 24); }
    const ExrAttribute* find(const std::string& name) const;
    bool getInt(const std::string& name, int& out) const;
    bool getFloat(const std::string& name, float& out) const;
    bool getM44d(const std::string& name, double out[16]) const;
};

struct ExrImage
{
    int width = 0, height = 0;      // data window size
    int dataX0 = 0, dataY0 = 0;     // data window origin
    int displayW = 0, displayH = 0; // display window size (origin 0,0)
    std::vector<std::string> channelNames;       // alphabetical, as stored
    std::vector<std::vector<float>> channels;    // one plane per channel, row-major width x height
    ExrAttributes attributes;                    // everything except the structural attributes
    int channelIndex(const std::string& name) const;
};

// throws std::runtime_error; headerOnly skips the pixel data (readImageMetadata / readImageSize)
void readExr(const std::string& path, ExrImage& out, bool headerOnly = false);

// The scan lines of a file as OpenEXR stores them — per line the channels one after the other (alphabetical), each `width` samples of its
// pixel type — WITHOUT being de-interleaved on the host: the bytes go to the device as they are and become linear float RGBA there
// (avdm_image_decode_exr_lines).  An uncompressed file is mapped, not read: `lines` then points into the mapping and consecutive lines are
// `lineStride` = bytes per line + the 8-byte chunk header apart; ZIP / ZIPS blocks are inflated (host cores) into one buffer, lineStride =
// bytes per line.  What image::readImage does for an .exr through OpenImageIO (image/io.cpp, called by mvsUtils/fileIO.cpp:389-443), split
// between host (container) and device (samples).
struct ExrLines
{
    int width = 0, height = 0;
    const uint8_t* lines = nullptr; // first sample of line 0
    size_t bytes = 0;               // from `lines` to the end of the last line
    long long lineStride = 0;
    long long chanOffset[4] = {-1, -1, -1, -1}; // byte offset inside a line of R, G, B, A (-1: absent; a Y-only file gives Y as R, G and B)
    int chanType[4] = {2, 2, 2, 2};             // OpenEXR pixel type: 0 UINT, 1 HALF, 2 FLOAT
    ExrLines() = default;
    ExrLines(const ExrLines&) = delete;
    ExrLines& operator=(const ExrLines&) = delete;
    ~ExrLines();
    // (owners of `lines`)
    void* mapBase = nullptr;
    size_t mapBytes = 0;
    std::unique_ptr<uint8_t[]> inflated;
};
// false: a layout this form does not take (line chunks not in increasing-y order at a uniform stride, no R,G,B or Y channel, sub-sampling):
// the caller falls back to readExr.  Throws like readExr on a broken file.
bool readExrLines(const std::string& path, ExrLines& out);

struct ExrChannelIn
{
    std::string name;
    const float* data; // width x height, row-major, dense
};
// storeHalf: all channels as HALF (EStorageDataType::Half) instead of FLOAT.  dataX0/dataY0 + displayW/H describe a tile of
// a larger image (oiio pixelRoi / displayRoi, mapIO.cpp:428-432); pass 0,0,width,height for a whole image.
void writeExr(const std::string& path, int width, int height, const std::vector<ExrChannelIn>& channels, bool storeHalf, const ExrAttributes& attributes,
              int dataX0, int dataY0, int displayW, int displayH);

uint16_t floatToHalf(float f);
float halfToFloat(uint16_t h);

} // namespace avdm_host
