// png.hpp — the one PNG flavour the depth-map filtering step exchanges between its two passes: 8-bit greyscale, non-interlaced
// (the modal-count map `<viewId>_nmodMap.png`, fuseCut/Fuser.cpp:220-223, mvsUtils/fileIO.cpp:291-297).  zlib only.
#pragma once

#include <string>
#include <vector>

namespace avdm_host {

void writePngGray8(const std::string& path, int width, int height, const unsigned char* data);

// An input image of the estimation program (mvsUtils::loadImage reads any format OpenImageIO decodes, fileIO.cpp:386-446; here: PNG next to
// OpenEXR): 8- or 16-bit samples, greyscale / greyscale + alpha / RGB / RGBA, non-interlaced.  The INTEGER samples are returned as they are
// stored (16-bit in host byte order): the conversion to linear float RGBA runs on the device (avdm_image_decode_integer).
struct PngImage
{
    int width = 0, height = 0;
    int channels = 0; // 1, 2, 3, 4
    int bits = 0;     // 8 or 16
    std::vector<unsigned char> samples;
};
void readPng(const std::string& path, PngImage& out, bool headerOnly = false);
// for tests and tools: 8- or 16-bit, 1-4 channels, filter type 0 ... 4 chosen per row (exercises every unfilter of the reader)
void writePng(const std::string& path, int width, int height, int channels, int bits, const void* samples);
// 8-bit greyscale (colour type 0) or 8-bit RGB / RGBA / grey+alpha (first channel is returned), non-interlaced; throws otherwise
void readPngGray8(const std::string& path, int& width, int& height, std::vector<unsigned char>& data);

} // namespace avdm_host
