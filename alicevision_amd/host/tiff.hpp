// tiff.hpp — a reader for the TIFF files photogrammetry datasets come in (what RAW converters and scanners write): classic TIFF 6.0,
// first image, 8- / 16-bit unsigned samples, grey or RGB with optional alpha, strips or tiles, chunky or planar, compression none /
// LZW / Deflate / PackBits, horizontal predictor, either byte order.  Like host/png.cpp it stops at the INTEGER samples: they go to the
// device as stored and become linear float RGBA there (avdm_image_decode_integer with the sRGB decoding: OpenImageIO reports 8- / 16-bit
// TIFF as sRGB, and image::readImage(path, img, LINEAR) of the reference converts from that, image/io.cpp:571-760).
// Not read: BigTIFF, JPEG-in-TIFF, palette and CMYK / YCbCr / Lab photometrics, 1- / 4- / 32-bit and floating-point samples.
#pragma once

#include <string>
#include <vector>

namespace avdm_host {

struct TiffImage
{
    int width = 0, height = 0, channels = 0, bits = 0; // channels: 1 Y, 2 YA, 3 RGB, 4 RGBA; bits: 8 or 16
    int orientation = 1;                               // tag 274 (reported; the reference does not rotate, image/io.cpp:512-514)
    std::vector<unsigned char> samples;                // interleaved, rows top to bottom, 16-bit samples in host byte order
};

// throws std::runtime_error with the reason; headerOnly: size and layout only
void readTiff(const std::string& filename, TiffImage& out, bool headerOnly = false);

} // namespace avdm_host
