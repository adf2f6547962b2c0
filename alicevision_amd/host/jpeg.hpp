// jpeg.hpp — the entropy-decoding half of a JPEG reader: markers, Huffman tables and scans (baseline / extended sequential and
// progressive DCT, 8-bit, Huffman coding) down to the QUANTISED DCT coefficients of every block.  The rest of the decode — dequantisation,
// the 8 x 8 inverse DCT, chroma up-sampling, YCbCr -> RGB and sRGB -> linear — is per-block / per-pixel integer work and runs on the device
// (avdm_image_decode_jpeg, csrc/avdm_jpeg.hip), restated on the CPU as the parity checker (oracle/: avo_image_decode_jpeg).
//
// The reference reads photographs through OpenImageIO (image/io.cpp: readImage), whose JPEG plugin is libjpeg(-turbo) with its defaults:
// the accurate integer inverse DCT (JDCT_ISLOW, jidctint.c) and "fancy" (triangle-filter) chroma up-sampling (jdsample.c).  Neither library
// is part of /root/reference; the algorithms are restated from the JPEG standard (ITU T.81, Annex F / G for the entropy coding) and the
// published libjpeg sources' arithmetic, and pinned by golden vectors decoded with Pillow's bundled libjpeg-turbo
// (tests/golden/jpeg/, tests/golden/make_jpeg_fixtures.py).
// Not read: arithmetic coding, 12-bit samples, lossless / hierarchical JPEG, CMYK / YCCK (four components).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace avdm_host {

struct JpegComponent
{
    int id = 0, h = 1, v = 1, tq = 0;       // identifier, sampling factors, quantisation table
    int blocksW = 0, blocksH = 0;           // blocks stored per row / column (padded to whole MCUs)
    int width = 0, height = 0;              // "downsampled" size in samples: ceil(image * h / hmax), ceil(image * v / vmax)
    std::vector<int16_t> coef;              // blocksH x blocksW x 64, natural (row-major) order inside a block
};

struct JpegImage
{
    int width = 0, height = 0;
    int hmax = 1, vmax = 1;
    bool progressive = false;
    bool jfif = false, adobe = false;
    int adobeTransform = -1;                // APP14 transform flag: 0 = the components are RGB (or CMYK), 1 = YCbCr, 2 = YCCK
    int exifOrientation = 0;                // 0 = not stated
    std::vector<JpegComponent> components;  // 1 (grey) or 3
    uint16_t quant[4][64] = {};             // natural order
    // true when the three components are stored as RGB (Adobe transform 0, or component identifiers 'R' 'G' 'B'), libjpeg's rule
    bool storedAsRgb() const;
};

// throws std::runtime_error with the reason; headerOnly: stop at the first scan (size, sampling, tables seen so far)
void readJpeg(const std::string& filename, JpegImage& out, bool headerOnly = false);
void readJpegMemory(const uint8_t* data, size_t size, JpegImage& out, bool headerOnly = false);

} // namespace avdm_host
