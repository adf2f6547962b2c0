// tiff.cpp — see tiff.hpp.  TIFF 6.0: image file directory, strips / tiles, the LZW (section 13), PackBits (section 9) and Deflate
// (Adobe supplement, zlib stream) schemes, differencing predictor (section 14).
#include "tiff.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace avdm_host {

namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("TIFF: " + what); }

struct File
{
    std::vector<uint8_t> b;
    bool le = true;
    uint16_t u16(size_t o) const
    {
        if(o + 2 > b.size())
            fail("truncated file");
        return le ? (uint16_t)(b[o] | (b[o + 1] << 8)) : (uint16_t)((b[o] << 8) | b[o + 1]);
    }
    uint32_t u32(size_t o) const
    {
        if(o + 4 > b.size())
            fail("truncated file");
        return le ? (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24)
                  : ((uint32_t)b[o] << 24) | ((uint32_t)b[o + 1] << 16) | ((uint32_t)b[o + 2] << 8) | (uint32_t)b[o + 3];
    }
};

// the values of one directory entry (BYTE, SHORT or LONG) as unsigned integers
std::vector<uint32_t> values(const File& f, size_t entry)
{
    const int type = f.u16(entry + 2);
    const uint32_t count = f.u32(entry + 4);
    const size_t size = type == 1 || type == 6 || type == 7 ? 1 : (type == 3 || type == 8 ? 2 : (type == 4 || type == 9 ? 4 : 0));
    if(size == 0)
        fail("directory entry of an unexpected type");
    if(count > (1u << 28))
        fail("directory entry with an absurd count");
    size_t at = entry + 8;
    if((size_t)count * size > 4)
        at = f.u32(entry + 8);
    std::vector<uint32_t> v(count);
    for(uint32_t i = 0; i < count; ++i)
        v[i] = size == 1 ? f.b.at(at + i) : (size == 2 ? f.u16(at + 2 * (size_t)i) : f.u32(at + 4 * (size_t)i));
    return v;
}

// TIFF LZW (TIFF 6.0 section 13): codes most-significant bit first, 9 .. 12 bits, 256 = clear, 257 = end of information; the code width
// grows when the table is ONE entry short of full for the current width ("early change", what every TIFF writer does)
void lzwDecode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected)
{
    std::vector<int> prefix(4096, -1);
    std::vector<uint8_t> last(4096, 0), first(4096, 0);
    std::vector<uint16_t> length(4096, 1);
    for(int i = 0; i < 256; ++i)
        last[i] = first[i] = (uint8_t)i;
    int next = 258, width = 9, old = -1;
    uint32_t acc = 0;
    int bits = 0;
    size_t p = 0;
    out.clear();
    out.reserve(expected);
    auto write = [&](int code) {
        const size_t len = length[code], at = out.size();
        out.resize(at + len);
        for(int c = code, k = (int)len - 1; c >= 0 && k >= 0; c = prefix[c], --k)
            out[at + (size_t)k] = last[c];
    };
    auto add = [&](int pre, uint8_t ch) {
        if(next >= 4096)
            return; // (a full table without a clear code: libtiff stops adding, too)
        prefix[next] = pre, last[next] = ch, first[next] = first[pre], length[next] = (uint16_t)(length[pre] + 1);
        ++next;
        if(next >= (1 << width) - 1 && width < 12)
            ++width;
    };
    while(out.size() < expected)
    {
        while(bits < width)
        {
            if(p >= n)
                return; // data ended without an end-of-information code: what was decoded stands (libtiff: a warning)
            acc = (acc << 8) | src[p++];
            bits += 8;
        }
        const int code = (int)((acc >> (bits - width)) & ((1u << width) - 1u));
        bits -= width;
        if(code == 257)
            break;
        if(code == 256)
        {
            next = 258, width = 9, old = -1;
            continue;
        }
        if(old < 0)
        {
            if(code >= 256)
                fail("corrupt LZW data (the first code after a clear is not a literal)");
            write(code);
        }
        else if(code < next)
        {
            write(code);
            add(old, first[code]);
        }
        else
        {
            if(code != next)
                fail("corrupt LZW data (code beyond the table)");
            add(old, first[old]); // the string of `old` followed by its own first byte: that is `code`
            write(code);
        }
        old = code;
    }
    if(out.size() > expected)
        out.resize(expected);
}

void packBitsDecode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected)
{
    out.clear();
    out.reserve(expected);
    size_t p = 0;
    while(p < n && out.size() < expected)
    {
        const int8_t c = (int8_t)src[p++];
        if(c >= 0)
        {
            const size_t len = (size_t)c + 1;
            if(p + len > n)
                fail("truncated PackBits data");
            out.insert(out.end(), src + p, src + p + len);
            p += len;
        }
        else if(c != -128)
        {
            if(p >= n)
                fail("truncated PackBits data");
            out.insert(out.end(), (size_t)(1 - c), src[p++]);
        }
    }
    out.resize(expected, 0);
}

void inflateAll(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected)
{
    out.assign(expected, 0);
    uLongf len = (uLongf)expected;
    const int rc = uncompress(out.data(), &len, src, (uLong)n);
    if(rc != Z_OK && rc != Z_BUF_ERROR)
        fail("corrupt Deflate data");
}

} // namespace

void readTiff(const std::string& filename, TiffImage& out, bool headerOnly)
{
    File f;
    {
        std::ifstream in(filename, std::ios::binary);
        if(!in)
            fail("cannot open '" + filename + "'");
        in.seekg(0, std::ios::end);
        const std::streamoff n = in.tellg();
        in.seekg(0);
        f.b.resize((size_t)std::max<std::streamoff>(n, 0));
        if(!f.b.empty())
            in.read(reinterpret_cast<char*>(f.b.data()), n);
    }
    if(f.b.size() < 8 || !((f.b[0] == 'I' && f.b[1] == 'I') || (f.b[0] == 'M' && f.b[1] == 'M')))
        fail("'" + filename + "' is not a TIFF file");
    f.le = f.b[0] == 'I';
    const int magic = f.u16(2);
    if(magic == 43)
        fail("BigTIFF is not supported");
    if(magic != 42)
        fail("'" + filename + "' is not a TIFF file");
    const size_t ifd = f.u32(4);
    const int nEntries = f.u16(ifd);

    uint32_t width = 0, height = 0, compression = 1, photometric = 1, spp = 1, rowsPerStrip = 0xffffffffu, planar = 1, predictor = 1;
    uint32_t tileW = 0, tileH = 0, orientation = 1;
    std::vector<uint32_t> bps = {1}, offsets, counts, tileOffsets, tileCounts, sampleFormat, extra;
    for(int i = 0; i < nEntries; ++i)
    {
        const size_t e = ifd + 2 + 12 * (size_t)i;
        const int tag = f.u16(e);
        switch(tag)
        {
            case 256: width = values(f, e).at(0); break;
            case 257: height = values(f, e).at(0); break;
            case 258: bps = values(f, e); break;
            case 259: compression = values(f, e).at(0); break;
            case 262: photometric = values(f, e).at(0); break;
            case 273: offsets = values(f, e); break;
            case 274: orientation = values(f, e).at(0); break;
            case 277: spp = values(f, e).at(0); break;
            case 278: rowsPerStrip = values(f, e).at(0); break;
            case 279: counts = values(f, e); break;
            case 284: planar = values(f, e).at(0); break;
            case 317: predictor = values(f, e).at(0); break;
            case 322: tileW = values(f, e).at(0); break;
            case 323: tileH = values(f, e).at(0); break;
            case 324: tileOffsets = values(f, e); break;
            case 325: tileCounts = values(f, e); break;
            case 338: extra = values(f, e); break;
            case 339: sampleFormat = values(f, e); break;
            default: break;
        }
    }
    if(width == 0 || height == 0 || width > 65535u * 4 || height > 65535u * 4)
        fail("empty or absurdly large image");
    if(spp < 1 || spp > 4)
        fail(std::to_string(spp) + " samples per pixel are not supported (1 to 4)");
    if(bps.size() != 1 && bps.size() != spp)
        fail("BitsPerSample does not match SamplesPerPixel");
    for(uint32_t b : bps)
        if(b != bps[0])
            fail("samples of different widths are not supported");
    if(bps[0] != 8 && bps[0] != 16)
        fail(std::to_string(bps[0]) + "-bit samples are not supported (8 or 16)");
    for(uint32_t sf : sampleFormat)
        if(sf != 1)
            fail("only unsigned integer samples are supported");
    if(!(photometric == 0 || photometric == 1 || photometric == 2))
        fail("photometric interpretation " + std::to_string(photometric) + " is not supported (grey or RGB)");
    if((photometric == 2 && spp < 3) || (photometric != 2 && spp > 2))
        fail("SamplesPerPixel does not fit the photometric interpretation");
    if(!(compression == 1 || compression == 5 || compression == 8 || compression == 32946 || compression == 32773))
        fail("compression scheme " + std::to_string(compression) + " is not supported (none, LZW, Deflate, PackBits)");
    if(predictor != 1 && predictor != 2)
        fail("predictor " + std::to_string(predictor) + " is not supported");
    if(planar != 1 && planar != 2)
        fail("bad PlanarConfiguration");

    out = TiffImage();
    out.width = (int)width, out.height = (int)height, out.channels = (int)spp, out.bits = (int)bps[0], out.orientation = (int)orientation;
    if(headerOnly)
        return;

    const size_t bytesPerSample = bps[0] / 8;
    const bool tiled = !tileOffsets.empty();
    if(tiled)
    {
        if(tileW == 0 || tileH == 0)
            fail("tiled image without tile dimensions");
        offsets = tileOffsets;
        counts = tileCounts;
    }
    else
    {
        tileW = width;
        tileH = std::min(rowsPerStrip, height);
        if(tileH == 0)
            fail("RowsPerStrip is zero");
    }
    if(offsets.empty() || offsets.size() != counts.size())
        fail("strip / tile offsets and byte counts do not match");
    // geometry from untrusted tags: size_t arithmetic, bounded before anything is allocated
    if(width == 0 || height == 0 || width > (1u << 20) || height > (1u << 20))
        fail("unreasonable image dimensions");
    if(tileW > 4u * width + 4096u || tileH > 4u * height + 4096u)
        fail("unreasonable tile / strip dimensions");
    if((size_t)width * height * spp * bytesPerSample > ((size_t)1 << 33))
        fail("image larger than 8 GiB of samples");
    const size_t across = ((size_t)width + tileW - 1) / tileW, down = ((size_t)height + tileH - 1) / tileH;
    const size_t planes = planar == 2 ? spp : 1, samplesPerChunkPixel = planar == 2 ? 1 : spp;
    if(offsets.size() < across * down * planes)
        fail("fewer strips / tiles than the image needs");

    out.samples.assign((size_t)width * height * spp * bytesPerSample, 0);
    std::vector<uint8_t> chunk;
    for(size_t plane = 0; plane < planes; ++plane)
        for(size_t ty = 0; ty < down; ++ty)
            for(size_t tx = 0; tx < across; ++tx)
            {
                const size_t idx = (plane * down + ty) * across + tx;
                const size_t off = offsets[idx], len = counts[idx];
                if(off > f.b.size() || len > f.b.size() - off)
                    fail("strip / tile outside the file");
                // rows a strip holds: the last one may be short; a tile is always whole
                const size_t rows = tiled ? tileH : std::min<size_t>(tileH, height - ty * tileH);
                const size_t rowBytes = (size_t)tileW * samplesPerChunkPixel * bytesPerSample, expected = rows * rowBytes;
                const uint8_t* src = f.b.data() + off;
                switch(compression)
                {
                    case 1:
                        if(len < expected)
                            fail("strip / tile shorter than its pixels");
                        chunk.assign(src, src + expected);
                        break;
                    case 5: lzwDecode(src, len, chunk, expected); chunk.resize(expected, 0); break;
                    case 32773: packBitsDecode(src, len, chunk, expected); break;
                    default: inflateAll(src, len, chunk, expected); break;
                }
                // byte order of 16-bit samples -> host (little endian), then the predictor on whole samples
                if(bytesPerSample == 2 && !f.le)
                    for(size_t i = 0; i + 1 < chunk.size(); i += 2)
                        std::swap(chunk[i], chunk[i + 1]);
                if(predictor == 2)
                    for(size_t r = 0; r < rows; ++r)
                    {
                        uint8_t* row = chunk.data() + r * rowBytes;
                        if(bytesPerSample == 1)
                            for(size_t i = samplesPerChunkPixel; i < (size_t)tileW * samplesPerChunkPixel; ++i)
                                row[i] = (uint8_t)(row[i] + row[i - samplesPerChunkPixel]);
                        else
                        {
                            uint16_t* r16 = reinterpret_cast<uint16_t*>(row);
                            for(size_t i = samplesPerChunkPixel; i < (size_t)tileW * samplesPerChunkPixel; ++i)
                                r16[i] = (uint16_t)(r16[i] + r16[i - samplesPerChunkPixel]);
                        }
                    }
                // place the chunk's pixels
                const size_t x0 = tx * tileW, y0 = ty * tileH;
                const size_t copyW = std::min<size_t>(tileW, width - x0);
                for(size_t r = 0; r < rows && y0 + r < height; ++r)
                {
                    const uint8_t* srow = chunk.data() + r * rowBytes;
                    uint8_t* drow = out.samples.data() + ((y0 + r) * width + x0) * spp * bytesPerSample;
                    if(planar == 1)
                        std::memcpy(drow, srow, copyW * spp * bytesPerSample);
                    else
                        for(size_t x = 0; x < copyW; ++x)
                            std::memcpy(drow + (x * spp + plane) * bytesPerSample, srow + x * bytesPerSample, bytesPerSample);
                }
            }
    if(photometric == 0)
    { // WhiteIsZero: the colour samples are inverted (not the alpha)
        const size_t n = (size_t)width * height;
        for(size_t i = 0; i < n; ++i)
        {
            if(bytesPerSample == 1)
                out.samples[i * spp] = (uint8_t)(255 - out.samples[i * spp]);
            else
            {
                uint16_t* s = reinterpret_cast<uint16_t*>(out.samples.data()) + i * spp;
                *s = (uint16_t)(65535 - *s);
            }
        }
    }
}

} // namespace avdm_host
