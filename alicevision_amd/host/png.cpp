// png.cpp — minimal PNG codec (ISO/IEC 15948): signature, IHDR / IDAT / IEND chunks with CRC-32, zlib stream, scan-line filters.
#include "png.hpp"

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>

namespace avdm_host {

namespace {
const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

void putU32(std::vector<unsigned char>& v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24));
    v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8));
    v.push_back((unsigned char)x);
}
uint32_t getU32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

void writeChunk(std::ofstream& f, const char type[4], const std::vector<unsigned char>& payload)
{
    std::vector<unsigned char> head;
    putU32(head, (uint32_t)payload.size());
    f.write((const char*)head.data(), 4);
    f.write(type, 4);
    if(!payload.empty())
        f.write((const char*)payload.data(), (std::streamsize)payload.size());
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if(!payload.empty())
        crc = crc32(crc, payload.data(), (uInt)payload.size());
    std::vector<unsigned char> tail;
    putU32(tail, (uint32_t)crc);
    f.write((const char*)tail.data(), 4);
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if(pa <= pb && pa <= pc)
        return a;
    return pb <= pc ? b : c;
}
} // namespace

void writePngGray8(const std::string& path, int width, int height, const unsigned char* data)
{
    if(width <= 0 || height <= 0)
        throw std::runtime_error("writePngGray8: empty image");
    // filter type 0 on every row, fastest deflate level: the maps are small-valued and flat (12 MP -> a few hundred KB either way)
    std::vector<unsigned char> raw((size_t)height * (width + 1));
    for(int y = 0; y < height; ++y)
    {
        raw[(size_t)y * (width + 1)] = 0;
        std::memcpy(&raw[(size_t)y * (width + 1) + 1], data + (size_t)y * width, (size_t)width);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if(compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) != Z_OK)
        throw std::runtime_error("writePngGray8: deflate failed");
    z.resize(zlen);

    std::ofstream f(path, std::ios::binary);
    if(!f)
        throw std::runtime_error("Cannot open '" + path + "' for writing.");
    f.write((const char*)kSignature, 8);
    std::vector<unsigned char> ihdr;
    putU32(ihdr, (uint32_t)width);
    putU32(ihdr, (uint32_t)height);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(0); // colour type: greyscale
    ihdr.push_back(0); // compression
    ihdr.push_back(0); // filter method
    ihdr.push_back(0); // no interlace
    writeChunk(f, "IHDR", ihdr);
    writeChunk(f, "IDAT", z);
    writeChunk(f, "IEND", {});
    if(!f)
        throw std::runtime_error("Error while writing '" + path + "'.");
}

void readPng(const std::string& path, PngImage& out, bool headerOnly)
{
    std::ifstream f(path, std::ios::binary);
    if(!f)
        throw std::runtime_error("Cannot open '" + path + "'.");
    std::vector<unsigned char> file;
    if(headerOnly)
    {
        file.resize(8 + 25);
        f.read((char*)file.data(), (std::streamsize)file.size());
        if(f.gcount() != (std::streamsize)file.size())
            throw std::runtime_error("'" + path + "' is not a PNG file.");
    }
    else
        file.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if(file.size() < 8 + 25 || std::memcmp(file.data(), kSignature, 8) != 0)
        throw std::runtime_error("'" + path + "' is not a PNG file.");
    size_t pos = 8;
    int bitDepth = 0, colourType = 0, interlace = 0;
    bool haveHeader = false;
    std::vector<unsigned char> z;
    while(pos + 12 <= file.size())
    {
        const uint32_t len = getU32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if(pos + 12 + (size_t)len > file.size())
            throw std::runtime_error("'" + path + "': truncated PNG chunk.");
        const unsigned char* payload = &file[pos + 8];
        const uLong crc = crc32(crc32(0L, (const Bytef*)type, 4), payload, len);
        if((uint32_t)crc != getU32(payload + len))
            throw std::runtime_error("'" + path + "': PNG chunk CRC mismatch.");
        if(std::memcmp(type, "IHDR", 4) == 0 && len == 13)
        {
            out.width = (int)getU32(payload);
            out.height = (int)getU32(payload + 4);
            bitDepth = payload[8], colourType = payload[9], interlace = payload[12];
            haveHeader = true;
            if(headerOnly)
                break;
        }
        else if(std::memcmp(type, "IDAT", 4) == 0)
            z.insert(z.end(), payload, payload + len);
        else if(std::memcmp(type, "IEND", 4) == 0)
            break;
        pos += 12 + (size_t)len;
    }
    if(!haveHeader || out.width <= 0 || out.height <= 0)
        throw std::runtime_error("'" + path + "': no PNG header.");
    switch(colourType)
    {
        case 0: out.channels = 1; break;
        case 2: out.channels = 3; break;
        case 4: out.channels = 2; break;
        case 6: out.channels = 4; break;
        default: throw std::runtime_error("'" + path + "': palette PNG files are not supported.");
    }
    if((bitDepth != 8 && bitDepth != 16) || interlace != 0)
        throw std::runtime_error("'" + path + "': only 8- and 16-bit non-interlaced PNG files are supported.");
    out.bits = bitDepth;
    if(headerOnly)
        return;
    const int bpp = out.channels * (bitDepth / 8); // bytes per pixel = the distance of the filters' "previous" byte
    const size_t stride = (size_t)out.width * bpp;
    std::vector<unsigned char> raw((size_t)out.height * (stride + 1));
    uLongf rawLen = (uLongf)raw.size();
    if(uncompress(raw.data(), &rawLen, z.data(), (uLong)z.size()) != Z_OK || rawLen != raw.size())
        throw std::runtime_error("'" + path + "': inflate failed.");
    // undo the scan-line filters (byte-wise, ISO/IEC 15948 section 9), row by row into the output
    out.samples.assign((size_t)out.height * stride, 0);
    const std::vector<unsigned char> zeroRow(stride, 0);
    for(int y = 0; y < out.height; ++y)
    {
        const unsigned char ft = raw[(size_t)y * (stride + 1)];
        const unsigned char* in = &raw[(size_t)y * (stride + 1) + 1];
        unsigned char* cur = &out.samples[(size_t)y * stride];
        const unsigned char* prev = y > 0 ? cur - stride : zeroRow.data();
        if(ft > 4)
            throw std::runtime_error("'" + path + "': unknown PNG filter type.");
        for(size_t i = 0; i < stride; ++i)
        {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = in[i];
            switch(ft)
            {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            cur[i] = (unsigned char)v;
        }
    }
    if(bitDepth == 16)
    { // network byte order -> host order
        uint16_t* w = reinterpret_cast<uint16_t*>(out.samples.data());
        const size_t n = out.samples.size() / 2;
        for(size_t i = 0; i < n; ++i)
        {
            const unsigned char* b = &out.samples[2 * i];
            w[i] = (uint16_t)(((unsigned)b[0] << 8) | (unsigned)b[1]);
        }
    }
}

void writePng(const std::string& path, int width, int height, int channels, int bits, const void* samples)
{
    if(width <= 0 || height <= 0 || channels < 1 || channels > 4 || (bits != 8 && bits != 16))
        throw std::runtime_error("writePng: unsupported image");
    const int bpp = channels * (bits / 8);
    const size_t stride = (size_t)width * bpp;
    // rows in network byte order
    std::vector<unsigned char> rows((size_t)height * stride);
    if(bits == 8)
        std::memcpy(rows.data(), samples, rows.size());
    else
    {
        const uint16_t* w = static_cast<const uint16_t*>(samples);
        for(size_t i = 0; i < rows.size() / 2; ++i)
        {
            rows[2 * i] = (unsigned char)(w[i] >> 8);
            rows[2 * i + 1] = (unsigned char)(w[i] & 0xff);
        }
    }
    std::vector<unsigned char> raw((size_t)height * (stride + 1));
    const std::vector<unsigned char> zeroRow(stride, 0);
    for(int y = 0; y < height; ++y)
    {
        const int ft = y % 5; // every filter type in turn
        unsigned char* o = &raw[(size_t)y * (stride + 1)];
        o[0] = (unsigned char)ft;
        const unsigned char* cur = &rows[(size_t)y * stride];
        const unsigned char* prev = y > 0 ? cur - stride : zeroRow.data();
        for(size_t i = 0; i < stride; ++i)
        {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch(ft)
            {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: pred = paeth(a, b, c); break;
                default: break;
            }
            o[1 + i] = (unsigned char)(cur[i] - pred);
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if(compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_BEST_SPEED) != Z_OK)
        throw std::runtime_error("writePng: deflate failed");
    z.resize(zlen);
    std::ofstream f(path, std::ios::binary);
    if(!f)
        throw std::runtime_error("Cannot open '" + path + "' for writing.");
    f.write((const char*)kSignature, 8);
    std::vector<unsigned char> ihdr;
    putU32(ihdr, (uint32_t)width);
    putU32(ihdr, (uint32_t)height);
    ihdr.push_back((unsigned char)bits);
    ihdr.push_back((unsigned char)(channels == 1 ? 0 : channels == 2 ? 4 : channels == 3 ? 2 : 6));
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    writeChunk(f, "IHDR", ihdr);
    writeChunk(f, "IDAT", z);
    writeChunk(f, "IEND", {});
    if(!f)
        throw std::runtime_error("Error while writing '" + path + "'.");
}

void readPngGray8(const std::string& path, int& width, int& height, std::vector<unsigned char>& data)
{
    PngImage img;
    readPng(path, img);
    if(img.bits != 8)
        throw std::runtime_error("'" + path + "': only 8-bit PNG files are supported here.");
    width = img.width;
    height = img.height;
    data.assign((size_t)width * height, 0);
    for(size_t i = 0; i < data.size(); ++i)
        data[i] = img.samples[i * img.channels];
}

} // namespace avdm_host
