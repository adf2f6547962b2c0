// exr.cpp — see exr.hpp.  Compression follows the OpenEXR ZIP scheme: byte de-interleave (even bytes, then odd bytes),
// delta predictor (d = cur - prev + 128), zlib deflate; a block that does not shrink is stored raw.
#include "exr.hpp"

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace avdm_host {

uint16_t floatToHalf(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if(((x >> 23) & 0xff) == 0xff) // inf / nan
        return (uint16_t)(sign | 0x7c00u | (man ? 0x200u | (man >> 13) : 0u));
    if(exp >= 31)
        return (uint16_t)(sign | 0x7c00u); // overflow -> inf
    if(exp <= 0)
    {
        if(exp < -10)
            return (uint16_t)sign; // underflow -> 0
        man |= 0x800000u;
        const int shift = 14 - exp; // 14..24
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if(rem > half || (rem == half && (h & 1u)))
            ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if(rem > 0x1000u || (rem == 0x1000u && (h & 1u)))
        ++h; // may carry into the exponent, which is the correct rounding (up to inf)
    return (uint16_t)(sign | h);
}

float halfToFloat(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, x;
    if(exp == 0)
    {
        if(man == 0)
            x = sign;
        else
        {
            int e = -1;
            do
            {
                ++e;
                man <<= 1;
            } while(!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    }
    else if(exp == 31)
        x = sign | 0x7f800000u | (man << 13);
    else
        x = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

void ExrAttributes::set(const std::string& name, const std::string& type, const void* data, size_t bytes)
{
    for(auto& a : list)
        if(a.name == name)
        {
            a.type = type;
            a.data.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
            return;
        }
    ExrAttribute a;
    a.name = name;
    a.type = type;
    a.data.assign((const uint8_t*)data, (const uint8_t*)data + bytes);
    list.push_back(std::move(a));
}
const ExrAttribute* ExrAttributes::find(const std::string& name) const
{
    for(const auto& a : list)
        if(a.name == name)
            return &a;
    return nullptr;
}
bool ExrAttributes::getInt(const std::string& name, int& out) const
{
    const ExrAttribute* a = find(name);
    if(!a || a->type != "int" || a->data.size() != 4)
        return false;
    std::memcpy(&out, a->data.data(), 4);
    return true;
}
bool ExrAttributes::getFloat(const std::string& name, float& out) const
{
    const ExrAttribute* a = find(name);
    if(!a || a->type != "float" || a->data.size() != 4)
        return false;
    std::memcpy(&out, a->data.data(), 4);
    return true;
}
bool ExrAttributes::getM44d(const std::string& name, double out[16]) const
{
    const ExrAttribute* a = find(name);
    if(!a || a->type != "m44d" || a->data.size() != 128)
        return false;
    std::memcpy(out, a->data.data(), 128);
    return true;
}
int ExrImage::channelIndex(const std::string& name) const
{
    for(size_t i = 0; i < channelNames.size(); ++i)
        if(channelNames[i] == name)
            return (int)i;
    return -1;
}

namespace {

enum { PT_UINT = 0, PT_HALF = 1, PT_FLOAT = 2 };
enum { C_NONE = 0, C_RLE = 1, C_ZIPS = 2, C_ZIP = 3 };

struct Chan
{
    std::string name;
    int type = PT_FLOAT;
};

struct Cursor
{
    const uint8_t* p;
    const uint8_t* e;
    void need(size_t n) const
    {
        if((size_t)(e - p) < n)
            throw std::runtime_error("EXR: truncated header");
    }
    std::string cstr()
    {
        const uint8_t* b = p;
        while(p < e && *p)
            ++p;
        if(p >= e)
            throw std::runtime_error("EXR: unterminated string");
        std::string s((const char*)b, (size_t)(p - b));
        ++p;
        return s;
    }
    int32_t i32()
    {
        need(4);
        int32_t v;
        std::memcpy(&v, p, 4);
        p += 4;
        return v;
    }
};

void zipUndo(std::vector<uint8_t>& buf, std::vector<uint8_t>& tmp)
{
    const size_t n = buf.size();
    if(n == 0)
        return;
    // predictor
    for(size_t i = 1; i < n; ++i)
        buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
    // interleave the two halves
    tmp.resize(n);
    const uint8_t* t1 = buf.data();
    const uint8_t* t2 = buf.data() + (n + 1) / 2;
    size_t o = 0;
    while(true)
    {
        if(o < n)
            tmp[o++] = *t1++;
        else
            break;
        if(o < n)
            tmp[o++] = *t2++;
        else
            break;
    }
    buf.swap(tmp);
}

void zipDo(const std::vector<uint8_t>& raw, std::vector<uint8_t>& out)
{
    const size_t n = raw.size();
    std::vector<uint8_t> tmp(n);
    uint8_t* t1 = tmp.data();
    uint8_t* t2 = tmp.data() + (n + 1) / 2;
    for(size_t i = 0; i < n; ++i)
    {
        if(i & 1)
            *t2++ = raw[i];
        else
            *t1++ = raw[i];
    }
    int p = n ? tmp[0] : 0;
    for(size_t i = 1; i < n; ++i)
    {
        const int d = (int)tmp[i] - p + (128 + 256);
        p = tmp[i];
        tmp[i] = (uint8_t)d;
    }
    uLongf cap = compressBound((uLong)n);
    out.resize(cap);
    // deflate level 4 = OpenEXR's own default since 3.1 (ImfCompression "zipCompressionLevel"); zlib's default (6) is ~1.6x slower here
    if(compress2(out.data(), &cap, tmp.data(), (uLong)n, 4) != Z_OK)
        throw std::runtime_error("EXR: zlib compress failed");
    if(cap >= n)
        out = raw; // stored raw when it does not shrink
    else
        out.resize(cap);
}

void putAttr(std::vector<uint8_t>& h, const std::string& name, const std::string& type, const void* data, size_t bytes)
{
    h.insert(h.end(), name.begin(), name.end());
    h.push_back(0);
    h.insert(h.end(), type.begin(), type.end());
    h.push_back(0);
    const int32_t sz = (int32_t)bytes;
    h.insert(h.end(), (const uint8_t*)&sz, (const uint8_t*)&sz + 4);
    h.insert(h.end(), (const uint8_t*)data, (const uint8_t*)data + bytes);
}

} // namespace

void readExr(const std::string& path, ExrImage& out, bool headerOnly)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if(!f)
        throw std::runtime_error("cannot open image '" + path + "'");
    const std::streamsize fileSize = f.tellg();
    f.seekg(0);
    // the header is small; read up to 1 MiB for it, the chunks are read by offset afterwards
    std::vector<uint8_t> head((size_t)std::min<std::streamsize>(fileSize, headerOnly ? (1 << 20) : fileSize));
    f.read((char*)head.data(), (std::streamsize)head.size());
    Cursor c{head.data(), head.data() + head.size()};
    if(c.i32() != 20000630)
        throw std::runtime_error("'" + path + "' is not an OpenEXR file");
    const int32_t version = c.i32();
    if((version & 0xff) != 2 || (version & 0x200) || (version & 0x1000) || (version & 0x800))
        throw std::runtime_error("EXR '" + path + "': only single-part scan-line files are supported");
    const bool longNames = (version & 0x400) != 0;
    (void)longNames;
    std::vector<Chan> chans;
    int compression = -1, lineOrder = 0;
    int dw[4] = {0, 0, -1, -1}, disp[4] = {0, 0, -1, -1};
    out = ExrImage();
    for(;;)
    {
        c.need(1);
        if(*c.p == 0)
        {
            ++c.p;
            break;
        }
        const std::string name = c.cstr(), type = c.cstr();
        const int32_t size = c.i32();
        if(size < 0)
            throw std::runtime_error("EXR: bad attribute size");
        c.need((size_t)size);
        const uint8_t* d = c.p;
        c.p += size;
        if(name == "channels")
        {
            Cursor cc{d, d + size};
            while(cc.p < cc.e && *cc.p)
            {
                Chan ch;
                ch.name = cc.cstr();
                ch.type = cc.i32();
                cc.need(4);
                cc.p += 4; // pLinear + reserved
                const int xs = cc.i32(), ys = cc.i32();
                if(xs != 1 || ys != 1)
                    throw std::runtime_error("EXR: sub-sampled channels are not supported");
                chans.push_back(ch);
            }
        }
        else if(name == "compression")
            compression = d[0];
        else if(name == "dataWindow")
            std::memcpy(dw, d, 16);
        else if(name == "displayWindow")
            std::memcpy(disp, d, 16);
        else if(name == "lineOrder")
            lineOrder = d[0];
        else if(name == "pixelAspectRatio" || name == "screenWindowCenter" || name == "screenWindowWidth")
            ;
        else
            out.attributes.set(name, type, d, (size_t)size);
    }
    (void)lineOrder;
    if(chans.empty() || dw[2] < dw[0] || dw[3] < dw[1])
        throw std::runtime_error("EXR '" + path + "': missing channels or data window");
    out.width = dw[2] - dw[0] + 1;
    out.height = dw[3] - dw[1] + 1;
    out.dataX0 = dw[0];
    out.dataY0 = dw[1];
    out.displayW = disp[2] - disp[0] + 1;
    out.displayH = disp[3] - disp[1] + 1;
    for(const Chan& ch : chans)
        out.channelNames.push_back(ch.name);
    if(headerOnly)
        return;
    if(compression != C_NONE && compression != C_ZIPS && compression != C_ZIP)
        throw std::runtime_error("EXR '" + path + "': compression " + std::to_string(compression) + " is not supported (use none, zips or zip)");

    const int linesPerBlock = compression == C_ZIP ? 16 : 1;
    const int nBlocks = (out.height + linesPerBlock - 1) / linesPerBlock;
    const size_t tableOff = (size_t)(c.p - head.data());
    if(head.size() < tableOff + (size_t)nBlocks * 8)
        throw std::runtime_error("EXR: truncated offset table");
    std::vector<uint64_t> offsets(nBlocks);
    std::memcpy(offsets.data(), head.data() + tableOff, (size_t)nBlocks * 8);

    size_t bytesPerLine = 0;
    for(const Chan& ch : chans)
        bytesPerLine += (size_t)out.width * (ch.type == PT_HALF ? 2 : 4);
    out.channels.assign(chans.size(), std::vector<float>((size_t)out.width * out.height));

    const int W = out.width, H = out.height;
    std::string err;
#pragma omp parallel
    {
        std::vector<uint8_t> buf, tmp;
#pragma omp for schedule(dynamic, 4)
        for(int b = 0; b < nBlocks; ++b)
        {
            const uint64_t off = offsets[b];
            if(off + 8 > head.size())
            {
#pragma omp critical
                err = "EXR: chunk offset out of range";
                continue;
            }
            int32_t y, sz;
            std::memcpy(&y, head.data() + off, 4);
            std::memcpy(&sz, head.data() + off + 4, 4);
            const int line0 = y - out.dataY0;
            const int nLines = std::min(linesPerBlock, H - line0);
            if(sz < 0 || off + 8 + (uint64_t)sz > head.size() || line0 < 0 || nLines <= 0)
            {
#pragma omp critical
                err = "EXR: bad chunk";
                continue;
            }
            const size_t rawSize = bytesPerLine * (size_t)nLines;
            buf.resize(rawSize);
            if((size_t)sz == rawSize)
                std::memcpy(buf.data(), head.data() + off + 8, rawSize);
            else
            {
                uLongf dst = (uLongf)rawSize;
                if(uncompress(buf.data(), &dst, head.data() + off + 8, (uLong)sz) != Z_OK || dst != rawSize)
                {
#pragma omp critical
                    err = "EXR: zlib inflate failed";
                    continue;
                }
                zipUndo(buf, tmp);
            }
            const uint8_t* p = buf.data();
            for(int l = 0; l < nLines; ++l)
                for(size_t ci = 0; ci < chans.size(); ++ci)
                {
                    float* dst = out.channels[ci].data() + (size_t)(line0 + l) * W;
                    if(chans[ci].type == PT_HALF)
                    {
                        for(int x = 0; x < W; ++x)
                        {
                            uint16_t h;
                            std::memcpy(&h, p + 2 * x, 2);
                            dst[x] = halfToFloat(h);
                        }
                        p += 2 * (size_t)W;
                    }
                    else if(chans[ci].type == PT_FLOAT)
                    {
                        std::memcpy(dst, p, 4 * (size_t)W);
                        p += 4 * (size_t)W;
                    }
                    else
                    {
                        for(int x = 0; x < W; ++x)
                        {
                            uint32_t u;
                            std::memcpy(&u, p + 4 * x, 4);
                            dst[x] = (float)u;
                        }
                        p += 4 * (size_t)W;
                    }
                }
        }
    }
    if(!err.empty())
        throw std::runtime_error(err + " ('" + path + "')");
}

ExrLines::~ExrLines()
{
    if(mapBase != nullptr)
        (void)munmap(mapBase, mapBytes);
}

bool readExrLines(const std::string& path, ExrLines& out)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if(fd < 0)
        throw std::runtime_error("cannot open image '" + path + "'");
    struct stat st;
    if(fstat(fd, &st) != 0 || st.st_size < 16)
    {
        close(fd);
        throw std::runtime_error("cannot read image '" + path + "'");
    }
    const size_t fileSize = (size_t)st.st_size;
    void* const base = mmap(nullptr, fileSize, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if(base == MAP_FAILED)
        return false;
    out.mapBase = base, out.mapBytes = fileSize; // (unmapped by the destructor on every path below)
    const uint8_t* const file = static_cast<const uint8_t*>(base);
    Cursor c{file, file + fileSize};
    if(c.i32() != 20000630)
        throw std::runtime_error("'" + path + "' is not an OpenEXR file");
    const int32_t version = c.i32();
    if((version & 0xff) != 2 || (version & 0x200) || (version & 0x1000) || (version & 0x800))
        throw std::runtime_error("EXR '" + path + "': only single-part scan-line files are supported");
    std::vector<Chan> chans;
    int compression = -1;
    int dw[4] = {0, 0, -1, -1};
    for(;;)
    {
        c.need(1);
        if(*c.p == 0)
        {
            ++c.p;
            break;
        }
        const std::string name = c.cstr(), type = c.cstr();
        const int32_t size = c.i32();
        if(size < 0)
            throw std::runtime_error("EXR: bad attribute size");
        c.need((size_t)size);
        const uint8_t* d = c.p;
        c.p += size;
        if(name == "channels")
        {
            Cursor cc{d, d + size};
            while(cc.p < cc.e && *cc.p)
            {
                Chan ch;
                ch.name = cc.cstr();
                ch.type = cc.i32();
                cc.need(4);
                cc.p += 4;
                const int xs = cc.i32(), ys = cc.i32();
                if(xs != 1 || ys != 1)
                    return false;
                chans.push_back(ch);
            }
        }
        else if(name == "compression")
            compression = d[0];
        else if(name == "dataWindow")
            std::memcpy(dw, d, 16);
    }
    if(chans.empty() || dw[2] < dw[0] || dw[3] < dw[1])
        throw std::runtime_error("EXR '" + path + "': missing channels or data window");
    if(compression != C_NONE && compression != C_ZIPS && compression != C_ZIP)
        throw std::runtime_error("EXR '" + path + "': compression " + std::to_string(compression) + " is not supported (use none, zips or zip)");
    const int W = dw[2] - dw[0] + 1, H = dw[3] - dw[1] + 1;
    out.width = W, out.height = H;
    size_t bytesPerLine = 0;
    std::vector<long long> offsetOf(chans.size());
    for(size_t i = 0; i < chans.size(); ++i)
    {
        if(chans[i].type != PT_UINT && chans[i].type != PT_HALF && chans[i].type != PT_FLOAT)
            return false;
        offsetOf[i] = (long long)bytesPerLine;
        bytesPerLine += (size_t)W * (chans[i].type == PT_HALF ? 2 : 4);
    }
    auto indexOf = [&](const char* n) {
        for(size_t i = 0; i < chans.size(); ++i)
            if(chans[i].name == n)
                return (int)i;
        return -1;
    };
    const int iR = indexOf("R"), iG = indexOf("G"), iB = indexOf("B"), iA = indexOf("A"), iY = indexOf("Y");
    if(!((iR >= 0 && iG >= 0 && iB >= 0) || iY >= 0))
        throw std::runtime_error("image '" + path + "' has neither R,G,B nor Y channels");
    const int pick[4] = {iR >= 0 ? iR : iY, iG >= 0 ? iG : iY, iB >= 0 ? iB : iY, iA};
    for(int k = 0; k < 4; ++k)
        if(pick[k] >= 0)
            out.chanOffset[k] = offsetOf[(size_t)pick[k]], out.chanType[k] = chans[(size_t)pick[k]].type;

    const int linesPerBlock = compression == C_ZIP ? 16 : 1;
    const int nBlocks = (H + linesPerBlock - 1) / linesPerBlock;
    const size_t tableOff = (size_t)(c.p - file);
    if(fileSize < tableOff + (size_t)nBlocks * 8)
        throw std::runtime_error("EXR: truncated offset table");
    std::vector<uint64_t> offsets((size_t)nBlocks);
    std::memcpy(offsets.data(), file + tableOff, (size_t)nBlocks * 8);
    auto chunkHead = [&](int b, int32_t& y, int32_t& sz) {
        const uint64_t off = offsets[(size_t)b];
        if(off + 8 > fileSize)
            throw std::runtime_error("EXR: chunk offset out of range ('" + path + "')");
        std::memcpy(&y, file + off, 4);
        std::memcpy(&sz, file + off + 4, 4);
        if(sz < 0 || off + 8 + (uint64_t)sz > fileSize)
            throw std::runtime_error("EXR: bad chunk ('" + path + "')");
    };
    if(compression == C_NONE)
    {
        // lines where they lie in the mapping: chunk b must be line b, its size one line, the chunks one stride apart
        const uint64_t stride = bytesPerLine + 8;
        for(int b = 0; b < nBlocks; ++b)
        {
            int32_t y, sz;
            chunkHead(b, y, sz);
            if(y != dw[1] + b || (size_t)sz != bytesPerLine || offsets[(size_t)b] != offsets[0] + (uint64_t)b * stride)
                return false;
        }
        out.lines = file + offsets[0] + 8;
        out.lineStride = (long long)stride;
        out.bytes = (size_t)(H - 1) * stride + bytesPerLine;
        (void)madvise(base, fileSize, MADV_SEQUENTIAL);
        return true;
    }
    // ZIP / ZIPS: inflate the blocks (host cores) into one buffer of lines
    out.inflated.reset(new uint8_t[(size_t)H * bytesPerLine]);
    std::string err;
#pragma omp parallel
    {
        std::vector<uint8_t> buf, tmp;
#pragma omp for schedule(dynamic, 4)
        for(int b = 0; b < nBlocks; ++b)
        {
            try
            {
                int32_t y, sz;
                chunkHead(b, y, sz);
                const int line0 = y - dw[1];
                const int nLines = std::min(linesPerBlock, H - line0);
                if(line0 < 0 || nLines <= 0 || line0 % linesPerBlock != 0)
                    throw std::runtime_error("EXR: bad chunk ('" + path + "')");
                const size_t rawSize = bytesPerLine * (size_t)nLines;
                uint8_t* const dst = out.inflated.get() + (size_t)line0 * bytesPerLine;
                const uint8_t* const src = file + offsets[(size_t)b] + 8;
                if((size_t)sz == rawSize)
                    std::memcpy(dst, src, rawSize);
                else
                {
                    buf.resize(rawSize);
                    uLongf n = (uLongf)rawSize;
                    if(uncompress(buf.data(), &n, src, (uLong)sz) != Z_OK || n != rawSize)
                        throw std::runtime_error("EXR: zlib inflate failed ('" + path + "')");
                    zipUndo(buf, tmp);
                    std::memcpy(dst, buf.data(), rawSize);
                }
            }
            catch(const std::exception& e)
            {
#pragma omp critical
                err = e.what();
            }
        }
    }
    if(!err.empty())
        throw std::runtime_error(err);
    (void)munmap(out.mapBase, out.mapBytes);
    out.mapBase = nullptr, out.mapBytes = 0;
    out.lines = out.inflated.get();
    out.lineStride = (long long)bytesPerLine;
    out.bytes = (size_t)H * bytesPerLine;
    return true;
}

void writeExr(const std::string& path, int width, int height, const std::vector<ExrChannelIn>& channelsIn, bool storeHalf, const ExrAttributes& attributes,
              int dataX0, int dataY0, int displayW, int displayH)
{
    if(width <= 0 || height <= 0 || channelsIn.empty())
        throw std::runtime_error("writeExr: empty image");
    std::vector<ExrChannelIn> chans = channelsIn;
    std::sort(chans.begin(), chans.end(), [](const ExrChannelIn& a, const ExrChannelIn& b) { return a.name < b.name; });

    std::vector<uint8_t> h;
    const int32_t magic = 20000630, version = 2;
    h.insert(h.end(), (const uint8_t*)&magic, (const uint8_t*)&magic + 4);
    h.insert(h.end(), (const uint8_t*)&version, (const uint8_t*)&version + 4);
    {
        std::vector<uint8_t> cl;
        for(const auto& ch : chans)
        {
            cl.insert(cl.end(), ch.name.begin(), ch.name.end());
            cl.push_back(0);
            const int32_t pt = storeHalf ? PT_HALF : PT_FLOAT, one = 1;
            cl.insert(cl.end(), (const uint8_t*)&pt, (const uint8_t*)&pt + 4);
            const uint8_t pl[4] = {0, 0, 0, 0};
            cl.insert(cl.end(), pl, pl + 4);
            cl.insert(cl.end(), (const uint8_t*)&one, (const uint8_t*)&one + 4);
            cl.insert(cl.end(), (const uint8_t*)&one, (const uint8_t*)&one + 4);
        }
        cl.push_back(0);
        putAttr(h, "channels", "chlist", cl.data(), cl.size());
    }
    const uint8_t comp = C_ZIP;
    putAttr(h, "compression", "compression", &comp, 1);
    const int32_t dwin[4] = {dataX0, dataY0, dataX0 + width - 1, dataY0 + height - 1};
    putAttr(h, "dataWindow", "box2i", dwin, 16);
    const int32_t disp[4] = {0, 0, displayW - 1, displayH - 1};
    putAttr(h, "displayWindow", "box2i", disp, 16);
    const uint8_t lo = 0;
    putAttr(h, "lineOrder", "lineOrder", &lo, 1);
    const float par = 1.0f, swc[2] = {0.0f, 0.0f}, sww = 1.0f;
    putAttr(h, "pixelAspectRatio", "float", &par, 4);
    putAttr(h, "screenWindowCenter", "v2f", swc, 8);
    putAttr(h, "screenWindowWidth", "float", &sww, 4);
    for(const auto& a : attributes.list)
        putAttr(h, a.name, a.type, a.data.data(), a.data.size());
    h.push_back(0);

    const int linesPerBlock = 16;
    const int nBlocks = (height + linesPerBlock - 1) / linesPerBlock;
    const size_t bpp = storeHalf ? 2 : 4;
    const size_t bytesPerLine = (size_t)width * bpp * chans.size();
    std::vector<std::vector<uint8_t>> blocks(nBlocks);
#pragma omp parallel
    {
        std::vector<uint8_t> raw;
#pragma omp for schedule(dynamic, 4)
        for(int b = 0; b < nBlocks; ++b)
        {
            const int line0 = b * linesPerBlock, nLines = std::min(linesPerBlock, height - line0);
            raw.resize(bytesPerLine * (size_t)nLines);
            uint8_t* p = raw.data();
            for(int l = 0; l < nLines; ++l)
                for(const auto& ch : chans)
                {
                    const float* src = ch.data + (size_t)(line0 + l) * width;
                    if(storeHalf)
                    {
                        for(int x = 0; x < width; ++x)
                        {
                            const uint16_t hv = floatToHalf(src[x]);
                            std::memcpy(p + 2 * x, &hv, 2);
                        }
                    }
                    else
                        std::memcpy(p, src, 4 * (size_t)width);
                    p += bpp * (size_t)width;
                }
            zipDo(raw, blocks[b]);
        }
    }
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if(!f)
        throw std::runtime_error("cannot write '" + path + "'");
    f.write((const char*)h.data(), (std::streamsize)h.size());
    uint64_t off = h.size() + (uint64_t)nBlocks * 8;
    for(int b = 0; b < nBlocks; ++b)
    {
        f.write((const char*)&off, 8);
        off += 8 + blocks[b].size();
    }
    for(int b = 0; b < nBlocks; ++b)
    {
        const int32_t y = dataY0 + b * linesPerBlock, sz = (int32_t)blocks[b].size();
        f.write((const char*)&y, 4);
        f.write((const char*)&sz, 4);
        f.write((const char*)blocks[b].data(), sz);
    }
    if(!f)
        throw std::runtime_error("write error on '" + path + "'");
}

} // namespace avdm_host
