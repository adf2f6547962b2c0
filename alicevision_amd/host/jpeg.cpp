// jpeg.cpp — see jpeg.hpp.  Marker parsing (ITU T.81 Annex B), Huffman decoding of sequential scans (Annex F.2) and of progressive scans
// (Annex G.2: spectral selection and successive approximation, with the end-of-band runs and the correction bits of the refinement passes).
#include "jpeg.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace avdm_host {

namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("JPEG: " + what); }

// natural (row-major) index of the k-th coefficient in zig-zag order
struct ZigZag
{
    uint8_t at[64 + 16];
    ZigZag()
    {
        int x = 0, y = 0;
        for(int k = 0; k < 64; ++k)
        {
            at[k] = (uint8_t)(8 * y + x);
            if((x + y) % 2 == 0)
            { // moving up-right
                if(x == 7)
                    ++y;
                else if(y == 0)
                    ++x;
                else
                    ++x, --y;
            }
            else
            { // moving down-left
                if(y == 7)
                    ++x;
                else if(x == 0)
                    ++y;
                else
                    --x, ++y;
            }
        }
        for(int k = 64; k < 80; ++k)
            at[k] = 63; // a corrupt run past the end lands on the last coefficient (libjpeg's jpeg_natural_order has the same 16 extra entries)
    }
};
const ZigZag kZigZag;

struct HuffTable
{
    bool defined = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    // decoding: codes of length l are mincode[l] .. maxcode[l], their values start at valptr[l]
    int32_t maxcode[18] = {}, mincode[17] = {};
    int valptr[17] = {};
    // 9-bit look-ahead: (length << 8) | value, 0 = longer code
    uint16_t look[512] = {};

    void build()
    {
        int code = 0, k = 0;
        for(int l = 1; l <= 16; ++l)
        {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        if(k > 256)
            fail("Huffman table with more than 256 codes");
        std::memset(look, 0, sizeof(look));
        code = 0, k = 0;
        for(int l = 1; l <= 9; ++l)
        {
            for(int i = 0; i < bits[l]; ++i, ++k, ++code)
            {
                const int first = code << (9 - l), n = 1 << (9 - l);
                if(first + n > 512)
                    fail("bad Huffman table (codes overflow their length)");
                for(int j = 0; j < n; ++j)
                    look[first + j] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
        defined = true;
    }
};

// entropy-coded data: bytes until the next marker, 0xFF00 = a stuffed 0xFF
class BitReader
{
public:
    BitReader(const uint8_t* data, size_t size) : _p(data), _end(data + size) {}
    void reset(const uint8_t* at)
    {
        _p = at;
        _acc = 0, _n = 0, _marker = 0;
    }
    const uint8_t* position() const { return _p; }
    int marker() const { return _marker; } // the marker the reader stopped in front of (0 = none yet)

    inline int peek(int n)
    {
        if(_n < n)
            fill();
        return (int)((_acc >> (_n - n)) & ((1u << n) - 1u));
    }
    inline void skip(int n) { _n -= n; }
    inline int get(int n)
    {
        if(n == 0)
            return 0;
        const int v = peek(n);
        _n -= n;
        return v;
    }
    inline int bit() { return get(1); }

    int decode(const HuffTable& t)
    {
        const int la = peek(9);
        const uint16_t e = t.look[la];
        if(e)
        {
            skip(e >> 8);
            return e & 0xff;
        }
        int code = la, l = 9;
        skip(9);
        while(true)
        {
            ++l;
            if(l > 16)
                fail("corrupt data: bad Huffman code");
            code = (code << 1) | bit();
            if(code <= t.maxcode[l])
                return t.vals[t.valptr[l] + code - t.mincode[l]];
        }
    }

private:
    const uint8_t *_p, *_end;
    uint64_t _acc = 0;
    int _n = 0, _marker = 0;

    void fill()
    {
        while(_n <= 48)
        {
            int b = 0;
            if(_marker == 0 && _p < _end)
            {
                b = *_p;
                if(b == 0xff)
                {
                    // a stuffed zero, fill bytes (FF FF ...), or a marker: past a marker the decoder sees zero bits (T.81 F.2.2.5)
                    const uint8_t* q = _p + 1;
                    while(q < _end && *q == 0xff)
                        ++q;
                    if(q < _end && *q == 0)
                        _p = q + 1;
                    else
                    {
                        _marker = q < _end ? *q : 0xd9;
                        b = 0;
                    }
                }
                else
                    ++_p;
            }
            else
                b = 0;
            _acc = (_acc << 8) | (uint64_t)b;
            _n += 8;
        }
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; } // T.81 F.2.2.1 EXTEND

struct ScanComponent
{
    int ci = 0, td = 0, ta = 0;
};

class Decoder
{
public:
    Decoder(const uint8_t* data, size_t size, JpegImage& out, bool headerOnly) : _d(data), _n(size), _img(out), _headerOnly(headerOnly) {}

    void run()
    {
        if(_n < 4 || _d[0] != 0xff || _d[1] != 0xd8)
            fail("not a JPEG file (no SOI marker)");
        size_t p = 2;
        bool sawFrame = false;
        while(true)
        {
            // next marker
            while(p < _n && _d[p] != 0xff)
                ++p; // (garbage between segments is skipped, as libjpeg does with a warning)
            while(p < _n && _d[p] == 0xff)
                ++p;
            if(p >= _n)
            {
                if(sawFrame && _sawScan)
                    break; // no EOI: what was decoded stands (libjpeg: "premature end of file" warning)
                fail("premature end of file");
            }
            const int m = _d[p++];
            if(m == 0xd9) // EOI
                break;
            if(m == 0x01 || (m >= 0xd0 && m <= 0xd7))
                continue; // TEM, stray RSTn
            if(p + 2 > _n)
                fail("truncated marker segment");
            const size_t len = ((size_t)_d[p] << 8) | _d[p + 1];
            if(len < 2 || p + len > _n)
                fail("truncated marker segment");
            const uint8_t* s = _d + p + 2;
            const size_t sl = len - 2;
            switch(m)
            {
                case 0xc0: // baseline
                case 0xc1: // extended sequential, Huffman
                case 0xc2: // progressive, Huffman
                    if(sawFrame)
                        fail("more than one frame");
                    frame(s, sl, m == 0xc2);
                    sawFrame = true;
                    break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    fail("lossless / hierarchical JPEG is not supported");
                case 0xc9: case 0xca:
                    fail("arithmetic-coded JPEG is not supported");
                case 0xc4: huffmanTables(s, sl); break;
                case 0xdb: quantTables(s, sl); break;
                case 0xdd:
                    if(sl < 2)
                        fail("bad DRI segment");
                    _restartInterval = (s[0] << 8) | s[1];
                    break;
                case 0xe0:
                    if(sl >= 5 && std::memcmp(s, "JFIF\0", 5) == 0)
                        _img.jfif = true;
                    break;
                case 0xe1:
                    if(sl >= 14 && std::memcmp(s, "Exif\0\0", 6) == 0)
                        exif(s + 6, sl - 6);
                    break;
                case 0xee:
                    if(sl >= 12 && std::memcmp(s, "Adobe", 5) == 0)
                    {
                        _img.adobe = true;
                        _img.adobeTransform = s[11];
                    }
                    break;
                case 0xda:
                {
                    if(!sawFrame)
                        fail("scan before the frame header");
                    if(_headerOnly)
                        return;
                    p = scan(s, sl, p + len);
                    _sawScan = true;
                    continue;
                }
                default: break; // APPn, COM, DNL (not used with a known height), ...
            }
            p += len;
        }
        if(!sawFrame)
            fail("no frame header");
        if(!_headerOnly && !_sawScan)
            fail("no scan");
    }

private:
    const uint8_t* _d;
    size_t _n;
    JpegImage& _img;
    bool _headerOnly, _sawScan = false;
    HuffTable _dc[4], _ac[4];
    int _restartInterval = 0;
    int _mcusX = 0, _mcusY = 0;

    void frame(const uint8_t* s, size_t n, bool progressive)
    {
        if(n < 6)
            fail("bad frame header");
        if(s[0] != 8)
            fail(std::to_string((int)s[0]) + "-bit samples are not supported (8-bit only)");
        _img.height = (s[1] << 8) | s[2];
        _img.width = (s[3] << 8) | s[4];
        const int nc = s[5];
        if(_img.width <= 0 || _img.height <= 0)
            fail("empty image (or a height defined by a DNL marker: not supported)");
        if(nc != 1 && nc != 3)
            fail(std::to_string(nc) + "-component JPEG (CMYK / YCCK) is not supported");
        if(n < 6 + 3 * (size_t)nc)
            fail("bad frame header");
        _img.progressive = progressive;
        _img.components.resize(nc);
        _img.hmax = _img.vmax = 1;
        for(int i = 0; i < nc; ++i)
        {
            JpegComponent& c = _img.components[i];
            c.id = s[6 + 3 * i];
            c.h = s[7 + 3 * i] >> 4;
            c.v = s[7 + 3 * i] & 15;
            c.tq = s[8 + 3 * i];
            if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                fail("bad sampling factors / quantisation table index");
            _img.hmax = std::max(_img.hmax, c.h);
            _img.vmax = std::max(_img.vmax, c.v);
        }
        _mcusX = (_img.width + 8 * _img.hmax - 1) / (8 * _img.hmax);
        _mcusY = (_img.height + 8 * _img.vmax - 1) / (8 * _img.vmax);
        // the header is untrusted: the coefficient planes (128 B per block) are bounded before they are allocated
        size_t totalBlocks = 0;
        for(const JpegComponent& c : _img.components)
            totalBlocks += (size_t)_mcusX * c.h * (size_t)_mcusY * c.v;
        if(totalBlocks > ((size_t)1 << 25))
            fail("image larger than 4 GiB of coefficients (" + std::to_string(_img.width) + " x " + std::to_string(_img.height) + ")");
        for(JpegComponent& c : _img.components)
        {
            c.width = (_img.width * c.h + _img.hmax - 1) / _img.hmax;
            c.height = (_img.height * c.v + _img.vmax - 1) / _img.vmax;
            c.blocksW = _mcusX * c.h;
            c.blocksH = _mcusY * c.v;
            if(!_headerOnly)
                c.coef.assign((size_t)c.blocksW * c.blocksH * 64, 0);
        }
    }

    void huffmanTables(const uint8_t* s, size_t n)
    {
        size_t i = 0;
        while(i < n)
        {
            if(i + 17 > n)
                fail("bad DHT segment");
            const int tc = s[i] >> 4, th = s[i] & 15;
            if(tc > 1 || th > 3)
                fail("bad Huffman table class / index");
            HuffTable& t = tc ? _ac[th] : _dc[th];
            int total = 0;
            t.bits[0] = 0;
            for(int l = 1; l <= 16; ++l)
            {
                t.bits[l] = s[i + l];
                total += t.bits[l];
            }
            i += 17;
            if(total > 256 || i + total > n)
                fail("bad DHT segment");
            std::memset(t.vals, 0, sizeof(t.vals));
            std::memcpy(t.vals, s + i, (size_t)total);
            i += (size_t)total;
            t.build();
        }
    }

    void quantTables(const uint8_t* s, size_t n)
    {
        size_t i = 0;
        while(i < n)
        {
            const int pq = s[i] >> 4, tq = s[i] & 15;
            if(tq > 3 || pq > 1)
                fail("bad DQT segment");
            ++i;
            const size_t need = pq ? 128 : 64;
            if(i + need > n)
                fail("bad DQT segment");
            for(int k = 0; k < 64; ++k)
                _img.quant[tq][kZigZag.at[k]] = pq ? (uint16_t)((s[i + 2 * k] << 8) | s[i + 2 * k + 1]) : s[i + k];
            i += need;
        }
    }

    void exif(const uint8_t* t, size_t n)
    {
        if(n < 8)
            return;
        const bool le = t[0] == 'I';
        auto u16 = [&](size_t o) -> unsigned { return o + 2 <= n ? (le ? t[o] | (t[o + 1] << 8) : (t[o] << 8) | t[o + 1]) : 0u; };
        auto u32 = [&](size_t o) -> unsigned { return o + 4 <= n ? (le ? u16(o) | (u16(o + 2) << 16) : (u16(o) << 16) | u16(o + 2)) : 0u; };
        const size_t ifd = u32(4);
        const unsigned count = u16(ifd);
        for(unsigned i = 0; i < count; ++i)
        {
            const size_t e = ifd + 2 + 12 * (size_t)i;
            if(e + 12 > n)
                return;
            if(u16(e) == 0x0112)
                _img.exifOrientation = (int)u16(e + 8);
        }
    }

    // ---- one scan; returns the position behind its entropy-coded data ----
    size_t scan(const uint8_t* s, size_t n, size_t dataPos)
    {
        if(n < 1)
            fail("bad scan header");
        const int ns = s[0];
        if(ns < 1 || ns > 3 || n < 1 + 2 * (size_t)ns + 3)
            fail("bad scan header");
        ScanComponent sc[3];
        for(int i = 0; i < ns; ++i)
        {
            const int id = s[1 + 2 * i];
            int ci = -1;
            for(size_t c = 0; c < _img.components.size(); ++c)
                if(_img.components[c].id == id)
                    ci = (int)c;
            if(ci < 0)
                fail("scan names an unknown component");
            sc[i].ci = ci;
            sc[i].td = s[2 + 2 * i] >> 4;
            sc[i].ta = s[2 + 2 * i] & 15;
            if(sc[i].td > 3 || sc[i].ta > 3)
                fail("bad Huffman table selector");
        }
        const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
        if(_img.progressive)
        {
            if(Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13 || (Ah != 0 && Ah != Al + 1))
                fail("bad progression parameters");
        }
        else if(Ss != 0 || Se != 63 || Ah != 0 || Al != 0)
        {
            // libjpeg warns and decodes the whole band; so do we
        }
        for(int i = 0; i < ns; ++i)
        {
            const bool needDc = !_img.progressive || (Ss == 0 && Ah == 0), needAc = !_img.progressive || Ss > 0;
            if(needDc && !_dc[sc[i].td].defined)
                fail("scan uses an undefined DC Huffman table");
            if(needAc && !_ac[sc[i].ta].defined)
                fail("scan uses an undefined AC Huffman table");
        }

        BitReader br(_d, _n);
        br.reset(_d + dataPos);
        int pred[3] = {0, 0, 0};
        int eobrun = 0;
        int restartsLeft = _restartInterval, nextRst = 0;

        // geometry: an interleaved scan walks MCUs, a single-component scan walks that component's own blocks
        const bool interleaved = ns > 1;
        const JpegComponent& c0 = _img.components[sc[0].ci];
        const int unitsX = interleaved ? _mcusX : (c0.width + 7) / 8, unitsY = interleaved ? _mcusY : (c0.height + 7) / 8;

        auto restart = [&]() {
            // the reader has stopped in front of a marker (or will: remaining bits of the byte are padding)
            const uint8_t* q = br.position();
            // find the RSTn marker from the reader's byte position
            while(q + 1 < _d + _n && !(q[0] == 0xff && q[1] != 0 && q[1] != 0xff))
                ++q;
            if(q + 1 >= _d + _n)
                fail("premature end of data (restart marker missing)");
            if(q[1] != 0xd0 + nextRst)
            {
                if(!(q[1] >= 0xd0 && q[1] <= 0xd7))
                    fail("corrupt data: expected a restart marker");
                // out-of-sequence restart marker: taken as is
            }
            nextRst = (nextRst + 1) & 7;
            br.reset(q + 2);
            pred[0] = pred[1] = pred[2] = 0;
            eobrun = 0;
            restartsLeft = _restartInterval;
        };

        for(int uy = 0; uy < unitsY; ++uy)
            for(int ux = 0; ux < unitsX; ++ux)
            {
                if(_restartInterval && restartsLeft == 0)
                    restart();
                for(int i = 0; i < ns; ++i)
                {
                    JpegComponent& c = _img.components[sc[i].ci];
                    const int bw = interleaved ? c.h : 1, bh = interleaved ? c.v : 1;
                    for(int by = 0; by < bh; ++by)
                        for(int bx = 0; bx < bw; ++bx)
                        {
                            const int X = interleaved ? ux * c.h + bx : ux, Y = interleaved ? uy * c.v + by : uy;
                            int16_t* blk = c.coef.data() + ((size_t)Y * c.blocksW + X) * 64;
                            if(!_img.progressive)
                                sequentialBlock(br, blk, _dc[sc[i].td], _ac[sc[i].ta], pred[i]);
                            else if(Ss == 0)
                            {
                                if(Ah == 0)
                                    dcFirst(br, blk, _dc[sc[i].td], pred[i], Al);
                                else if(br.bit())
                                    blk[0] = (int16_t)(blk[0] | (1 << Al));
                            }
                            else if(Ah == 0)
                                acFirst(br, blk, _ac[sc[i].ta], Ss, Se, Al, eobrun);
                            else
                                acRefine(br, blk, _ac[sc[i].ta], Ss, Se, Al, eobrun);
                        }
                }
                if(_restartInterval)
                    --restartsLeft;
            }

        // the position behind the data: the marker the reader ran into, else scan forward for one
        const uint8_t* q = br.position();
        while(q + 1 < _d + _n && !(q[0] == 0xff && q[1] != 0 && q[1] != 0xff && !(q[1] >= 0xd0 && q[1] <= 0xd7)))
            ++q;
        return (size_t)(q - _d);
    }

    static void sequentialBlock(BitReader& br, int16_t* blk, const HuffTable& dc, const HuffTable& ac, int& pred)
    {
        int s = br.decode(dc);
        if(s > 15)
            fail("corrupt data: bad DC category");
        if(s)
            pred += extend(br.get(s), s);
        blk[0] = (int16_t)pred;
        for(int k = 1; k < 64;)
        {
            const int rs = br.decode(ac), r = rs >> 4;
            s = rs & 15;
            if(s)
            {
                k += r;
                blk[kZigZag.at[k]] = (int16_t)extend(br.get(s), s);
                ++k;
            }
            else
            {
                if(r != 15)
                    break; // EOB
                k += 16;
            }
        }
    }

    static void dcFirst(BitReader& br, int16_t* blk, const HuffTable& dc, int& pred, int Al)
    {
        const int s = br.decode(dc);
        if(s > 15)
            fail("corrupt data: bad DC category");
        if(s)
            pred += extend(br.get(s), s);
        blk[0] = (int16_t)(pred * (1 << Al));
    }

    static void acFirst(BitReader& br, int16_t* blk, const HuffTable& ac, int Ss, int Se, int Al, int& eobrun)
    {
        if(eobrun > 0)
        {
            --eobrun;
            return;
        }
        for(int k = Ss; k <= Se; ++k)
        {
            const int rs = br.decode(ac), r = rs >> 4, s = rs & 15;
            if(s)
            {
                k += r;
                blk[kZigZag.at[k]] = (int16_t)(extend(br.get(s), s) * (1 << Al));
            }
            else
            {
                if(r == 15)
                    k += 15; // ZRL: sixteen zeros
                else
                {
                    eobrun = 1 << r;
                    if(r)
                        eobrun += br.get(r);
                    --eobrun; // this block is the first of the run
                    break;
                }
            }
        }
    }

    static void acRefine(BitReader& br, int16_t* blk, const HuffTable& ac, int Ss, int Se, int Al, int& eobrun)
    {
        const int p1 = 1 << Al, m1 = -(1 << Al);
        int k = Ss;
        if(eobrun == 0)
        {
            for(; k <= Se; ++k)
            {
                const int rs = br.decode(ac);
                int r = rs >> 4, s = rs & 15;
                if(s)
                    s = br.bit() ? p1 : m1; // (s == 1 in a valid stream) the new coefficient's sign
                else if(r != 15)
                {
                    eobrun = 1 << r;
                    if(r)
                        eobrun += br.get(r);
                    break; // the rest of this block is handled below, as the first block of the run
                }
                // skip r coefficients that are still zero, correcting the non-zero ones passed on the way
                do
                {
                    int16_t& c = blk[kZigZag.at[k]];
                    if(c != 0)
                    {
                        if(br.bit() && (c & p1) == 0)
                            c = (int16_t)(c + (c >= 0 ? p1 : m1));
                    }
                    else if(--r < 0)
                        break;
                    ++k;
                } while(k <= Se);
                if(s)
                    blk[kZigZag.at[k]] = (int16_t)s;
            }
        }
        if(eobrun > 0)
        {
            for(; k <= Se; ++k)
            {
                int16_t& c = blk[kZigZag.at[k]];
                if(c != 0 && br.bit() && (c & p1) == 0)
                    c = (int16_t)(c + (c >= 0 ? p1 : m1));
            }
            --eobrun;
        }
    }
};

} // namespace

bool JpegImage::storedAsRgb() const
{
    if(components.size() != 3)
        return false;
    // libjpeg's default_decompress_parms (jdapimin.c): JFIF says YCbCr; an Adobe marker says what its transform flag says; with neither,
    // component identifiers 1 2 3 mean YCbCr and 'R' 'G' 'B' mean RGB (anything else: YCbCr)
    if(jfif)
        return false;
    if(adobe)
        return adobeTransform == 0;
    return components[0].id == 'R' && components[1].id == 'G' && components[2].id == 'B';
}

void readJpegMemory(const uint8_t* data, size_t size, JpegImage& out, bool headerOnly)
{
    out = JpegImage();
    Decoder(data, size, out, headerOnly).run();
}

void readJpeg(const std::string& filename, JpegImage& out, bool headerOnly)
{
    std::ifstream f(filename, std::ios::binary);
    if(!f)
        throw std::runtime_error("JPEG: cannot open '" + filename + "'");
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> bytes((size_t)std::max<std::streamoff>(n, 0));
    if(!bytes.empty())
        f.read(reinterpret_cast<char*>(bytes.data()), n);
    readJpegMemory(bytes.data(), bytes.size(), out, headerOnly);
}

} // namespace avdm_host
