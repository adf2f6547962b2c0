// avdm_jpeg.hip — the pixel half of a JPEG decode on the device (image ingest): quantised DCT coefficients (entropy-decoded on
// the host, host/jpeg.cpp) -> 8-bit RGB, exactly as libjpeg / libjpeg-turbo produce it with their defaults — what the reference receives
// from OpenImageIO's JPEG reader (image/io.cpp: readImage):
//   * dequantisation and the accurate integer inverse DCT (JDCT_ISLOW, jidctint.c: 13-bit constants, two passes with 2 extra bits between
//     them, range-limited + 128),
//   * "fancy" chroma up-sampling (jdsample.c: triangle filter, h2v1 (3 a + b + 1 or 2) >> 2, h2v2 (9 a + 3 b + 3 c + d + 8 or 7) >> 4 by
//     columns of 3 a + b, edges replicated),
//   * YCbCr -> RGB with 16-bit fixed-point constants (jdcolor.c).
// Integer arithmetic throughout: bit-exact against the CPU restatement (oracle/avdm_oracle.c: avo_image_decode_jpeg), which is pinned to
// golden vectors decoded by libjpeg-turbo (tests/golden/jpeg).  HBM-bound byte work: 128 B of coefficients in and 64 B of samples out
// per block, then 1-3 B in and 3 B out per pixel.
#include "avdm_device.h"

#include <stdint.h>

#include <algorithm>

namespace avdm {

namespace {

struct JpegPlane
{
    const int16_t* coef; // blocks_h x blocks_w x 64
    uint8_t* samples;    // (8 blocks_h) x (8 blocks_w)
    int blocksW, blocksH, width, height, hExpand, vExpand;
    uint16_t quant[64];
};
struct JpegArgs
{
    JpegPlane p[3];
    int nComps, width, height, yccToRgb;
};

// jidctint.c: FIX(x) = round(x * 2^13)
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
              FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
              FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D pass over 8 values (the even / odd decomposition of jidctint.c); out[k] before descaling
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8], int shiftDc)
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0], z3 = in[4];
    int tmp0 = (z2 + z3) * (1 << shiftDc);
    int tmp1 = (z2 - z3) * (1 << shiftDc);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7], tmp1 = in[5], tmp2 = in[3], tmp3 = in[1];
    z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336, tmp1 *= FIX_2_053119869, tmp2 *= FIX_3_072711026, tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223, z2 *= -FIX_2_562915447, z3 *= -FIX_1_961570560, z4 *= -FIX_0_390180644;
    z3 += z5, z4 += z5;
    tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
    out[0] = tmp10 + tmp3, out[7] = tmp10 - tmp3, out[1] = tmp11 + tmp2, out[6] = tmp11 - tmp2;
    out[2] = tmp12 + tmp1, out[5] = tmp12 - tmp1, out[3] = tmp13 + tmp0, out[4] = tmp13 - tmp0;
}

// range_limit[(x) & RANGE_MASK] of jdmaster.c's table, centred on 128: clamp(x + 128, 0, 255) for |x| < 512, wrapping beyond like the table
__device__ __forceinline__ int idct_range_limit(int x)
{
    x &= 1023;
    return x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896));
}

// 8 lanes per block: lane j = column j in pass 1, row j in pass 2; the 8 x 8 workspace goes through LDS
__global__ void __launch_bounds__(256) jpeg_idct_kernel(JpegArgs A)
{
    __shared__ int ws[32][8][9];
    const JpegPlane& P = A.p[blockIdx.y];
    const int lane8 = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const long long nBlocks = (long long)P.blocksW * P.blocksH;
    const long long b = (long long)blockIdx.x * 32 + slot;
    const bool live = b < nBlocks;
    if(live)
    {
        const int16_t* c = P.coef + b * 64;
        int in[8], out[8];
#pragma unroll
        for(int r = 0; r < 8; ++r)
            in[r] = (int)c[8 * r + lane8] * (int)P.quant[8 * r + lane8];
        idct8(in, out, CONST_BITS);
#pragma unroll
        for(int r = 0; r < 8; ++r)
            ws[slot][r][lane8] = descale(out[r], CONST_BITS - PASS1_BITS);
    }
    __syncthreads();
    if(live)
    {
        int in[8], out[8];
#pragma unroll
        for(int k = 0; k < 8; ++k)
            in[k] = ws[slot][lane8][k];
        idct8(in, out, CONST_BITS);
        const int bx = (int)(b % P.blocksW), by = (int)(b / P.blocksW);
        uint8_t* dst = P.samples + ((long long)(8 * by + lane8) * (8 * P.blocksW) + 8 * bx);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            lo |= (unsigned)idct_range_limit(descale(out[k], CONST_BITS + PASS1_BITS + 3)) << (8 * k);
            hi |= (unsigned)idct_range_limit(descale(out[4 + k], CONST_BITS + PASS1_BITS + 3)) << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
    }
}

// one sample of a component at full-resolution pixel (x, y): as is, or through the triangle filters of jdsample.c
__device__ __forceinline__ int jpeg_sample(const JpegPlane& P, int x, int y)
{
    const int pitch = 8 * P.blocksW;
    if(P.hExpand == 1 && P.vExpand == 1)
        return P.samples[(long long)y * pitch + x];
    const int cx = x >> 1, nx = min(max((x & 1) ? cx + 1 : cx - 1, 0), P.width - 1);
    if(P.vExpand == 1)
    { // h2v1: (3 * this + neighbour + 1) >> 2 for the left output of a pair, + 2 for the right one; up to 2 input columns: replication
        const uint8_t* row = P.samples + (long long)y * pitch;
        if(P.width <= 2)
            return row[cx];
        return (3 * row[cx] + row[nx] + ((x & 1) ? 2 : 1)) >> 2;
    }
    // h2v2: columns of 3 * nearer row + further row, then (3 * this + neighbour + 8 or 7) >> 4
    const int cy = y >> 1, ny = min(max((y & 1) ? cy + 1 : cy - 1, 0), P.height - 1);
    const uint8_t *r0 = P.samples + (long long)cy * pitch, *r1 = P.samples + (long long)ny * pitch;
    if(P.width <= 2)
        return r0[cx];
    const int thisSum = 3 * r0[cx] + r1[cx], nextSum = 3 * r0[nx] + r1[nx];
    return (3 * thisSum + nextSum + ((x & 1) ? 7 : 8)) >> 4;
}

__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }

// 4 pixels (12 bytes) per lane
__global__ void __launch_bounds__(256) jpeg_color_kernel(uint8_t* dst, int dst_pitch, JpegArgs A)
{
    const int x0 = 4 * (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if(x0 >= A.width || y >= A.height)
        return;
    uint8_t px[12];
    const int n = min(4, A.width - x0);
    for(int k = 0; k < n; ++k)
    {
        const int x = x0 + k;
        const int c0 = jpeg_sample(A.p[0], x, y);
        int r = c0, g = c0, bl = c0;
        if(A.nComps == 3)
        {
            const int c1 = jpeg_sample(A.p[1], x, y), c2 = jpeg_sample(A.p[2], x, y);
            if(A.yccToRgb)
            {
                // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 32768
                const int cb = c1 - 128, cr = c2 - 128;
                r = clamp255(c0 + ((91881 * cr + 32768) >> 16));
                bl = clamp255(c0 + ((116130 * cb + 32768) >> 16));
                g = clamp255(c0 + ((-22554 * cb + 32768 + (-46802) * cr) >> 16));
            }
            else
                g = c1, bl = c2;
        }
        px[3 * k] = (uint8_t)r, px[3 * k + 1] = (uint8_t)g, px[3 * k + 2] = (uint8_t)bl;
    }
    uint8_t* o = dst + (long long)y * dst_pitch + 3 * x0;
    if(n == 4 && ((dst_pitch & 3) == 0))
    {
        unsigned w[3];
#pragma unroll
        for(int i = 0; i < 3; ++i)
            w[i] = px[4 * i] | (px[4 * i + 1] << 8) | (px[4 * i + 2] << 16) | ((unsigned)px[4 * i + 3] << 24);
        unsigned* ow = reinterpret_cast<unsigned*>(o); // 3 * x0 is a multiple of 12
        ow[0] = w[0], ow[1] = w[1], ow[2] = w[2];
    }
    else
        for(int i = 0; i < 3 * n; ++i)
            o[i] = px[i];
}

int check_components(const avdm_jpeg_component_t* comps, int n, int width, int height, int hmax, int vmax, const char** why)
{
    if(comps == nullptr || (n != 1 && n != 3))
        return *why = "avdm_image_decode_jpeg: 1 (grey) or 3 components", 1;
    if(width <= 0 || height <= 0 || hmax < 1 || vmax < 1)
        return *why = "avdm_image_decode_jpeg: empty image", 1;
    for(int i = 0; i < n; ++i)
    {
        const avdm_jpeg_component_t& c = comps[i];
        if(c.blocks_w <= 0 || c.blocks_h <= 0 || c.width <= 0 || c.height <= 0 || c.width > 8 * c.blocks_w || c.height > 8 * c.blocks_h)
            return *why = "avdm_image_decode_jpeg: bad component geometry", 1;
        const int he = c.h_samp > 0 && hmax % c.h_samp == 0 ? hmax / c.h_samp : 0, ve = c.v_samp > 0 && vmax % c.v_samp == 0 ? vmax / c.v_samp : 0;
        // what libjpeg's up-sampler does with triangle filters or a copy; other ratios (box replication there) are not built
        if(!((he == 1 && ve == 1) || (he == 2 && ve == 1) || (he == 2 && ve == 2)))
            return *why = "avdm_image_decode_jpeg: chroma sampling other than 4:4:4, 4:2:2 (h2v1) or 4:2:0 (h2v2) is not supported", 1;
        if(c.width * he < width || c.height * ve < height)
            return *why = "avdm_image_decode_jpeg: a component is smaller than the image", 1;
    }
    return 0;
}

} // namespace
} // namespace avdm

using namespace avdm;

extern "C" {

size_t avdm_image_decode_jpeg_scratch_bytes(const avdm_jpeg_component_t* comps, int n_comps)
{
    size_t total = 0;
    for(int i = 0; comps != nullptr && i < n_comps && i < 3; ++i)
        total += (((size_t)64 * (size_t)comps[i].blocks_w * (size_t)comps[i].blocks_h) + 255) & ~(size_t)255;
    return total;
}

int avdm_image_decode_jpeg(uint8_t* dst_rgb, int dst_pitch, int width, int height, const avdm_jpeg_component_t* comps, int n_comps, int hmax, int vmax,
                           int ycc_to_rgb, void* scratch, void* stream)
{
    const char* why = nullptr;
    if(check_components(comps, n_comps, width, height, hmax, vmax, &why))
        return set_error_msg(1, why);
    if(dst_rgb == nullptr || scratch == nullptr || dst_pitch < 3 * width)
        return set_error_msg(1, "avdm_image_decode_jpeg: null buffer or a pitch below 3 * width");
    hipStream_t st = (hipStream_t)stream;
    JpegArgs A;
    A.nComps = n_comps, A.width = width, A.height = height, A.yccToRgb = ycc_to_rgb ? 1 : 0;
    size_t off = 0;
    long long maxBlocks = 0;
    for(int i = 0; i < 3; ++i)
    {
        const avdm_jpeg_component_t& c = comps[i < n_comps ? i : 0];
        JpegPlane& P = A.p[i];
        P.coef = c.coef;
        P.samples = (uint8_t*)scratch + (i < n_comps ? off : 0);
        P.blocksW = c.blocks_w, P.blocksH = c.blocks_h, P.width = c.width, P.height = c.height;
        P.hExpand = hmax / c.h_samp, P.vExpand = vmax / c.v_samp;
        for(int k = 0; k < 64; ++k)
            P.quant[k] = c.quant[k];
        if(i < n_comps)
        {
            if(c.coef == nullptr)
                return set_error_msg(1, "avdm_image_decode_jpeg: null coefficient pointer");
            off += (((size_t)64 * (size_t)c.blocks_w * (size_t)c.blocks_h) + 255) & ~(size_t)255;
            maxBlocks = std::max(maxBlocks, (long long)c.blocks_w * c.blocks_h);
        }
    }
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((maxBlocks + 31) / 32), (unsigned)n_comps), dim3(256), 0, st, A);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(divUp((unsigned)width, 256u), divUp((unsigned)height, 4u)), dim3(256), 0, st, dst_rgb, dst_pitch, A);
    AVDM_LAUNCH_CHECK("avdm_image_decode_jpeg");
}

} // extern "C"
