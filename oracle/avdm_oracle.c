/*
 * avdm_oracle.c — CPU restatement (the parity ORACLE) of AliceVision's depthMap CUDA kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (alicevision_amd/, include/) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PINNED TO THE REFERENCE'S OWN CODE (round 2): the reference (/root/reference, snapshot 2024-10-24) holds no test, golden vector or
 * fixture for src/aliceVision/depthMap and its module cannot be built as such (CUDA, Boost, Eigen, OpenImageIO), but
 * its kernel-launch layer — the 19 cuda_* wrappers, every kernel and device helper, DeviceMipmapImage, buildCustomPatchPattern —
 * compiles UNCHANGED with g++ over the stand-in CUDA runtime of oracle/ref/shim (oracle/_ref, recipe oracle/ref/Makefile).  This
 * file follows the reference sources function by function (each function cites the file:line it restates, paths relative to
 * /root/reference/src/aliceVision/depthMap) and tests/test_oracle_ref.py holds it to that library BIT FOR BIT: whole tiles through
 * every stage in both filter modes, and the committed vectors tests/golden/ that library produced (tests/golden/make_golden.py).
 * What remains restated on both sides — and is named "unpinned" in DESIGN.md — is NVIDIA's part: the texture unit's filtering
 * arithmetic and the fast-math intrinsics; plus fillHostCameraParameters (needs MultiViewParams), checked by projection identities.
 * PARITY UNPINNED for one function: avo_image_resize restates OpenImageIO's default resize filter (a third-party dependency that is not
 * under /root/reference and not installed here) from its published source; see the comment above it.
 *
 * Arithmetic conventions (see DESIGN.md §"Texture unit restatement" and §"Fast-math intrinsics"):
 *   - compiled with -ffp-contract=off: every fp32 operation is the IEEE operation written;
 *   - CUDA fast intrinsics are restated by their exact counterparts: __fdividef -> '/', __expf -> expf,
 *     __fsqrt_rn -> sqrtf, norm3df -> sqrtf(x*x+y*y+z*z);
 *   - tex2DLod on an fp16 mip-mapped array is restated in software: clamp addressing, bilinear within a
 *     level, linear between levels; weights either exact fp32 (AVDM_FILTER_EXACT) or quantised to the
 *     texture unit's 1.8 fixed point (AVDM_FILTER_CUDA_FIXED8, per the CUDA Programming Guide, appendix
 *     "Texture Fetching": "alpha, beta are stored in 9-bit fixed point format with 8 bits of fractional value");
 *   - the one transcendental inside the integer-valued SGM recurrence (the adaptive P2 sigmoid) uses the
 *     fully specified avo_exp_p2() so that the stage can be compared BIT-EXACTLY (CUDA's own expf has a
 *     2-ulp error bound, so any <=1-ulp evaluation is as faithful as the reference is to itself).
 *
 * Volume layout matches the product (z-fastest, see include/avdm.h); layout is not part of the algorithm.
 */
#include "avdm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* ------------------------------------------------------------------------------------------------
 * fp16 <-> fp32 (IEEE binary16, round-to-nearest-even) — __float2half / __half2float
 * ---------------------------------------------------------------------------------------------- */
static inline uint16_t f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if(x >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));
    if(x >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if(x < 0x33000001u) /* < 2^-25 (or exactly 2^-25 -> ties to even = 0) */
        return (uint16_t)sign;
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t base;
    if(e < -14)
    { /* subnormal half */
        shift = 13 + (-14 - e);
        base = 0;
    }
    else
    {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7fffffu;
    }
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if(rem > half || (rem == half && (r & 1u)))
        r++;
    return (uint16_t)(sign | (base + r)); /* mantissa carry correctly bumps the exponent */
}

static inline float h2f(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if(e == 0)
    {
        if(m == 0)
            x = sign;
        else
        {
            int ee = -1;
            do
            {
                m <<= 1;
                ee++;
            } while(!(m & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - ee) << 23) | ((m & 0x3ffu) << 13);
        }
    }
    else if(e == 31)
        x = sign | 0x7f800000u | (m << 13);
    else
        x = sign | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

uint16_t avo_float_to_half(float f) { return f2h(f); }
float avo_half_to_float(uint16_t h) { return h2f(h); }

/* ------------------------------------------------------------------------------------------------
 * cuda/device/operators.cuh, matrix.cuh
 * ---------------------------------------------------------------------------------------------- */
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f2 mk2(float x, float y) { f2 r = {x, y}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, float d) { return mk3(a.x * d, a.y * d, a.z * d); }
static inline f3 div3(f3 a, float d) { return mk3(a.x / d, a.y / d, a.z / d); }
static inline f3 cam3(const float* v) { return mk3(v[0], v[1], v[2]); }

/* matrix.cuh:28-31 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* matrix.cuh:38-41 */
static inline float size3(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline float size2(f2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
/* matrix.cuh:58-61 */
static inline f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* matrix.cuh:63-75 (__fdividef(1, sqrtf(dot)) restated as an exact division) */
static inline f3 normalize3(f3 a)
{
    const float dInv = 1.0f / sqrtf(dot3(a, a));
    return mk3(a.x * dInv, a.y * dInv, a.z * dInv);
}
/* matrix.cuh:89-94 */
static inline f3 M3x3mulV3(const float* M, f3 V)
{
    return mk3(M[0] * V.x + M[3] * V.y + M[6] * V.z, M[1] * V.x + M[4] * V.y + M[7] * V.z, M[2] * V.x + M[5] * V.y + M[8] * V.z);
}
/* matrix.cuh:96-101 */
static inline f3 M3x3mulV2(const float* M, f2 V)
{
    return mk3(M[0] * V.x + M[3] * V.y + M[6], M[1] * V.x + M[4] * V.y + M[7], M[2] * V.x + M[5] * V.y + M[8]);
}
/* matrix.cuh:103-108 */
static inline f3 M3x4mulV3(const float* M, f3 V)
{
    return mk3(M[0] * V.x + M[3] * V.y + M[6] * V.z + M[9], M[1] * V.x + M[4] * V.y + M[7] * V.z + M[10],
               M[2] * V.x + M[5] * V.y + M[8] * V.z + M[11]);
}
/* matrix.cuh:117-126 */
static inline f2 project3DPoint(const float* M3x4, f3 V)
{
    const f3 p = M3x4mulV3(M3x4, V);
    const float pzInv = 1.0f / p.z;
    return mk2(p.x * pzInv, p.y * pzInv);
}
/* matrix.cuh:182-189 */
static inline f3 linePlaneIntersect(f3 linePoint, f3 lineVect, f3 planePoint, f3 planeNormal)
{
    const float k = (dot3(planePoint, planeNormal) - dot3(planeNormal, linePoint)) / dot3(planeNormal, lineVect);
    return add3(linePoint, mul3(lineVect, k));
}
/* matrix.cuh:196-199 */
static inline f3 closestPointToLine3D(f3 point, f3 linePoint, f3 lineVectNormalized)
{
    return add3(linePoint, mul3(lineVectNormalized, dot3(lineVectNormalized, sub3(point, linePoint))));
}
/* matrix.cuh:201-204 */
static inline float pointLineDistance3D(f3 point, f3 linePoint, f3 lineVectNormalized)
{
    return size3(cross3(lineVectNormalized, sub3(linePoint, point)));
}
/* matrix.cuh:218-230 */
static inline float angleBetwABandAC(f3 A, f3 B, f3 C)
{
    f3 V1 = normalize3(sub3(B, A));
    f3 V2 = normalize3(sub3(C, A));
    const double x = (double)(V1.x * V2.x + V1.y * V2.y + V1.z * V2.z);
    double a = acos(x);
    a = isinf(a) ? 0.0 : a;
    return (float)(fabs(a) / (3.14159265358979323846 / 180.0));
}
/* matrix.cuh:334-337 */
static inline float sigmoidf_(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
    return zeroVal + (endVal - zeroVal) * (1.0f / (1.0f + expf(10.0f * ((xval - sigMid) / sigwidth))));
}
/* matrix.cuh:343-346 */
static inline float sigmoid2f_(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
    return zeroVal + (endVal - zeroVal) * (1.0f / (1.0f + expf(10.0f * ((sigMid - xval) / sigwidth))));
}

/* Fully specified exp for the SGM P2 sigmoid (see file header).  Cody–Waite reduction with
 * ln2 = 0.693359375 + (-2.12194440e-4), Cephes degree-5 minimax polynomial, every operation a separately
 * rounded fp32 op in the order written; result = ldexp(poly, n).  |error| < 1 ulp on [-80, 88]. */
float avo_exp_p2(float x)
{
    if(x > 88.0f)
        x = 88.0f;
    if(x < -80.0f)
        x = -80.0f;
    const float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float r2 = r * r;
    float y = p * r2 + r;
    y = y + 1.0f;
    return ldexpf(y, (int)n);
}

/* ------------------------------------------------------------------------------------------------
 * Software texture unit: tex2DLod<float4> on the fp16 pyramid
 *   (cudaTextureDesc of deviceMipmappedArray.cu:329-351: normalized coords, linear + mip-linear, clamp)
 * ---------------------------------------------------------------------------------------------- */
static inline float quant8(float a) { return floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f); }
/* texel index of a (floored) filter coordinate: coordinates far outside the image (the wrapped unsigned taps of the pyramid kernels,
 * see avo_pyramid_build_levels) must clamp to the edge they are beyond, not overflow the int conversion */
static inline int tex_index(float f) { return f < -4.0f ? -4 : (f > 1.0e9f ? 1000000000 : (int)f); }

static inline f4 texel(const avdm_pyramid_t* p, int level, int x, int y)
{
    const int W = p->width[level], H = p->height[level];
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    const uint16_t* t = (const uint16_t*)((const char*)p->base + p->offset[level] + (long long)y * p->pitch[level]) + 4 * (long long)x;
    f4 r = {h2f(t[0]), h2f(t[1]), h2f(t[2]), h2f(t[3])};
    return r;
}

static inline f4 tex_level(const avdm_pyramid_t* p, int level, float u, float v)
{
    const float x = u * (float)p->width[level] - 0.5f;
    const float y = v * (float)p->height[level] - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    if(p->filter_mode == AVDM_FILTER_CUDA_FIXED8)
    {
        a = quant8(a);
        b = quant8(b);
    }
    const int i = tex_index(fx), j = tex_index(fy);
    const f4 t00 = texel(p, level, i, j), t10 = texel(p, level, i + 1, j), t01 = texel(p, level, i, j + 1), t11 = texel(p, level, i + 1, j + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    f4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}

static f4 tex2DLod(const avdm_pyramid_t* p, float u, float v, float lod)
{
    const float maxl = (float)(p->levels - 1);
    if(!(lod > 0.0f))
        lod = 0.0f;
    if(lod > maxl)
        lod = maxl;
    const float fl = floorf(lod);
    float g = lod - fl;
    if(p->filter_mode == AVDM_FILTER_CUDA_FIXED8)
        g = quant8(g);
    const int l0 = (int)fl;
    const f4 c0 = tex_level(p, l0, u, v);
    if(g == 0.0f || l0 + 1 >= p->levels)
        return c0;
    const f4 c1 = tex_level(p, l0 + 1, u, v);
    f4 r;
    r.x = (1.0f - g) * c0.x + g * c1.x;
    r.y = (1.0f - g) * c0.y + g * c1.y;
    r.z = (1.0f - g) * c0.z + g * c1.z;
    r.w = (1.0f - g) * c0.w + g * c1.w;
    return r;
}

void avo_tex2dlod(const avdm_pyramid_t* p, float u, float v, float lod, float out[4])
{
    const f4 c = tex2DLod(p, u, v, lod);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}

/* DeviceMipmapImage.cpp:92-99 */
static inline float pyr_level(const avdm_pyramid_t* p, int downscale) { return log2f((float)downscale / (float)p->min_downscale); }
/* DeviceMipmapImage.cpp:101-108 */
static inline int pyr_dim_w(const avdm_pyramid_t* p, int downscale) { return (p->width0 + downscale - 1) / downscale; }
static inline int pyr_dim_h(const avdm_pyramid_t* p, int downscale) { return (p->height0 + downscale - 1) / downscale; }

/* ------------------------------------------------------------------------------------------------
 * Image side
 * ---------------------------------------------------------------------------------------------- */
/* DeviceMipmapImage.cpp:28-35, deviceMipmappedArray.cu:242-250 (levels: floor halving) */
int avo_pyramid_layout(avdm_pyramid_t* p, int width, int height, int min_downscale, int max_downscale, int filter_mode)
{
    if(min_downscale < 1 || max_downscale < min_downscale)
        return 1;
    memset(p, 0, sizeof(*p));
    p->filter_mode = filter_mode;
    p->min_downscale = min_downscale;
    p->width0 = width;
    p->height0 = height;
    int levels = (int)log2((double)(max_downscale / min_downscale)) + 1;
    if(levels > AVDM_MAX_LEVELS)
        levels = AVDM_MAX_LEVELS;
    int w = (width + min_downscale - 1) / min_downscale, h = (height + min_downscale - 1) / min_downscale;
    long long off = 0;
    int l = 0;
    for(; l < levels && w > 0 && h > 0; ++l)
    {
        p->width[l] = w;
        p->height[l] = h;
        p->pitch[l] = ((w * 8 + 127) / 128) * 128;
        p->offset[l] = off;
        off += (long long)p->pitch[l] * h;
        w /= 2;
        h /= 2;
    }
    p->levels = l;
    p->bytes = off;
    return 0;
}

/* DeviceCache.cpp:249-280 */
void avo_image_rgba_f32_to_f16x255(uint16_t* out, int out_pitch, const float* in, int in_pitch, int width, int height)
{
#pragma omp parallel for schedule(static)
    for(int y = 0; y < height; ++y)
    {
        const float* s = (const float*)((const char*)in + (long long)y * in_pitch);
        uint16_t* d = (uint16_t*)((char*)out + (long long)y * out_pitch);
        for(int x = 0; x < 4 * width; ++x)
            d[x] = f2h(s[x] * 255.0f);
    }
}

/* color.cuh:65-70 */
static inline f3 rgb2xyz(f3 c)
{
    return mk3(0.4124564f * c.x + 0.3575761f * c.y + 0.1804375f * c.z, 0.2126729f * c.x + 0.7151522f * c.y + 0.0721750f * c.z,
               0.0193339f * c.x + 0.1191920f * c.y + 0.9503041f * c.z);
}
/* color.cuh:124-141 */
static inline f3 xyz2lab(f3 c)
{
    const f3 r = mk3(c.x / 0.95047f, c.y, c.z / 1.08883f);
    const f3 f = mk3((r.x > 216.0f / 24389.0f ? cbrtf(r.x) : (24389.0f / 27.0f * r.x + 16.0f) / 116.0f),
                     (r.y > 216.0f / 24389.0f ? cbrtf(r.y) : (24389.0f / 27.0f * r.y + 16.0f) / 116.0f),
                     (r.z > 216.0f / 24389.0f ? cbrtf(r.z) : (24389.0f / 27.0f * r.z + 16.0f) / 116.0f));
    f3 out = mk3(116.0f * f.y - 16.0f, 500.0f * (f.x - f.y), 200.0f * (f.y - f.z));
    out.x = out.x * 2.55f;
    out.y = out.y * 2.55f;
    out.z = out.z * 2.55f;
    return out;
}

/* deviceColorConversion.cu:16-42 (rgb2lab_kernel) */
void avo_rgb2lab(uint16_t* img, int pitch, int width, int height)
{
    const float d = 1 / 255.f;
#pragma omp parallel for schedule(static)
    for(int y = 0; y < height; ++y)
    {
        uint16_t* row = (uint16_t*)((char*)img + (long long)y * pitch);
        for(int x = 0; x < width; ++x)
        {
            uint16_t* t = row + 4 * x;
            const f3 lab = xyz2lab(rgb2xyz(mk3(h2f(t[0]) * d, h2f(t[1]) * d, h2f(t[2]) * d)));
            t[0] = f2h(lab.x);
            t[1] = f2h(lab.y);
            t[2] = f2h(lab.z);
        }
    }
}

/* deviceGaussianFilter.cu:240-252: taps exp(-(x*x)/(2*delta*delta)), delta = 1, radius = scale+1 */
static inline float getGauss(int scale, int idx)
{
    const int radius = scale + 1;
    const int x = idx - radius;
    return expf(-(x * x) / (2 * 1.0f * 1.0f));
}

/* non-normalized, non-mipmapped linear texture over a plain fp16x4 image (CudaRGBATexture, memory.hpp:918-961) */
static inline f4 tex2D_plain(const uint16_t* img, int pitch, int W, int H, int filter_mode, float xf, float yf)
{
    avdm_pyramid_t p;
    memset(&p, 0, sizeof(p));
    p.base = (void*)img;
    p.levels = 1;
    p.filter_mode = filter_mode;
    p.width[0] = W;
    p.height[0] = H;
    p.pitch[0] = pitch;
    /* unnormalised coords: x = xf - 0.5 — reuse tex_level by normalising exactly in the filter domain */
    const float x = xf - 0.5f, y = yf - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    if(filter_mode == AVDM_FILTER_CUDA_FIXED8)
    {
        a = quant8(a);
        b = quant8(b);
    }
    const int i = tex_index(fx), j = tex_index(fy);
    const f4 t00 = texel(&p, 0, i, j), t10 = texel(&p, 0, i + 1, j), t01 = texel(&p, 0, i, j + 1), t11 = texel(&p, 0, i + 1, j + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    f4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}

/* deviceGaussianFilter.cu:44-80 (downscaleWithGaussianBlur_kernel) */
void avo_downscale_with_gaussian_blur(uint16_t* out, int out_pitch, int out_w, int out_h, const uint16_t* in, int in_pitch, int in_w, int in_h,
                                      int downscale, int gaussRadius, int filter_mode)
{
#pragma omp parallel for schedule(static)
    for(int y = 0; y < out_h; ++y)
        for(int x = 0; x < out_w; ++x)
        {
            const float s = (float)downscale * 0.5f;
            f4 acc = {0, 0, 0, 0};
            float sumFactor = 0.0f;
            for(int i = -gaussRadius; i <= gaussRadius; i++)
                for(int j = -gaussRadius; j <= gaussRadius; j++)
                {
                    /* `float(x * downscale + j)` with x UNSIGNED (deviceGaussianFilter.cu:54-55,65): taps left of / above the image wrap
                     * to ~4.29e9 and clamp to the RIGHT / BOTTOM edge — found by oracle/_ref, restated as the reference computes it */
                    const f4 c = tex2D_plain(in, in_pitch, in_w, in_h, filter_mode, (float)((unsigned)x * (unsigned)downscale + (unsigned)j) + s,
                                             (float)((unsigned)y * (unsigned)downscale + (unsigned)i) + s);
                    const float factor = getGauss(downscale - 1, i + gaussRadius) * getGauss(downscale - 1, j + gaussRadius);
                    acc.x = acc.x + c.x * factor;
                    acc.y = acc.y + c.y * factor;
                    acc.z = acc.z + c.z * factor;
                    acc.w = acc.w + c.w * factor;
                    sumFactor += factor;
                }
            uint16_t* t = (uint16_t*)((char*)out + (long long)y * out_pitch) + 4 * x;
            t[0] = f2h(acc.x / sumFactor);
            t[1] = f2h(acc.y / sumFactor);
            t[2] = f2h(acc.z / sumFactor);
            t[3] = f2h(acc.w / sumFactor);
        }
}

/* deviceMipmappedArray.cu:20-92 (createMipmappedArrayLevel_kernel<2>), host loop :242-327 */
void avo_pyramid_build_levels(const avdm_pyramid_t* p)
{
    for(int l = 1; l < p->levels; ++l)
    {
        const int width = p->width[l], height = p->height[l];
        avdm_pyramid_t prev = *p; /* previous level bound as a 1-level linear/clamp/normalised texture */
        prev.base = (char*)p->base + p->offset[l - 1];
        prev.levels = 1;
        prev.width[0] = p->width[l - 1];
        prev.height[0] = p->height[l - 1];
        prev.pitch[0] = p->pitch[l - 1];
        prev.offset[0] = 0;
#pragma omp parallel for schedule(static)
        for(int y = 0; y < height; ++y)
            for(int x = 0; x < width; ++x)
            {
                const float px = 1.f / (float)width;
                const float py = 1.f / (float)height;
                f4 sum = {0, 0, 0, 0};
                float sumFactor = 0.0f;
                for(int i = -2; i <= 2; i++)
                    for(int j = -2; j <= 2; j++)
                    {
                        const float factor = getGauss(1, i + 2) * getGauss(1, j + 2);
                        /* `(x + j + 0.5f)` with x UNSIGNED (deviceMipmappedArray.cu:28-29,52-53): x + j is evaluated in unsigned
                         * arithmetic, so the taps left of / above the image wrap to ~4.29e9 and the clamp addressing sends them to the
                         * RIGHT / BOTTOM edge texel — the two first rows and columns of every level carry that quirk (found by oracle/_ref) */
                        const float u = ((float)((unsigned)x + (unsigned)j) + 0.5f) * px;
                        const float v = ((float)((unsigned)y + (unsigned)i) + 0.5f) * py;
                        const f4 c = tex_level(&prev, 0, u, v);
                        sum.x = sum.x + c.x * factor;
                        sum.y = sum.y + c.y * factor;
                        sum.z = sum.z + c.z * factor;
                        sum.w = sum.w + c.w * factor;
                        sumFactor += factor;
                    }
                uint16_t* t = (uint16_t*)((char*)p->base + p->offset[l] + (long long)y * p->pitch[l]) + 4 * x;
                t[0] = f2h(sum.x / sumFactor);
                t[1] = f2h(sum.y / sumFactor);
                t[2] = f2h(sum.z / sumFactor);
                t[3] = f2h(sum.w / sumFactor);
            }
    }
}

/* DeviceCache.cpp:222-281 + DeviceMipmapImage.cpp:28-90 */
int avo_pyramid_fill(const avdm_pyramid_t* p, const float* rgba, int in_pitch)
{
    const int W = p->width0, H = p->height0;
    if(p->min_downscale > 1)
    {
        const int pitch = W * 8;
        uint16_t* full = (uint16_t*)malloc((size_t)pitch * H);
        if(!full)
            return 1;
        avo_image_rgba_f32_to_f16x255(full, pitch, rgba, in_pitch, W, H);
        avo_downscale_with_gaussian_blur((uint16_t*)p->base, p->pitch[0], p->width[0], p->height[0], full, pitch, W, H, p->min_downscale,
                                         p->min_downscale, p->filter_mode);
        free(full);
    }
    else
        avo_image_rgba_f32_to_f16x255((uint16_t*)p->base, p->pitch[0], rgba, in_pitch, W, H);
    avo_rgb2lab((uint16_t*)p->base, p->pitch[0], p->width[0], p->height[0]);
    avo_pyramid_build_levels(p);
    return 0;
}

/* ---- image ingest: the --downscale resize ---------------------------------------------------------------------------------------------
 * Reference call site: mvsUtils/fileIO.cpp:432-441 (loadImage: `imageAlgo::resizeImage(processScale, img, bmpr)`), which is
 * image/imageAlgo.cpp:220-235, 326-368: out = in.width / downscale x in.height / downscale (integer division) and
 * `oiio::ImageBufAlgo::resize(outBuf, inBuf, filter = "", filterSize = 0, ROI::All())`.
 *
 * THIRD-PARTY ALGORITHM, NOT IN /root/reference: OpenImageIO (pinned only by the CI image alicevision/alicevision-deps:2024.10.22,
 * i.e. OpenImageIO 2.5.x).  Restated from its published source (libOpenImageIO/imagebufalgo_xform.cpp `resize_` / `get_resize_filter`,
 * libutil/filter.cpp `FilterLanczos3_2D`); parity for this function is UNPINNED (no OpenImageIO here to generate vectors):
 *   - an empty filter name selects "lanczos3" when shrinking (and "blackman-harris" when enlarging, which this path never does), with
 *     the filter's own width (6 destination pixels) when filterSize is 0;
 *   - destination pixel x samples the source at  src_xf = (x + 0.5) / dst_w * src_w,  split into src_x = floor and frac; the taps are the
 *     source pixels src_x - rad .. src_x + rad with  rad = ceil(3 / ratio),  ratio = dst_w / src_w,  weights
 *     lanczos3(ratio * (i - rad - (frac - 0.5))), normalised by their sum (a zero sum leaves the pixel black); rows likewise;
 *   - the filter is separable: pel += (wy * wx) * src over j (rows) outer, i (columns) inner, zero products skipped, source
 *     coordinates clamped to the image (ImageBuf::WrapClamp);
 *   - lanczos3(x): 0 beyond 3, 1 below 1e-4, else 3 / (x^2 pi^2) * sin(x pi / 3)-based product with sin(pi x) obtained from
 *     sin(pi x / 3) through the triple-angle identity (filter.cpp: "full-precision sin(), but use the trig identity"). */
static float avo_lanczos3(float x)
{
    const float a = 3.0f;
    const float ainv = 1.0f / a;
    const float m_pi = (float)3.14159265358979323846; /* float(M_PI) */
    x = fabsf(x);
    if(x > a)
        return 0.0f;
    if(x < 0.0001f)
        return 1.0f;
    const float s1 = sinf(x * ainv * m_pi);
    const float s3 = (-4.0f * s1 * s1 + 3.0f) * s1;
    return a / (x * x * (m_pi * m_pi)) * s3 * s1;
}

/* tap table of one axis: returns the number of taps per destination pixel; weights[d * taps + i] (normalised), first[d] = source index
 * of tap 0 (may be negative / past the end: the caller clamps) */
int avo_image_resize_taps(int dst_n, int src_n, float* weights, int* first)
{
    const float srcf = (float)src_n, dstf = (float)dst_n;
    const float ratio = dstf / srcf;
    const float dstpixel = 1.0f / dstf;
    const float filterrad = 6.0f * fmaxf(1.0f, ratio) / 2.0f; /* get_resize_filter: width = fd.width * max(1, ratio) */
    const float wscale = 6.0f / (6.0f * fmaxf(1.0f, ratio));  /* FilterLanczos3_2D: m_wscale = 6 / width */
    const int rad = (int)ceilf(filterrad / ratio);
    const int taps = 2 * rad + 1;
    if(weights == NULL || first == NULL)
        return taps;
    for(int d = 0; d < dst_n; ++d)
    {
        const float s = ((float)d - 0.0f + 0.5f) * dstpixel;
        const float src_f = 0.0f + s * srcf;
        const float fl = floorf(src_f);
        const int src_i = (int)fl;
        const float frac = src_f - fl;
        float total = 0.0f;
        float* w = weights + (size_t)d * taps;
        for(int i = 0; i < taps; ++i)
        {
            w[i] = avo_lanczos3((ratio * ((float)(i - rad) - (frac - 0.5f))) * wscale);
            total += w[i];
        }
        if(total != 0.0f)
            for(int i = 0; i < taps; ++i)
                w[i] /= total;
        first[d] = src_i - rad;
    }
    return taps;
}

/* image::readImage(path, img, EImageColorSpace::LINEAR) for an integer file, as mvsUtils::loadImage receives it (mvsUtils/fileIO.cpp:386-446):
 * the decoder's samples scaled to [0, 1], the colour channels through OpenImageIO's sRGB decoding (third party, restated from its
 * published source — fmath.h sRGB_to_linear: x <= 0.04045 ? x * (1 / 12.92) : powf((x + 0.055) * (1 / 1.055), 2.4) — parity unpinned),
 * alpha untouched, one channel replicated, a missing alpha = 1.  16-bit samples in host byte order. */
int avo_image_decode_integer(float* dst, int dst_pitch, const void* src, int src_pitch, int width, int height, int channels, int bits, int srgb_to_linear)
{
    if(width <= 0 || height <= 0 || channels < 1 || channels > 4 || (bits != 8 && bits != 16))
        return 1;
    const float inv = 1.0f / (float)((bits == 8 ? 256 : 65536) - 1);
    for(int y = 0; y < height; ++y)
    {
        float* o = (float*)((char*)dst + (long long)y * dst_pitch);
        const unsigned char* row = (const unsigned char*)src + (long long)y * src_pitch;
        for(int x = 0; x < width; ++x)
        {
            float v[4];
            for(int c = 0; c < channels; ++c)
            {
                const unsigned s = bits == 8 ? row[(size_t)x * channels + c] : ((const uint16_t*)row)[(size_t)x * channels + c];
                const float f = (float)s * inv;
                const int isAlpha = (channels == 2 && c == 1) || (channels == 4 && c == 3);
                v[c] = (isAlpha || !srgb_to_linear) ? f : (f <= 0.04045f ? f * (1.0f / 12.92f) : powf((f + 0.055f) * (1.0f / 1.055f), 2.4f));
            }
            if(channels >= 3)
            {
                o[4 * x + 0] = v[0], o[4 * x + 1] = v[1], o[4 * x + 2] = v[2];
                o[4 * x + 3] = channels == 4 ? v[3] : 1.0f;
            }
            else
            {
                o[4 * x + 0] = o[4 * x + 1] = o[4 * x + 2] = v[0];
                o[4 * x + 3] = channels == 2 ? v[1] : 1.0f;
            }
        }
    }
    return 0;
}

/* ---- JPEG: coefficients -> RGB, as libjpeg / libjpeg-turbo decode with their defaults -----------------------------------------------
 * The library behind OpenImageIO's JPEG reader is not part of /root/reference; this restates its published arithmetic in the library's own
 * order of work (a block at a time, then a row at a time with the context rows its main controller provides) and is pinned to golden
 * vectors decoded by libjpeg-turbo (tests/golden/jpeg, made with Pillow).
 *   jidctint.c  jpeg_idct_islow: the Loeffler-Ligtenberg-Moschytz inverse DCT with 13-bit constants; pass 1 over columns keeps 2 extra bits,
 *               pass 2 over rows removes 2 + 13 + 3 bits, adds 128 and range-limits through the 1024-entry wrap-around table;
 *   jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample (triangle filters), fullsize_upsample; replication for <= 2 input columns;
 *   jdmainct.c  context rows: the row above the first row and the row below the last REAL row (downsampled_height) are those rows again;
 *   jdcolor.c   ycc_rgb_convert with the tables of build_ycc_rgb_table (16 fractional bits). */
static int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

static void jidct_1d(const int in[8], int out[8])
{
    /* even part */
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    int tmp0 = (in[0] + in[4]) * 8192;
    int tmp1 = (in[0] - in[4]) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    /* odd part */
    tmp0 = in[7], tmp1 = in[5], tmp2 = in[3], tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446;
    tmp1 *= 16819;
    tmp2 *= 25172;
    tmp3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 *= -16069;
    z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = tmp10 + tmp3, out[7] = tmp10 - tmp3;
    out[1] = tmp11 + tmp2, out[6] = tmp11 - tmp2;
    out[2] = tmp12 + tmp1, out[5] = tmp12 - tmp1;
    out[3] = tmp13 + tmp0, out[4] = tmp13 - tmp0;
}

static void jidct_block(const int16_t* coef, const uint16_t* quant, uint8_t* out, int out_pitch, const uint8_t* range_limit /* centred */)
{
    int ws[64];
    for(int c = 0; c < 8; ++c)
    {
        int in[8], o[8];
        for(int r = 0; r < 8; ++r)
            in[r] = (int)coef[8 * r + c] * (int)quant[8 * r + c];
        jidct_1d(in, o);
        for(int r = 0; r < 8; ++r)
            ws[8 * r + c] = jdescale(o[r], 13 - 2);
    }
    for(int r = 0; r < 8; ++r)
    {
        int o[8];
        jidct_1d(ws + 8 * r, o);
        for(int c = 0; c < 8; ++c)
            out[(long long)r * out_pitch + c] = range_limit[jdescale(o[c], 13 + 2 + 3) & 1023];
    }
}

int avo_image_decode_jpeg(uint8_t* dst_rgb, int dst_pitch, int width, int height, const avdm_jpeg_component_t* comps, int n_comps, int hmax, int vmax,
                          int ycc_to_rgb)
{
    if(comps == NULL || (n_comps != 1 && n_comps != 3) || width <= 0 || height <= 0 || hmax < 1 || vmax < 1 || dst_pitch < 3 * width)
        return 1;
    /* the post-IDCT range-limit table (prepare_range_limit_table): indices 0..127 -> 128..255, ..511 -> 255, ..895 -> 0, ..1023 -> 0..127 */
    uint8_t limit[1024];
    for(int i = 0; i < 1024; ++i)
        limit[i] = (uint8_t)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
    uint8_t* planes[3] = {NULL, NULL, NULL};
    uint8_t* full[3] = {NULL, NULL, NULL}; /* components at full resolution (up-sampled), width x height */
    int rc = 0;
    for(int i = 0; i < n_comps && rc == 0; ++i)
    {
        const avdm_jpeg_component_t* c = &comps[i];
        const int he = (c->h_samp > 0 && hmax % c->h_samp == 0) ? hmax / c->h_samp : 0, ve = (c->v_samp > 0 && vmax % c->v_samp == 0) ? vmax / c->v_samp : 0;
        if(c->coef == NULL || c->blocks_w <= 0 || c->blocks_h <= 0 || c->width <= 0 || c->height <= 0 || c->width > 8 * c->blocks_w ||
           c->height > 8 * c->blocks_h || !((he == 1 && ve == 1) || (he == 2 && ve == 1) || (he == 2 && ve == 2)) || c->width * he < width ||
           c->height * ve < height)
        {
            rc = 1;
            break;
        }
        const int pitch = 8 * c->blocks_w;
        planes[i] = (uint8_t*)malloc((size_t)pitch * 8 * c->blocks_h);
        for(int by = 0; by < c->blocks_h; ++by)
            for(int bx = 0; bx < c->blocks_w; ++bx)
                jidct_block(c->coef + ((size_t)by * c->blocks_w + bx) * 64, c->quant, planes[i] + (size_t)8 * by * pitch + 8 * bx, pitch, limit);
        /* up-sampling, an output row pair at a time */
        const int ow = c->width * he;
        full[i] = (uint8_t*)malloc((size_t)ow * c->height * ve);
        for(int inrow = 0; inrow < c->height; ++inrow)
        {
            const uint8_t* in0 = planes[i] + (size_t)inrow * pitch;
            if(he == 1)
                memcpy(full[i] + (size_t)inrow * ow, in0, (size_t)ow);
            else if(ve == 1)
            {
                uint8_t* o = full[i] + (size_t)inrow * ow;
                if(c->width <= 2) /* h2v1_upsample */
                    for(int x = 0; x < c->width; ++x)
                        o[2 * x] = o[2 * x + 1] = in0[x];
                else
                {
                    int x = 0;
                    o[0] = in0[0];
                    o[1] = (uint8_t)((in0[0] * 3 + in0[1] + 2) >> 2);
                    for(x = 1; x < c->width - 1; ++x)
                    {
                        const int v3 = in0[x] * 3;
                        o[2 * x] = (uint8_t)((v3 + in0[x - 1] + 1) >> 2);
                        o[2 * x + 1] = (uint8_t)((v3 + in0[x + 1] + 2) >> 2);
                    }
                    o[2 * x] = (uint8_t)((in0[x] * 3 + in0[x - 1] + 1) >> 2);
                    o[2 * x + 1] = in0[x];
                }
            }
            else
                for(int v = 0; v < 2; ++v)
                {
                    /* the nearer neighbour row: above for the upper output row, below for the lower; the image edges provide themselves */
                    int nrow = v == 0 ? inrow - 1 : inrow + 1;
                    nrow = nrow < 0 ? 0 : (nrow > c->height - 1 ? c->height - 1 : nrow);
                    const uint8_t* in1 = planes[i] + (size_t)nrow * pitch;
                    uint8_t* o = full[i] + (size_t)(2 * inrow + v) * ow;
                    if(c->width <= 2) /* h2v2_upsample */
                    {
                        for(int x = 0; x < c->width; ++x)
                            o[2 * x] = o[2 * x + 1] = in0[x];
                        continue;
                    }
                    int thiscolsum = in0[0] * 3 + in1[0], nextcolsum = in0[1] * 3 + in1[1], lastcolsum;
                    o[0] = (uint8_t)((thiscolsum * 4 + 8) >> 4);
                    o[1] = (uint8_t)((thiscolsum * 3 + nextcolsum + 7) >> 4);
                    lastcolsum = thiscolsum;
                    thiscolsum = nextcolsum;
                    int x;
                    for(x = 1; x < c->width - 1; ++x)
                    {
                        nextcolsum = in0[x + 1] * 3 + in1[x + 1];
                        o[2 * x] = (uint8_t)((thiscolsum * 3 + lastcolsum + 8) >> 4);
                        o[2 * x + 1] = (uint8_t)((thiscolsum * 3 + nextcolsum + 7) >> 4);
                        lastcolsum = thiscolsum;
                        thiscolsum = nextcolsum;
                    }
                    o[2 * x] = (uint8_t)((thiscolsum * 3 + lastcolsum + 8) >> 4);
                    o[2 * x + 1] = (uint8_t)((thiscolsum * 4 + 7) >> 4);
                }
        }
    }
    if(rc == 0)
    {
        /* build_ycc_rgb_table */
        int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        for(int i = 0; i < 256; ++i)
        {
            const int x = i - 128;
            cr_r[i] = (91881 * x + 32768) >> 16;
            cb_b[i] = (116130 * x + 32768) >> 16;
            cr_g[i] = -46802 * x;
            cb_g[i] = -22554 * x + 32768;
        }
        const int ow0 = comps[0].width * (hmax / comps[0].h_samp);
        for(int y = 0; y < height; ++y)
        {
            uint8_t* o = dst_rgb + (long long)y * dst_pitch;
            for(int x = 0; x < width; ++x)
            {
                const int c0 = full[0][(size_t)y * ow0 + x];
                int r = c0, g = c0, b = c0;
                if(n_comps == 3)
                {
                    const int c1 = full[1][(size_t)y * (comps[1].width * (hmax / comps[1].h_samp)) + x];
                    const int c2 = full[2][(size_t)y * (comps[2].width * (hmax / comps[2].h_samp)) + x];
                    if(ycc_to_rgb)
                    {
                        r = c0 + cr_r[c2];
                        g = c0 + ((cb_g[c1] + cr_g[c2]) >> 16);
                        b = c0 + cb_b[c1];
                        r = r < 0 ? 0 : (r > 255 ? 255 : r);
                        g = g < 0 ? 0 : (g > 255 ? 255 : g);
                        b = b < 0 ? 0 : (b > 255 ? 255 : b);
                    }
                    else
                        g = c1, b = c2;
                }
                o[3 * x] = (uint8_t)r, o[3 * x + 1] = (uint8_t)g, o[3 * x + 2] = (uint8_t)b;
            }
        }
    }
    for(int i = 0; i < 3; ++i)
    {
        free(planes[i]);
        free(full[i]);
    }
    return rc;
}

int avo_image_resize(float* dst, int dst_pitch, int dst_w, int dst_h, const float* src, int src_pitch, int src_w, int src_h, int nchannels)
{
    if(dst_w <= 0 || dst_h <= 0 || src_w <= 0 || src_h <= 0 || nchannels < 1 || nchannels > 4 || dst_w > src_w || dst_h > src_h)
        return 1;
    const int xtaps = avo_image_resize_taps(dst_w, src_w, NULL, NULL), ytaps = avo_image_resize_taps(dst_h, src_h, NULL, NULL);
    float* wx = (float*)malloc(sizeof(float) * (size_t)xtaps * dst_w);
    float* wy = (float*)malloc(sizeof(float) * (size_t)ytaps * dst_h);
    int* fx = (int*)malloc(sizeof(int) * dst_w);
    int* fy = (int*)malloc(sizeof(int) * dst_h);
    avo_image_resize_taps(dst_w, src_w, wx, fx);
    avo_image_resize_taps(dst_h, src_h, wy, fy);
#pragma omp parallel for schedule(dynamic, 4)
    for(int y = 0; y < dst_h; ++y)
    {
        const float* yw = wy + (size_t)y * ytaps;
        float* drow = (float*)((char*)dst + (size_t)y * dst_pitch);
        for(int x = 0; x < dst_w; ++x)
        {
            const float* xw = wx + (size_t)x * xtaps;
            float pel[4] = {0.f, 0.f, 0.f, 0.f};
            float totalx = 0.0f;
            for(int i = 0; i < xtaps; ++i)
                totalx += xw[i];
            if(totalx != 0.0f)
                for(int j = 0; j < ytaps; ++j)
                {
                    const float wyj = yw[j];
                    if(wyj == 0.0f)
                        continue;
                    int sy = fy[y] + j;
                    sy = sy < 0 ? 0 : (sy > src_h - 1 ? src_h - 1 : sy);
                    const float* srow = (const float*)((const char*)src + (size_t)sy * src_pitch);
                    for(int i = 0; i < xtaps; ++i)
                    {
                        const float w = wyj * xw[i];
                        if(w != 0.0f)
                        {
                            int sx = fx[x] + i;
                            sx = sx < 0 ? 0 : (sx > src_w - 1 ? src_w - 1 : sx);
                            for(int c = 0; c < nchannels; ++c)
                                pel[c] += w * srow[(size_t)sx * nchannels + c];
                        }
                    }
                }
            for(int c = 0; c < nchannels; ++c)
                drow[(size_t)x * nchannels + c] = pel[c];
        }
    }
    free(wx);
    free(wy);
    free(fx);
    free(fy);
    return 0;
}


/* ---- image ingest: undistortion ------------------------------------------------------------------------------------------------------
 * camera::UndistortImage(imageIn, intrinsicPtr, image_ud, fillcolor) (camera/cameraUndistortImage.hpp:81-139), the call of
 * software/pipeline/main_prepareDenseScene.cpp:71-79.  Everything it uses is in the reference tree:
 *   getDistortedPixel = cam2ima(addDistortion(ima2cam(p)))      camera/IntrinsicScaleOffsetDisto.cpp:80
 *   ima2cam / cam2ima / principal point = offset + size / 2      camera/IntrinsicScaleOffset.cpp:31, 55-66
 *   addDistortion for radialk1 / radialk3 / radialk3pt            camera/DistortionRadial.cpp:18-24, 110-124, 262-277
 *   Image::contains(int y, int x) on the (truncated) doubles      image/Image.hpp:178
 *   Sampler2d<SamplerLinear>::operator()(src, float y, float x)   image/Sampler.hpp:57-75, 377-477 */
static void avo_add_distortion(const avdm_intrinsic_t* c, double* x, double* y)
{
    double coeff = 1.0;
    if(c->distortion_model == AVDM_DISTORTION_RADIALK1)
    {
        const double r2 = (*x) * (*x) + (*y) * (*y);
        coeff = (1. + c->k[0] * r2);
    }
    else if(c->distortion_model == AVDM_DISTORTION_RADIALK3 || c->distortion_model == AVDM_DISTORTION_RADIALK3PT)
    {
        const double r = sqrt((*x) * (*x) + (*y) * (*y));
        const double r2 = r * r;
        const double r4 = r2 * r2;
        const double r6 = r4 * r2;
        coeff = (1. + c->k[0] * r2 + c->k[1] * r4 + c->k[2] * r6);
        if(c->distortion_model == AVDM_DISTORTION_RADIALK3PT)
            coeff = coeff / (1.0 + c->k[0] + c->k[1] + c->k[2]);
    }
    *x = (*x) * coeff;
    *y = (*y) * coeff;
}

int avo_image_undistort(float* dst, int dst_pitch, const float* src, int src_pitch, const avdm_intrinsic_t* cam, const float fill[4])
{
    const int W = cam->width, H = cam->height;
    if(W <= 0 || H <= 0 || cam->distortion_model < AVDM_DISTORTION_NONE || cam->distortion_model > AVDM_DISTORTION_RADIALK3PT)
        return 1;
    if(cam->distortion_model == AVDM_DISTORTION_NONE)
    { /* hasDistortion() false: a direct copy (cameraUndistortImage.hpp:89-93) */
        for(int y = 0; y < H; ++y)
            memcpy((char*)dst + (size_t)y * dst_pitch, (const char*)src + (size_t)y * src_pitch, (size_t)W * 16);
        return 0;
    }
    const double ppx = cam->offset_x + (double)W * 0.5, ppy = cam->offset_y + (double)H * 0.5;
#pragma omp parallel for schedule(dynamic, 8)
    for(int py = 0; py < H; ++py)
        for(int px = 0; px < W; ++px)
        {
            double cx = ((double)px - ppx) / cam->scale_x, cy = ((double)py - ppy) / cam->scale_y;
            avo_add_distortion(cam, &cx, &cy);
            const double dxp = cx * cam->scale_x + ppx, dyp = cy * cam->scale_y + ppy;
            float* o = (float*)((char*)dst + (size_t)py * dst_pitch) + 4 * (size_t)px;
            o[0] = fill[0], o[1] = fill[1], o[2] = fill[2], o[3] = fill[3];
            const int ix = (int)dxp, iy = (int)dyp;
            if(!(0 <= ix && ix < W && 0 <= iy && iy < H))
                continue;
            const float x = (float)dxp, y = (float)dyp;
            const double fxl = floor((double)x), fyl = floor((double)y);
            const double dx = (double)x - fxl, dy = (double)y - fyl;
            const double coefsX[2] = {1.0 - dx, dx}, coefsY[2] = {1.0 - dy, dy};
            const int gridX = (int)fxl, gridY = (int)fyl;
            double res[4] = {0.0, 0.0, 0.0, 0.0}, totalWeight = 0.0;
            for(int i = 0; i < 2; ++i)
            {
                const int iCurrent = gridY + 1 + i - 1;
                if(iCurrent < 0 || iCurrent >= H)
                    continue;
                const float* row = (const float*)((const char*)src + (size_t)iCurrent * src_pitch);
                for(int j = 0; j < 2; ++j)
                {
                    const int jCurrent = gridX + 1 + j - 1;
                    if(jCurrent < 0 || jCurrent >= W)
                        continue;
                    const double w = coefsX[j] * coefsY[i];
                    for(int c = 0; c < 4; ++c)
                        res[c] += (double)row[4 * (size_t)jCurrent + c] * w;
                    totalWeight += w;
                }
            }
            if(totalWeight <= 0.2)
            {
                int row = (int)floor((double)y), col = (int)floor((double)x);
                row = row < 0 ? 0 : (row >= H ? H - 1 : row);
                col = col < 0 ? 0 : (col >= W ? W - 1 : col);
                const float* p = (const float*)((const char*)src + (size_t)row * src_pitch) + 4 * (size_t)col;
                o[0] = p[0], o[1] = p[1], o[2] = p[2], o[3] = p[3];
                continue;
            }
            for(int c = 0; c < 4; ++c)
            {
                if(totalWeight != 1.0)
                    res[c] /= totalWeight;
                o[c] = (float)res[c];
            }
        }
    return 0;
}


/* DeviceCache.cpp:41-134 (fillHostCameraParameters), mvsData/Matrix3x3.hpp:268-287 (inverse) */
static void inv3(const double* m, double* o) /* row-major */
{
    const double m11 = m[0], m12 = m[1], m13 = m[2], m21 = m[3], m22 = m[4], m23 = m[5], m31 = m[6], m32 = m[7], m33 = m[8];
    const double dt = m11 * (m33 * m22 - m32 * m23) - m21 * (m33 * m12 - m32 * m13) + m31 * (m23 * m12 - m22 * m13);
    o[0] = (m33 * m22 - m32 * m23) / dt;
    o[1] = -(m33 * m12 - m32 * m13) / dt;
    o[2] = (m23 * m12 - m22 * m13) / dt;
    o[3] = -(m33 * m21 - m31 * m23) / dt;
    o[4] = (m33 * m11 - m31 * m13) / dt;
    o[5] = -(m23 * m11 - m21 * m13) / dt;
    o[6] = (m32 * m21 - m31 * m22) / dt;
    o[7] = -(m32 * m11 - m31 * m12) / dt;
    o[8] = (m22 * m11 - m21 * m12) / dt;
}
static void mul33(const double* a, const double* b, double* o)
{
    for(int r = 0; r < 3; ++r)
        for(int c = 0; c < 3; ++c)
            o[3 * r + c] = a[3 * r + 0] * b[0 + c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
static void colmajor33(const double* m, float* o)
{
    for(int c = 0; c < 3; ++c)
        for(int r = 0; r < 3; ++r)
            o[3 * c + r] = (float)m[3 * r + c];
}
void avo_camera_fill(avdm_camera_t* out, const double K_[9], const double R[9], const double C[3], int downscale)
{
    double scaleM[9] = {1.0 / (float)downscale, 0, 0, 0, 1.0 / (float)downscale, 0, 0, 0, 1.0};
    double K[9], iK[9], iR[9], iP[9], P[12], RC[3], t[3];
    mul33(scaleM, K_, K);
    inv3(K, iK);
    inv3(R, iR);
    for(int r = 0; r < 3; ++r)
        RC[r] = R[3 * r] * C[0] + R[3 * r + 1] * C[1] + R[3 * r + 2] * C[2];
    for(int r = 0; r < 3; ++r)
        t[r] = 0.0 - RC[r];
    for(int r = 0; r < 3; ++r)
    {
        for(int c = 0; c < 3; ++c)
            P[4 * r + c] = K[3 * r] * R[c] + K[3 * r + 1] * R[3 + c] + K[3 * r + 2] * R[6 + c];
        P[4 * r + 3] = K[3 * r] * t[0] + K[3 * r + 1] * t[1] + K[3 * r + 2] * t[2];
    }
    mul33(iR, iK, iP);
    for(int c = 0; c < 4; ++c)
        for(int r = 0; r < 3; ++r)
            out->P[3 * c + r] = (float)P[4 * r + c];
    colmajor33(iP, out->iP);
    colmajor33(R, out->R);
    colmajor33(iR, out->iR);
    colmajor33(K, out->K);
    colmajor33(iK, out->iK);
    out->C[0] = (float)C[0];
    out->C[1] = (float)C[1];
    out->C[2] = (float)C[2];
    /* DeviceCache.cpp:22-32: host-side normalize uses plain division */
    const f3 ex[3] = {mk3(1.f, 0.f, 0.f), mk3(0.f, 1.f, 0.f), mk3(0.f, 0.f, 1.f)};
    float* dst[3] = {out->XVect, out->YVect, out->ZVect};
    for(int k = 0; k < 3; ++k)
    {
        f3 v = M3x3mulV3(out->iR, ex[k]);
        const float d = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        dst[k][0] = v.x / d;
        dst[k][1] = v.y / d;
        dst[k][2] = v.z / d;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Patch / NCC (cuda/device/Patch.cuh, SimStat.cuh, color.cuh)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { f3 p, n, x, y; float d; } Patch;

/* Patch.cuh:137-145 */
static inline float computePixSize(const avdm_camera_t* cam, f3 p)
{
    const f2 rp = project3DPoint(cam->P, p);
    const f2 rp1 = mk2(rp.x + 1.0f, rp.y + 0.0f);
    const f3 refvect = normalize3(M3x3mulV2(cam->iP, rp1));
    return pointLineDistance3D(p, cam3(cam->C), refvect);
}
/* Patch.cuh:157-163 */
static inline f3 get3DPointForPixelAndFrontoParellePlaneRC(const avdm_camera_t* cam, f2 pix, float fpPlaneDepth)
{
    const f3 planep = add3(cam3(cam->C), mul3(cam3(cam->ZVect), fpPlaneDepth));
    const f3 v = normalize3(M3x3mulV2(cam->iP, pix));
    return linePlaneIntersect(cam3(cam->C), v, planep, cam3(cam->ZVect));
}
/* Patch.cuh:165-170 */
static inline f3 get3DPointForPixelAndDepthFromRC(const avdm_camera_t* cam, f2 pix, float depth)
{
    const f3 rpv = normalize3(M3x3mulV2(cam->iP, pix));
    return add3(cam3(cam->C), mul3(rpv, depth));
}
/* Patch.cuh:111-135 */
static inline void computeRotCSEpip(Patch* ptch, const avdm_camera_t* rc, const avdm_camera_t* tc)
{
    const f3 v1 = normalize3(sub3(cam3(rc->C), ptch->p));
    const f3 v2 = normalize3(sub3(cam3(tc->C), ptch->p));
    ptch->y = normalize3(cross3(v1, v2));
    ptch->n = normalize3(div3(add3(v1, v2), 2.0f));
    ptch->x = normalize3(cross3(ptch->y, ptch->n));
}
/* color.cuh:40-44 (norm3df restated) */
static inline float euclideanDist3(f4 a, f4 b)
{
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return sqrtf(dx * dx + dy * dy + dz * dz);
}
/* color.cuh:167-210 */
static inline float CostYKfromLab(int dx, int dy, f4 c1, f4 c2, float invGammaC, float invGammaP)
{
    float deltaC = euclideanDist3(c1, c2);
    deltaC *= invGammaC;
    float deltaP = sqrtf((float)(dx * dx + dy * dy));
    deltaP *= invGammaP;
    deltaC += deltaP;
    return expf(-deltaC);
}
/* Patch.cuh:250-308 */
static void computeRcTcMipmapLevels(float* out_rc, float* out_tc, float mipmapLevel, const avdm_camera_t* rc, const avdm_camera_t* tc, f2 rp0, f2 tp0,
                                    f3 p0)
{
    const float rcDepth = size3(sub3(cam3(rc->C), p0));
    const float tcDepth = size3(sub3(cam3(tc->C), p0));
    const f2 rp1 = mk2(rp0.x + 1.f, rp0.y + 0.f);
    const f2 tp1 = mk2(tp0.x + 1.f, tp0.y + 0.f);
    const f3 rpv = normalize3(M3x3mulV2(rc->iP, rp1));
    const f3 prp1 = add3(cam3(rc->C), mul3(rpv, rcDepth));
    const f3 tpv = normalize3(M3x3mulV2(tc->iP, tp1));
    const f3 ptp1 = add3(cam3(tc->C), mul3(tpv, tcDepth));
    const float rcDist = size3(sub3(p0, prp1));
    const float tcDist = size3(sub3(p0, ptp1));
    const float distFactor = rcDist / tcDist;
    if(distFactor < 1.f)
    {
        *out_tc = mipmapLevel - log2f(1.f / distFactor);
        if(*out_tc < 0.f)
        {
            *out_rc = mipmapLevel + fabsf(*out_tc);
            *out_tc = 0.f;
        }
    }
    else
    {
        *out_rc = mipmapLevel;
        *out_tc = mipmapLevel + log2f(distFactor);
    }
}

/* The weighted-NCC statistics of SimStat.cuh (E[x^2] - E[x]^2 with sum(w) ~ 1 and x ~ 200) lose 4-5 decimal digits to
 * cancellation in fp32: a faithful fp32 restatement is itself +-2..3 uint8 levels away from the exact value of the same
 * formula, and so is the reference on any GPU (DESIGN.md "NCC conditioning").  avo_set_ncc_precision(1) evaluates the SAME
 * formula with the six running sums in double precision (geometry, texture filtering and weights unchanged): this is the
 * well-defined value the reference's fp32 arithmetic approximates, used by the tolerance-class parity tests. */
static int g_ncc_f64 = 0;
void avo_set_ncc_precision(int f64) { g_ncc_f64 = f64; }

/* The R-side border test `rp.x < wsh + 2` (Patch.cuh:490-493) is applied to the REPROJECTION of a point that lies on the ray
 * of pixel (x, y): in exact arithmetic rp == (x, y).  Where x == wsh + 2 (or W - 1 - (wsh + 2)) the reference's outcome is a
 * coin flip of fp32 rounding, per voxel, and SGM then spreads it along whole rows (DESIGN.md "knife-edge rows").
 * avo_set_exact_rc_pixel(1) evaluates the test (and the centre fetch) on the exact pixel instead: the value the reference's
 * arithmetic approximates.  avo_set_exact_rc_pixel(2) (round 4: what the GPU kernels do, and what oracle.well_posed() selects) keeps the
 * reference's BORDER TEST on the re-projected centre — coin flips included: the kernels evaluate it with the reference's operations on the
 * knife-edge rows, elsewhere the two tests cannot differ — and fetches the centre colour at the exact pixel (the custom patch pattern,
 * whose margin is 2 pixels, keeps the exact pixel for both).  Default 0 = literal restatement. */
static int g_exact_rc_pixel = 0;
void avo_set_exact_rc_pixel(int on) { g_exact_rc_pixel = on; }

/* Patch.cuh:466-572 (compNCCby3DptsYK<TInvertAndFilter>) + SimStat.cuh:72-113,144-153; returns INFINITY when invalid */
static float compNCCby3DptsYK(int invertAndFilter, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rcTex,
                              const avdm_pyramid_t* tcTex, unsigned rcLevelWidth, unsigned rcLevelHeight, unsigned tcLevelWidth,
                              unsigned tcLevelHeight, float mipmapLevel, int wsh, float invGammaC, float invGammaP, int useConsistentScale,
                              const Patch* patch, f2 rcPixel)
{
    const f2 rpLit = project3DPoint(rc->P, patch->p);
    const f2 rpB = g_exact_rc_pixel == 1 ? rcPixel : rpLit; /* the border test: mode 2 keeps the reference's re-projection (its coin flip on the knife-edge rows) */
    const f2 rp = g_exact_rc_pixel ? rcPixel : rpLit;       /* the centre fetch (and the consistent-scale levels) */
    const f2 tp = project3DPoint(tc->P, patch->p);
    const float dd = (float)wsh + 2.0f;
    if((rpB.x < dd) || (rpB.x > (float)(rcLevelWidth - 1) - dd) || (tp.x < dd) || (tp.x > (float)(tcLevelWidth - 1) - dd) || (rpB.y < dd) ||
       (rpB.y > (float)(rcLevelHeight - 1) - dd) || (tp.y < dd) || (tp.y > (float)(tcLevelHeight - 1) - dd))
        return INFINITY;

    const float rcInvLevelWidth = 1.f / (float)rcLevelWidth;
    const float rcInvLevelHeight = 1.f / (float)rcLevelHeight;
    const float tcInvLevelWidth = 1.f / (float)tcLevelWidth;
    const float tcInvLevelHeight = 1.f / (float)tcLevelHeight;

    float rcMipmapLevel = mipmapLevel;
    float tcMipmapLevel = mipmapLevel;
    if(useConsistentScale)
        computeRcTcMipmapLevels(&rcMipmapLevel, &tcMipmapLevel, mipmapLevel, rc, tc, rp, tp, patch->p);

    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
    double dxsum = 0., dysum = 0., dxxsum = 0., dyysum = 0., dxysum = 0., dwsum = 0.; /* AVO_NCC_F64 evaluation, see avo_set_ncc_precision */

    const f4 rcCenterColor = tex2DLod(rcTex, (rp.x + 0.5f) * rcInvLevelWidth, (rp.y + 0.5f) * rcInvLevelHeight, rcMipmapLevel);
    const f4 tcCenterColor = tex2DLod(tcTex, (tp.x + 0.5f) * tcInvLevelWidth, (tp.y + 0.5f) * tcInvLevelHeight, tcMipmapLevel);

    /* color.cuh:12,15 */
    if(rcCenterColor.w < (255.f * 0.9f) || tcCenterColor.w < (255.f * 0.4f))
        return INFINITY;

    for(int yp = -wsh; yp <= wsh; ++yp)
        for(int xp = -wsh; xp <= wsh; ++xp)
        {
            const f3 p = add3(add3(patch->p, mul3(patch->x, (float)(patch->d * (float)xp))), mul3(patch->y, (float)(patch->d * (float)yp)));
            const f2 rpc = project3DPoint(rc->P, p);
            const f2 tpc = project3DPoint(tc->P, p);
            const f4 rcC = tex2DLod(rcTex, (rpc.x + 0.5f) * rcInvLevelWidth, (rpc.y + 0.5f) * rcInvLevelHeight, rcMipmapLevel);
            const f4 tcC = tex2DLod(tcTex, (tpc.x + 0.5f) * tcInvLevelWidth, (tpc.y + 0.5f) * tcInvLevelHeight, tcMipmapLevel);
            const float wr = CostYKfromLab(xp, yp, rcCenterColor, rcC, invGammaC, invGammaP);
            const float wt = CostYKfromLab(xp, yp, tcCenterColor, tcC, invGammaC, invGammaP);
            const float w = wr * wt;
            /* simStat::update(gx, gy, w) SimStat.cuh:144-153 */
            const float gx = rcC.x, gy = tcC.x;
            wsum += w;
            xsum += w * gx;
            ysum += w * gy;
            xxsum += w * gx * gx;
            yysum += w * gy * gy;
            xysum += w * gx * gy;
            if(g_ncc_f64)
            {
                const double dw = (double)wr * (double)wt, dgx = gx, dgy = gy;
                dwsum += dw;
                dxsum += dw * dgx;
                dysum += dw * dgy;
                dxxsum += dw * dgx * dgx;
                dyysum += dw * dgy * dgy;
                dxysum += dw * dgx * dgy;
            }
        }

    float sim;
    if(g_ncc_f64)
    {
        const double varXW = (dxxsum - dxsum * dxsum / dwsum) / dwsum;
        const double varYW = (dyysum - dysum * dysum / dwsum) / dwsum;
        const double varXYW = (dxysum - dxsum * dysum / dwsum) / dwsum;
        const double rawSim = varXYW / sqrt(varXW * varYW);
        sim = isfinite(rawSim) ? (float)-rawSim : 1.0f;
    }
    else
    {
        /* simStat::computeWSim SimStat.cuh:72-113 */
        const float varXW = (xxsum - xsum * xsum / wsum) / wsum;
        const float varYW = (yysum - ysum * ysum / wsum) / wsum;
        const float varXYW = (xysum - xsum * ysum / wsum) / wsum;
        const float rawSim = varXYW / sqrtf(varXW * varYW);
        sim = isfinite(rawSim) ? -rawSim : 1.0f;
    }
    if(invertAndFilter)
        return sigmoidf_(0.0f, 1.0f, 0.7f, -0.7f, sim);
    return sim;
}

/* ------------------------------------------------------------------------------------------------
 * Custom patch pattern (cuda/host/patchPattern.cpp:18-251, cuda/device/Patch.cuh:598-773, DevicePatchPattern.hpp)
 * ---------------------------------------------------------------------------------------------- */
static avdm_patch_pattern_t g_patchPattern; /* constantPatchPattern_d */
static int g_patchPatternSet = 0;

/* patchPattern.cpp:18-251.  In the non-grouped form the reference uses subpart.nbCoordinates before assigning it (:196-201,
 * uninitialised pinned memory); the parameter's value is used here, as in the library. */
int avo_build_custom_patch_pattern(int n_subparts, const avdm_patch_subpart_params_t* subparts, int group, avdm_patch_pattern_t* out)
{
    if(n_subparts <= 0 || subparts == NULL)
        return 1;
    /* std::map<int, int> nbCoordsPerSubparts: keys in ascending order */
    int keys[64], counts[64], nkeys = 0;
    if(n_subparts > 64)
        return 1;
    for(int i = 0; i < n_subparts; ++i)
    {
        const avdm_patch_subpart_params_t* sp = &subparts[i];
        if(sp->radius <= 0.f)
            return 1;
        if(sp->isCircle && sp->nbCoordinates <= 0)
            return 1;
        const int key = group ? sp->level : i;
        int k = 0;
        while(k < nkeys && keys[k] != key)
            ++k;
        if(k < nkeys && group && !sp->isCircle)
            return 1; /* Cannot group more than one full patch pattern subpart */
        if(k == nkeys)
        {
            /* sorted insert */
            int pos = nkeys;
            while(pos > 0 && keys[pos - 1] > key)
            {
                keys[pos] = keys[pos - 1];
                counts[pos] = counts[pos - 1];
                --pos;
            }
            keys[pos] = key;
            counts[pos] = 0;
            ++nkeys;
            k = pos;
        }
        counts[k] += sp->isCircle ? sp->nbCoordinates : 0;
    }
    int maxCoords = 0;
    for(int k = 0; k < nkeys; ++k)
        maxCoords = counts[k] > maxCoords ? counts[k] : maxCoords;
    if(nkeys > AVDM_PATCH_MAX_SUBPARTS || maxCoords > AVDM_PATCH_MAX_COORDS_PER_SUBPART)
        return 1;

    avdm_patch_pattern_t pp;
    memset(&pp, 0, sizeof(pp));
    pp.nbSubparts = nkeys;
    for(int i = 0; i < n_subparts; ++i)
    {
        const avdm_patch_subpart_params_t* sp = &subparts[i];
        int k = 0;
        const int key = group ? sp->level : i;
        while(keys[k] != key)
            ++k;
        avdm_patch_pattern_subpart_t* part = &pp.subparts[k];
        if(sp->isCircle)
        {
            const float angleDifference = (float)((3.14159265358979323846 * 2.f) / sp->nbCoordinates); /* (M_PI * 2.f) / n in double, then float */
            const int first = group ? part->nbCoordinates : 0;
            for(int j = 0; j < sp->nbCoordinates; ++j)
            {
                const float radians = angleDifference * (float)j;
                part->coordinates[first + j][0] = cosf(radians) * sp->radius;
                part->coordinates[first + j][1] = sinf(radians) * sp->radius;
            }
            const int w = (int)(sp->radius + powf(2.f, (float)sp->level - 1.f));
            part->wsh = group ? (part->wsh > w ? part->wsh : w) : w;
            part->nbCoordinates = group ? part->nbCoordinates + sp->nbCoordinates : sp->nbCoordinates;
        }
        else
        {
            const int w = (int)sp->radius;
            part->wsh = group ? (part->wsh > w ? part->wsh : w) : w;
            if(!group)
                part->nbCoordinates = 0;
        }
        part->level = (float)sp->level;
        part->downscale = powf(2.f, part->level);
        part->weight = sp->weight;
        part->isCircle = sp->isCircle ? 1 : 0;
    }
    g_patchPattern = pp;
    g_patchPatternSet = 1;
    if(out != NULL)
        *out = pp;
    return 0;
}

/* color.cuh:226-232 */
static inline float CostYKfromLab3(f4 c1, f4 c2, float invGammaC) { return expf(-(euclideanDist3(c1, c2) * invGammaC)); }

/* Patch.cuh:598-773 (compNCCby3DptsYK_customPatchPattern<TInvertAndFilter>) */
static float compNCCby3DptsYK_customPatchPattern(int invertAndFilter, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rcTex,
                                                 const avdm_pyramid_t* tcTex, unsigned rcLevelWidth, unsigned rcLevelHeight, unsigned tcLevelWidth,
                                                 unsigned tcLevelHeight, float mipmapLevel, float invGammaC, float invGammaP, int useConsistentScale,
                                                 const Patch* patch, f2 rcPixel)
{
    const f2 rp = g_exact_rc_pixel ? rcPixel : project3DPoint(rc->P, patch->p);
    const f2 tp = project3DPoint(tc->P, patch->p);
    const float dd = 2.f;
    if((rp.x < dd) || (rp.x > (float)(rcLevelWidth - 1) - dd) || (tp.x < dd) || (tp.x > (float)(tcLevelWidth - 1) - dd) || (rp.y < dd) ||
       (rp.y > (float)(rcLevelHeight - 1) - dd) || (tp.y < dd) || (tp.y > (float)(tcLevelHeight - 1) - dd))
        return INFINITY;
    const float rcInvLevelWidth = 1.f / (float)rcLevelWidth, rcInvLevelHeight = 1.f / (float)rcLevelHeight;
    const float tcInvLevelWidth = 1.f / (float)tcLevelWidth, tcInvLevelHeight = 1.f / (float)tcLevelHeight;
    const float rcAlpha = tex2DLod(rcTex, (rp.x + 0.5f) * rcInvLevelWidth, (rp.y + 0.5f) * rcInvLevelHeight, mipmapLevel).w;
    const float tcAlpha = tex2DLod(tcTex, (tp.x + 0.5f) * tcInvLevelWidth, (tp.y + 0.5f) * tcInvLevelHeight, mipmapLevel).w;
    if(rcAlpha < (255.f * 0.9f) || tcAlpha < (255.f * 0.4f))
        return INFINITY;
    float rcMipmapLevel = mipmapLevel, tcMipmapLevel = mipmapLevel;
    if(useConsistentScale)
        computeRcTcMipmapLevels(&rcMipmapLevel, &tcMipmapLevel, mipmapLevel, rc, tc, rp, tp, patch->p);

    float fsim = 0.f, wsum = 0.f;
    for(int s = 0; s < g_patchPattern.nbSubparts; ++s)
    {
        const avdm_patch_pattern_subpart_t* subpart = &g_patchPattern.subparts[s];
        const f4 rcCenterColor = tex2DLod(rcTex, (rp.x + 0.5f) * rcInvLevelWidth, (rp.y + 0.5f) * rcInvLevelHeight, rcMipmapLevel + subpart->level);
        const f4 tcCenterColor = tex2DLod(tcTex, (tp.x + 0.5f) * tcInvLevelWidth, (tp.y + 0.5f) * tcInvLevelHeight, tcMipmapLevel + subpart->level);
        /* simStat in double when the well-conditioned evaluation is on (see avo_set_ncc_precision), else fp32 like SimStat.cuh */
        double xsum = 0., ysum = 0., xxsum = 0., yysum = 0., xysum = 0., sw = 0.;
        float fxsum = 0.f, fysum = 0.f, fxxsum = 0.f, fyysum = 0.f, fxysum = 0.f, fsw = 0.f;
        const int side = 2 * subpart->wsh + 1;
        const int n = subpart->isCircle ? subpart->nbCoordinates : side * side;
        for(int c = 0; c < n; ++c)
        {
            float cx, cy;
            int xp = 0, yp = 0;
            if(subpart->isCircle)
            {
                cx = subpart->coordinates[c][0];
                cy = subpart->coordinates[c][1];
            }
            else
            {
                yp = c / side - subpart->wsh;
                xp = c % side - subpart->wsh;
                cx = (float)xp * subpart->downscale;
                cy = (float)yp * subpart->downscale;
            }
            const f3 p = add3(add3(patch->p, mul3(patch->x, (float)(patch->d * cx))), mul3(patch->y, (float)(patch->d * cy)));
            const f2 rpc = project3DPoint(rc->P, p);
            const f2 tpc = project3DPoint(tc->P, p);
            const f4 rcC = tex2DLod(rcTex, (rpc.x + 0.5f) * rcInvLevelWidth, (rpc.y + 0.5f) * rcInvLevelHeight, rcMipmapLevel + subpart->level);
            const f4 tcC = tex2DLod(tcTex, (tpc.x + 0.5f) * tcInvLevelWidth, (tpc.y + 0.5f) * tcInvLevelHeight, tcMipmapLevel + subpart->level);
            float wr, wt;
            if(subpart->isCircle)
            {
                wr = CostYKfromLab3(rcCenterColor, rcC, invGammaC);
                wt = CostYKfromLab3(tcCenterColor, tcC, invGammaC);
            }
            else
            {
                wr = CostYKfromLab(xp, yp, rcCenterColor, rcC, invGammaC, invGammaP);
                wt = CostYKfromLab(xp, yp, tcCenterColor, tcC, invGammaC, invGammaP);
            }
            const float w = wr * wt, gx = rcC.x, gy = tcC.x;
            fsw += w;
            fxsum += w * gx;
            fysum += w * gy;
            fxxsum += w * gx * gx;
            fyysum += w * gy * gy;
            fxysum += w * gx * gy;
            const double dw = (double)wr * (double)wt, dgx = gx, dgy = gy;
            sw += dw;
            xsum += dw * dgx;
            ysum += dw * dgy;
            xxsum += dw * dgx * dgx;
            yysum += dw * dgy * dgy;
            xysum += dw * dgx * dgy;
        }
        float fsimSubpart;
        if(g_ncc_f64)
        {
            const double varXW = (xxsum - xsum * xsum / sw) / sw, varYW = (yysum - ysum * ysum / sw) / sw, varXYW = (xysum - xsum * ysum / sw) / sw;
            const double rawSim = varXYW / sqrt(varXW * varYW);
            fsimSubpart = isfinite(rawSim) ? (float)-rawSim : 1.0f;
        }
        else
        {
            const float varXW = (fxxsum - fxsum * fxsum / fsw) / fsw, varYW = (fyysum - fysum * fysum / fsw) / fsw;
            const float varXYW = (fxysum - fxsum * fysum / fsw) / fsw;
            const float rawSim = varXYW / sqrtf(varXW * varYW);
            fsimSubpart = isfinite(rawSim) ? -rawSim : 1.0f;
        }
        if(fsimSubpart < 0.f)
        {
            if(invertAndFilter)
                fsim += sigmoidf_(0.0f, 1.0f, 0.7f, -0.7f, fsimSubpart) * subpart->weight;
            else
                fsim += fsimSubpart * subpart->weight;
            wsum += subpart->weight;
        }
    }
    if(wsum == 0.f)
        return INFINITY;
    if(invertAndFilter)
        return fsim; /* "for now, we do not average" */
    return fsim / wsum;
}

/* ------------------------------------------------------------------------------------------------
 * Similarity volume kernels (planeSweeping/deviceSimilarityVolumeKernels.cuh)
 * ---------------------------------------------------------------------------------------------- */
#define VOL8(base, x, y, z) ((base) + (long long)(y) * pitch_y + (long long)(x) * pitch_x + (z))

/* kernels.cuh:48-62 */
void avo_volume_initialize_u8(uint8_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, uint8_t value)
{
#pragma omp parallel for schedule(static)
    for(int y = 0; y < dimY; ++y)
        for(int x = 0; x < dimX; ++x)
            memset(VOL8(vol, x, y, 0), value, (size_t)dimZ);
}
void avo_volume_initialize_f16(uint16_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, float value)
{
    const uint16_t h = f2h(value);
#pragma omp parallel for schedule(static)
    for(int y = 0; y < dimY; ++y)
        for(int x = 0; x < dimX; ++x)
        {
            uint16_t* p = (uint16_t*)((char*)vol + (long long)y * pitch_y + (long long)x * pitch_x);
            for(int z = 0; z < dimZ; ++z)
                p[z] = h;
        }
}
/* kernels.cuh:64-85 */
void avo_volume_add_f16(uint16_t* inout, const uint16_t* in, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ)
{
#pragma omp parallel for schedule(static)
    for(int y = 0; y < dimY; ++y)
        for(int x = 0; x < dimX; ++x)
        {
            uint16_t* p = (uint16_t*)((char*)inout + (long long)y * pitch_y + (long long)x * pitch_x);
            const uint16_t* q = (const uint16_t*)((const char*)in + (long long)y * pitch_y + (long long)x * pitch_x);
            for(int z = 0; z < dimZ; ++z)
                p[z] = f2h(h2f(p[z]) + h2f(q[z]));
        }
}
/* kernels.cuh:87-107 */
void avo_volume_update_uninitialized(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ)
{
#pragma omp parallel for schedule(static)
    for(int y = 0; y < dimY; ++y)
        for(int x = 0; x < dimX; ++x)
            for(int z = 0; z < dimZ; ++z)
                if(*VOL8(second, x, y, z) >= 255)
                    *VOL8(second, x, y, z) = *VOL8(best, x, y, z);
}

/* kernels.cuh:109-233 (volume_computeSimilarity_kernel), launch constants deviceSimilarityVolume.cu:155-206 */
void avo_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                                   const avdm_camera_t* tc, const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr,
                                   const avdm_sgm_params_t* sp, avdm_range_t depthRange, avdm_roi_t roi)
{
    const float rcMipmapLevel = pyr_level(rcPyr, sp->scale);
    const unsigned rcW = pyr_dim_w(rcPyr, sp->scale), rcH = pyr_dim_h(rcPyr, sp->scale);
    const unsigned tcW = pyr_dim_w(tcPyr, sp->scale), tcH = pyr_dim_h(tcPyr, sp->scale);
    const float invGammaC = 1.f / (float)sp->gammaC, invGammaP = 1.f / (float)sp->gammaP;
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    const int nz = (int)(depthRange.end - depthRange.begin);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for(int vy = 0; vy < roiH; ++vy)
        for(int vx = 0; vx < roiW; ++vx)
            for(int rz = 0; rz < nz; ++rz)
            {
                const int vz = (int)depthRange.begin + rz;
                const float x = (float)(roi.x.begin + vx) * (float)sp->stepXY;
                const float y = (float)(roi.y.begin + vy) * (float)sp->stepXY;
                const float depthPlane = depths[vz];
                Patch patch;
                /* volume_computePatch kernels.cuh:26-35 */
                patch.p = get3DPointForPixelAndFrontoParellePlaneRC(rc, mk2(x, y), depthPlane);
                patch.d = computePixSize(rc, patch.p);
                computeRotCSEpip(&patch, rc, tc);
                /* kernels.cuh:163-192: custom patch pattern or the wsh square */
                float fsim = (sp->useCustomPatchPattern && g_patchPatternSet)
                                 ? compNCCby3DptsYK_customPatchPattern(0, rc, tc, rcPyr, tcPyr, rcW, rcH, tcW, tcH, rcMipmapLevel, invGammaC, invGammaP,
                                                                       sp->useConsistentScale, &patch, mk2(x, y))
                                 : compNCCby3DptsYK(0, rc, tc, rcPyr, tcPyr, rcW, rcH, tcW, tcH, rcMipmapLevel, sp->wsh, invGammaC, invGammaP,
                                                    sp->useConsistentScale, &patch, mk2(x, y));
                if(fsim == INFINITY)
                    fsim = 255.0f;
                else
                {
                    fsim = (fsim - (-1.0f)) * (1.0f / (1.0f - (-1.0f)));
                    fsim = fminf(1.0f, fmaxf(0.0f, fsim));
                    fsim *= 254.0f;
                }
                uint8_t* f1 = VOL8(best, vx, vy, vz);
                uint8_t* f2_ = VOL8(second, vx, vy, vz);
                if(fsim < (float)*f1)
                {
                    *f2_ = *f1;
                    *f1 = (uint8_t)fsim;
                }
                else if(fsim < (float)*f2_)
                    *f2_ = (uint8_t)fsim;
            }
}

/* kernels.cuh:17-24 */
static inline f3 move3DPointByRcPixSize(f3 p, const avdm_camera_t* rc, float rcPixSize)
{
    const f3 rpv = normalize3(sub3(p, cam3(rc->C)));
    return add3(p, mul3(rpv, rcPixSize));
}

/* kernels.cuh:235-391 (volume_refineSimilarity_kernel), launch constants deviceSimilarityVolume.cu:208-259 */
void avo_volume_refine_similarity(uint16_t* vol, long long pitch_y, int pitch_x, int volDimZ, const float* sgmDepthPixSize, int map_pitch,
                                  const float* sgmNormal, int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc,
                                  const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, const avdm_refine_params_t* rp, avdm_range_t depthRange,
                                  avdm_roi_t roi)
{
    const float rcMipmapLevel = pyr_level(rcPyr, rp->scale);
    const unsigned rcW = pyr_dim_w(rcPyr, rp->scale), rcH = pyr_dim_h(rcPyr, rp->scale);
    const unsigned tcW = pyr_dim_w(tcPyr, rp->scale), tcH = pyr_dim_h(tcPyr, rp->scale);
    const float invGammaC = 1.f / (float)rp->gammaC, invGammaP = 1.f / (float)rp->gammaP;
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    const int nz = (int)(depthRange.end - depthRange.begin);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for(int vy = 0; vy < roiH; ++vy)
        for(int vx = 0; vx < roiW; ++vx)
        {
            const float* dps = (const float*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + 2 * vx;
            if(dps[0] <= 0.0f)
                continue;
            for(int rz = 0; rz < nz; ++rz)
            {
                const int vz = (int)depthRange.begin + rz;
                const float x = (float)(roi.x.begin + vx) * (float)rp->stepXY;
                const float y = (float)(roi.y.begin + vy) * (float)rp->stepXY;
                f3 p = get3DPointForPixelAndDepthFromRC(rc, mk2(x, y), dps[0]);
                const int relativeDepthIndexOffset = vz - ((volDimZ - 1) / 2);
                if(relativeDepthIndexOffset != 0)
                {
                    const float pixSizeOffset = (float)relativeDepthIndexOffset * dps[1];
                    p = move3DPointByRcPixSize(p, rc, pixSizeOffset);
                }
                Patch patch;
                patch.p = p;
                patch.d = computePixSize(rc, p);
                {
                    const f3 v1 = normalize3(sub3(cam3(rc->C), patch.p));
                    const f3 v2 = normalize3(sub3(cam3(tc->C), patch.p));
                    patch.y = normalize3(cross3(v1, v2));
                    if(sgmNormal != NULL)
                    {
                        const float* n = (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx;
                        patch.n = mk3(n[0], n[1], n[2]);
                    }
                    else
                        patch.n = normalize3(div3(add3(v1, v2), 2.0f));
                    patch.x = normalize3(cross3(patch.y, patch.n));
                }
                /* kernels.cuh:339-368 */
                const float fsim = (rp->useCustomPatchPattern && g_patchPatternSet)
                                       ? compNCCby3DptsYK_customPatchPattern(1, rc, tc, rcPyr, tcPyr, rcW, rcH, tcW, tcH, rcMipmapLevel, invGammaC,
                                                                             invGammaP, rp->useConsistentScale, &patch, mk2(x, y))
                                       : compNCCby3DptsYK(1, rc, tc, rcPyr, tcPyr, rcW, rcH, tcW, tcH, rcMipmapLevel, rp->wsh, invGammaC, invGammaP,
                                                          rp->useConsistentScale, &patch, mk2(x, y));
                if(fsim == INFINITY)
                    continue;
                uint16_t* out = (uint16_t*)((char*)vol + (long long)vy * pitch_y + (long long)vx * pitch_x) + vz;
                *out = f2h(h2f(*out) + fsim);
            }
        }
}

/* kernels.cuh:37-46 */
static inline float depthPlaneToDepth(const avdm_camera_t* cam, float fpPlaneDepth, f2 pix)
{
    const f3 planep = add3(cam3(cam->C), mul3(cam3(cam->ZVect), fpPlaneDepth));
    const f3 v = normalize3(M3x3mulV2(cam->iP, pix));
    const f3 p = linePlaneIntersect(cam3(cam->C), v, planep, cam3(cam->ZVect));
    return size3(sub3(cam3(cam->C), p));
}

/* kernels.cuh:393-512 (volume_retrieveBestDepth_kernel), launch constants deviceSimilarityVolume.cu:427-467 */
void avo_volume_retrieve_best_depth(float* outDT, int dt_pitch, float* outDS, int ds_pitch, const float* depths, const uint8_t* vol,
                                    long long pitch_y, int pitch_x, int volDimZ, const avdm_camera_t* rc1, const avdm_sgm_params_t* sp,
                                    avdm_range_t depthRange, avdm_roi_t roi)
{
    const int scaleStep = sp->scale * sp->stepXY;
    const float thicknessMultFactor = 1.f + (float)sp->depthThicknessInflate;
    const float maxSimilarity = (float)sp->maxSimilarity * 254.f;
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
#pragma omp parallel for schedule(static)
    for(int vy = 0; vy < roiH; ++vy)
        for(int vx = 0; vx < roiW; ++vx)
        {
            const f2 pix = mk2((float)((roi.x.begin + vx) * scaleStep), (float)((roi.y.begin + vy) * scaleStep));
            float* dt = (float*)((char*)outDT + (long long)vy * dt_pitch) + 2 * vx;
            float* ds = outDS ? (float*)((char*)outDS + (long long)vy * ds_pitch) + 2 * vx : NULL;
            float bestSim = 255.f;
            int bestZIdx = -1;
            for(int vz = (int)depthRange.begin; vz < (int)depthRange.end; ++vz)
            {
                const float simAtZ = (float)*VOL8(vol, vx, vy, vz);
                if(simAtZ < bestSim)
                {
                    bestSim = simAtZ;
                    bestZIdx = vz;
                }
            }
            if((bestZIdx == -1) || (bestSim > maxSimilarity))
            {
                dt[0] = -1.f;
                dt[1] = -1.f;
                if(ds)
                {
                    ds[0] = -1.f;
                    ds[1] = 1.f;
                }
                continue;
            }
            const int m1 = bestZIdx - 1 > 0 ? bestZIdx - 1 : 0;
            const int p1 = bestZIdx + 1 < volDimZ - 1 ? bestZIdx + 1 : volDimZ - 1;
            const float bestDepth = depthPlaneToDepth(rc1, depths[bestZIdx], pix);
            const float bestDepth_m1 = depthPlaneToDepth(rc1, depths[m1], pix);
            const float bestDepth_p1 = depthPlaneToDepth(rc1, depths[p1], pix);
            const float out_bestDepth = bestDepth;
            const float out_bestSim = (bestSim / 255.0f) * 2.0f - 1.0f;
            const float a = bestDepth_p1 - out_bestDepth, b = out_bestDepth - bestDepth_m1;
            const float thick = (a > b ? a : b) * thicknessMultFactor;
            dt[0] = out_bestDepth;
            dt[1] = thick;
            if(ds)
            {
                ds[0] = out_bestDepth;
                ds[1] = out_bestSim;
            }
        }
}

/* kernels.cuh:515-594 (volume_refineBestDepth_kernel), launch constants deviceSimilarityVolume.cu:469-502 */
void avo_volume_refine_best_depth(float* out, int out_pitch, const float* sgmDepthPixSize, int map_pitch, const uint16_t* vol, long long pitch_y,
                                  int pitch_x, int volDimZ, const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    const int samplesPerPixSize = rp->nbSubsamples;
    const int halfNbSamples = rp->nbSubsamples * rp->halfNbDepths;
    const int halfNbDepths = rp->halfNbDepths;
    const float twoTimesSigmaPowerTwo = (float)(2.0 * rp->sigma * rp->sigma);
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
#pragma omp parallel for schedule(dynamic, 4)
    for(int vy = 0; vy < roiH; ++vy)
        for(int vx = 0; vx < roiW; ++vx)
        {
            const float* dps = (const float*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + 2 * vx;
            float* o = (float*)((char*)out + (long long)vy * out_pitch) + 2 * vx;
            if(dps[0] <= 0.0f)
            {
                o[0] = dps[0];
                o[1] = 1.0f;
                continue;
            }
            const uint16_t* v = (const uint16_t*)((const char*)vol + (long long)vy * pitch_y + (long long)vx * pitch_x);
            float bestSampleSim = 0.f;
            int bestSampleOffsetIndex = 0;
            for(int sample = -halfNbSamples; sample <= halfNbSamples; ++sample)
            {
                float sampleSim = 0.f;
                for(int vz = 0; vz < volDimZ; ++vz)
                {
                    const int rz = (vz - halfNbDepths);
                    const int zs = rz * samplesPerPixSize;
                    const float invSimSum = h2f(v[vz]);
                    const float simSum = -invSimSum;
                    sampleSim += simSum * expf(-(float)((zs - sample) * (zs - sample)) / twoTimesSigmaPowerTwo);
                }
                if(sampleSim < bestSampleSim)
                {
                    bestSampleOffsetIndex = sample;
                    bestSampleSim = sampleSim;
                }
            }
            const float sampleSize = dps[1] / (float)samplesPerPixSize;
            const float sampleSizeOffset = (float)bestSampleOffsetIndex * sampleSize;
            o[0] = dps[0] + sampleSizeOffset;
            o[1] = bestSampleSim;
        }
}

/* ------------------------------------------------------------------------------------------------
 * SGM path aggregation: deviceSimilarityVolume.cu:262-425 (cuda_volumeAggregatePath, cuda_volumeOptimize),
 * kernels.cuh:596-744 (initVolumeYSlice, getVolumeXZSlice, computeBestZInSlice, agregateCostVolumeAtXinSlices)
 * ---------------------------------------------------------------------------------------------- */
static void aggregate_path(uint8_t* out, const uint8_t* in, long long pitch_y, int pitch_x, const int volDim[3], const int axisT[3],
                           const avdm_pyramid_t* rcPyr, unsigned rcW, unsigned rcH, float rcMipmapLevel, const avdm_sgm_params_t* sp,
                           int filteringIndex, int invY, avdm_roi_t roi, uint32_t* sliceA, uint32_t* sliceB, uint32_t* bestAcc)
{
    const int volDimX = volDim[axisT[0]];
    const int volDimY = volDim[axisT[1]];
    const int volDimZ = volDim[axisT[2]];
    const int ySign = invY ? -1 : 1;
    const float step = (float)sp->stepXY;
    const float P1 = (float)sp->p1;
    const float _P2 = (float)sp->p2Weighting;
    uint32_t* cur = sliceA;  /* xzSliceForY   */
    uint32_t* prev = sliceB; /* xzSliceForYm1 */

    /* getVolumeXZSlice(y = 0) -> prev; initVolumeYSlice(out, y = 0, 255) */
    for(int x = 0; x < volDimX; ++x)
        for(int z = 0; z < volDimZ; ++z)
        {
            int v[3];
            v[axisT[0]] = x;
            v[axisT[1]] = 0;
            v[axisT[2]] = z;
            prev[(size_t)x * volDimZ + z] = (uint32_t)*VOL8(in, v[0], v[1], v[2]);
            *VOL8(out, v[0], v[1], v[2]) = 255;
        }

    for(int iy = 1; iy < volDimY; ++iy)
    {
        const int y = invY ? volDimY - 1 - iy : iy;
#pragma omp parallel for schedule(static)
        for(int x = 0; x < volDimX; ++x)
        {
            /* computeBestZInSlice kernels.cuh:635-650 */
            uint32_t bestCst = prev[(size_t)x * volDimZ];
            for(int z = 1; z < volDimZ; ++z)
            {
                const uint32_t cst = prev[(size_t)x * volDimZ + z];
                bestCst = cst < bestCst ? cst : bestCst;
            }
            bestAcc[x] = bestCst;

            int v[3];
            v[axisT[0]] = x;
            v[axisT[1]] = y;

            /* P2 depends on (x, y) only — hoisted out of the z loop (kernels.cuh:696-720) */
            float P2 = 0;
            if(_P2 < 0)
                P2 = fabsf(_P2);
            else
            {
                const int beginX = sp->strictRoiQuirk ? ((axisT[0] == 0) ? (int)roi.x.begin : (int)roi.y.begin) : (int)roi.x.begin;
                const int beginY = sp->strictRoiQuirk ? ((axisT[0] == 0) ? (int)roi.y.begin : (int)roi.x.begin) : (int)roi.y.begin;
                const int imX0 = (int)((float)(beginX + v[0]) * step);
                const int imY0 = (int)((float)(beginY + v[1]) * step);
                const int imX1 = (int)((float)imX0 - (float)ySign * step * (float)(axisT[1] == 0));
                const int imY1 = (int)((float)imY0 - (float)ySign * step * (float)(axisT[1] == 1));
                const f4 gcr0 = tex2DLod(rcPyr, ((float)imX0 + 0.5f) / (float)rcW, ((float)imY0 + 0.5f) / (float)rcH, rcMipmapLevel);
                const f4 gcr1 = tex2DLod(rcPyr, ((float)imX1 + 0.5f) / (float)rcW, ((float)imY1 + 0.5f) / (float)rcH, rcMipmapLevel);
                const float deltaC = euclideanDist3(gcr0, gcr1);
                /* sigmoid(80, 255, 80, _P2, deltaC) with the specified exp */
                P2 = 80.f + (255.f - 80.f) * (1.0f / (1.0f + avo_exp_p2(10.0f * ((deltaC - _P2) / 80.f))));
            }

            for(int z = 0; z < volDimZ; ++z)
            {
                v[axisT[2]] = z;
                /* getVolumeXZSlice(y) -> cur */
                const uint32_t sim_xz = (uint32_t)*VOL8(in, v[0], v[1], v[2]);
                float pathCost = 255.0f;
                if((z >= 1) && (z < volDim[2] - 1))
                {
                    const uint32_t bestCostInColM1 = bestAcc[x];
                    const uint32_t pathCostMDM1 = prev[(size_t)x * volDimZ + z - 1];
                    const uint32_t pathCostMD = prev[(size_t)x * volDimZ + z];
                    const uint32_t pathCostMDP1 = prev[(size_t)x * volDimZ + z + 1];
                    const float minCost =
                      fminf(fminf(fminf((float)pathCostMD, (float)pathCostMDM1 + P1), (float)pathCostMDP1 + P1), (float)bestCostInColM1 + P2);
                    pathCost = (float)sim_xz + minCost - (float)bestCostInColM1;
                }
                cur[(size_t)x * volDimZ + z] = (uint32_t)pathCost;
                pathCost = fminf(255.0f, fmaxf(0.0f, pathCost));
                uint8_t* o = VOL8(out, v[0], v[1], v[2]);
                const float val = ((float)*o * (float)filteringIndex + pathCost) / (float)(filteringIndex + 1);
                *o = (uint8_t)val;
            }
        }
        uint32_t* t = prev;
        prev = cur;
        cur = t;
    }
}

void avo_volume_optimize(uint8_t* out, const uint8_t* in, long long pitch_y, int pitch_x, int dimX, int dimY, const avdm_pyramid_t* rcPyr,
                         const avdm_sgm_params_t* sp, int lastDepthIndex, avdm_roi_t roi)
{
    const float rcMipmapLevel = pyr_level(rcPyr, sp->scale);
    const unsigned rcW = pyr_dim_w(rcPyr, sp->scale), rcH = pyr_dim_h(rcPyr, sp->scale);
    const int volDim[3] = {dimX, dimY, lastDepthIndex};
    const int maxSide = dimX > dimY ? dimX : dimY;
    uint32_t* sliceA = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)maxSide * lastDepthIndex);
    uint32_t* sliceB = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)maxSide * lastDepthIndex);
    uint32_t* bestAcc = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)maxSide);
    int npaths = 0;
    for(const char* a = sp->filteringAxes; *a; ++a)
    {
        int axisT[3];
        if(*a == 'X') { axisT[0] = 1; axisT[1] = 0; axisT[2] = 2; }
        else if(*a == 'Y') { axisT[0] = 0; axisT[1] = 1; axisT[2] = 2; }
        else continue;
        aggregate_path(out, in, pitch_y, pitch_x, volDim, axisT, rcPyr, rcW, rcH, rcMipmapLevel, sp, npaths++, 0, roi, sliceA, sliceB, bestAcc);
        aggregate_path(out, in, pitch_y, pitch_x, volDim, axisT, rcPyr, rcW, rcH, rcMipmapLevel, sp, npaths++, 1, roi, sliceA, sliceB, bestAcc);
    }
    free(sliceA);
    free(sliceB);
    free(bestAcc);
}

/* ------------------------------------------------------------------------------------------------
 * Depth/sim map kernels (planeSweeping/deviceDepthSimilarityMapKernels.cuh, deviceDepthSimilarityMap.cu)
 * ---------------------------------------------------------------------------------------------- */
#define MAP2(base, pitch, x, y) ((float*)((char*)(base) + (long long)(y) * (pitch)) + 2 * (long long)(x))
#define MAP1(base, pitch, x, y) ((float*)((char*)(base) + (long long)(y) * (pitch)) + (long long)(x))
#define MAP3(base, pitch, x, y) ((float*)((char*)(base) + (long long)(y) * (pitch)) + 3 * (long long)(x))

/* mapKernels.cuh:110-126 */
void avo_depth_sim_map_copy_depth_only(float* out, int out_pitch, const float* in, int in_pitch, int width, int height, float defaultSim)
{
    for(int y = 0; y < height; ++y)
        for(int x = 0; x < width; ++x)
        {
            MAP2(out, out_pitch, x, y)[0] = MAP2(in, in_pitch, x, y)[0];
            MAP2(out, out_pitch, x, y)[1] = defaultSim;
        }
}

/* mapKernels.cuh:128-149 */
void avo_normal_map_upscale(float* out, int out_pitch, const float* in, int in_pitch, float ratio, avdm_roi_t roi)
{
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    for(int y = 0; y < roiH; ++y)
        for(int x = 0; x < roiW; ++x)
        {
            const float ox = ((float)x - 0.5f) * ratio;
            const float oy = ((float)y - 0.5f) * ratio;
            int xp = (int)floor(ox + 0.5);
            int yp = (int)floor(oy + 0.5);
            const int mx = (int)((float)roiW * ratio) - 1, my = (int)((float)roiH * ratio) - 1;
            xp = xp < mx ? xp : mx;
            yp = yp < my ? yp : my;
            memcpy(MAP3(out, out_pitch, x, y), MAP3(in, in_pitch, xp, yp), 12);
        }
}

/* mapKernels.cuh:393-477 (wsh = 3) + cuda_stat3d (cuda/device/eig33.cuh:351-445): normal = eigenvector of the smallest eigenvalue of the
 * double-precision covariance matrix of the neighbourhood's 3-D points, oriented towards the camera.  The reference calls its tred2 / tql2
 * routines; the oracle uses the closed-form (trigonometric) eigenvalues of a symmetric 3 x 3 matrix and the cross product of two rows of
 * (A - lambda I) — a third, independent solver; tests/ also check against numpy.linalg.eigh.  Neighbours outside the ROI are skipped on all
 * four sides (the reference reads beyond the tile on the upper sides: stale data of the allocated map). */
static void smallest_eigvec_sym3(const double A[3][3], double v[3])
{
    const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
    double lmin;
    if(p1 == 0.0)
    {
        lmin = A[0][0];
        if(A[1][1] < lmin) lmin = A[1][1];
        if(A[2][2] < lmin) lmin = A[2][2];
    }
    else
    {
        const double p2 = (A[0][0] - q) * (A[0][0] - q) + (A[1][1] - q) * (A[1][1] - q) + (A[2][2] - q) * (A[2][2] - q) + 2.0 * p1;
        const double pp = sqrt(p2 / 6.0);
        double B[3][3];
        for(int i = 0; i < 3; ++i)
            for(int j = 0; j < 3; ++j)
                B[i][j] = (A[i][j] - (i == j ? q : 0.0)) / pp;
        double r = (B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                    B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0])) / 2.0;
        r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
        const double phi = acos(r) / 3.0;
        lmin = q + 2.0 * pp * cos(phi + 2.0 * 3.14159265358979323846 / 3.0); /* the smallest of the three roots */
    }
    double M[3][3];
    for(int i = 0; i < 3; ++i)
        for(int j = 0; j < 3; ++j)
            M[i][j] = A[i][j] - (i == j ? lmin : 0.0);
    /* the null vector of M is orthogonal to its rows: take the largest cross product of two rows */
    double best = -1.0;
    for(int a = 0; a < 3; ++a)
        for(int b = a + 1; b < 3; ++b)
        {
            const double c[3] = {M[a][1] * M[b][2] - M[a][2] * M[b][1], M[a][2] * M[b][0] - M[a][0] * M[b][2], M[a][0] * M[b][1] - M[a][1] * M[b][0]};
            const double n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
            if(n2 > best)
            {
                best = n2;
                v[0] = c[0], v[1] = c[1], v[2] = c[2];
            }
        }
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if(n > 0.0)
        v[0] /= n, v[1] /= n, v[2] /= n;
}

void avo_depth_sim_map_compute_normal(float* out, int out_pitch, const float* depthSim, int in_pitch, const avdm_camera_t* rc, int stepXY, avdm_roi_t roi)
{
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
#pragma omp parallel for schedule(dynamic, 4)
    for(int ry = 0; ry < roiH; ++ry)
        for(int rx = 0; rx < roiW; ++rx)
        {
            float* o = MAP3(out, out_pitch, rx, ry);
            const unsigned x = (roi.x.begin + (unsigned)rx) * (unsigned)stepXY, y = (roi.y.begin + (unsigned)ry) * (unsigned)stepXY;
            const float in_depth = MAP2(depthSim, in_pitch, rx, ry)[0];
            if(in_depth <= 0.0f)
            {
                o[0] = o[1] = o[2] = -1.f;
                continue;
            }
            const f3 p = get3DPointForPixelAndDepthFromRC(rc, mk2((float)x, (float)y), in_depth);
            const float pixSize = size3(sub3(p, get3DPointForPixelAndDepthFromRC(rc, mk2((float)(x + 1), (float)y), in_depth)));
            double xs = 0, ys = 0, zs = 0, xx = 0, yy = 0, zz = 0, xy = 0, xz = 0, yz = 0, count = 0;
            for(int yp = -3; yp <= 3; ++yp)
                for(int xp = -3; xp <= 3; ++xp)
                {
                    const int qx = rx + xp, qy = ry + yp;
                    if(qx < 0 || qy < 0 || qx >= roiW || qy >= roiH)
                        continue;
                    const float depthP = MAP2(depthSim, in_pitch, qx, qy)[0];
                    if((depthP > 0.0f) && (fabsf(depthP - in_depth) < 30.0f * pixSize))
                    {
                        const f3 q = get3DPointForPixelAndDepthFromRC(rc, mk2((float)((int)x + xp), (float)((int)y + yp)), depthP);
                        xx += (double)q.x * (double)q.x, yy += (double)q.y * (double)q.y, zz += (double)q.z * (double)q.z;
                        xy += (double)q.x * (double)q.y, xz += (double)q.x * (double)q.z, yz += (double)q.y * (double)q.z;
                        xs += (double)q.x, ys += (double)q.y, zs += (double)q.z;
                        count += 1.0;
                    }
                }
            if(count < 3.0)
            {
                o[0] = o[1] = o[2] = -1.f;
                continue;
            }
            const double xm = xs / count, ym = ys / count, zm = zs / count;
            double A[3][3], v[3] = {0, 0, 1};
            A[0][0] = (xx - xs * xm - xs * xm + xm * xm * count) / count;
            A[0][1] = A[1][0] = (xy - ys * xm - xs * ym + xm * ym * count) / count;
            A[0][2] = A[2][0] = (xz - zs * xm - xs * zm + xm * zm * count) / count;
            A[1][1] = (yy - ys * ym - ys * ym + ym * ym * count) / count;
            A[1][2] = A[2][1] = (yz - zs * ym - ys * zm + ym * zm * count) / count;
            A[2][2] = (zz - zs * zm - zs * zm + zm * zm * count) / count;
            smallest_eigvec_sym3(A, v);
            f3 nn = normalize3(mk3((float)v[0], (float)v[1], (float)v[2]));
            const f3 pp = mk3((float)xm, (float)ym, (float)zm);
            const f3 nc = normalize3(sub3(cam3(rc->C), p));
            if((dot3(add3(pp, nn), nc) - dot3(pp, nc)) < 0.0f)
                nn = mk3(-nn.x, -nn.y, -nn.z);
            o[0] = nn.x, o[1] = nn.y, o[2] = nn.z;
        }
}

/* mapKernels.cuh:151-211, launch constants Map.cu:72-104.  In place; only the centre pixel's .y is written and
 * neighbours are read for .x only, so the sequential order is immaterial. */
void avo_depth_thickness_smooth_thickness(float* map, int pitch, const avdm_sgm_params_t* sp, const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    const int sgmScaleStep = sp->scale * sp->stepXY;
    const int refineScaleStep = rp->scale * rp->stepXY;
    const float minNbRefineSamples = 2.f;
    const float q = (float)sgmScaleStep / (float)refineScaleStep;
    const float maxNbRefineSamples = q > minNbRefineSamples ? q : minNbRefineSamples;
    const float minThicknessInflate = (float)rp->halfNbDepths / maxNbRefineSamples;
    const float maxThicknessInflate = (float)rp->halfNbDepths / minNbRefineSamples;
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    for(int roiY = 0; roiY < roiH; ++roiY)
        for(int roiX = 0; roiX < roiW; ++roiX)
        {
            float* dt = MAP2(map, pitch, roiX, roiY);
            if(dt[0] <= 0.0f)
                continue;
            const float minThickness = minThicknessInflate * dt[1];
            const float maxThickness = maxThicknessInflate * dt[1];
            float sumCenterDepthDist = 0.f;
            int nbValidPatchPixels = 0;
            for(int yp = -1; yp <= 1; ++yp)
                for(int xp = -1; xp <= 1; ++xp)
                {
                    const int roiXp = roiX + xp, roiYp = roiY + yp;
                    if((xp == 0 && yp == 0) || roiXp < 0 || roiXp >= roiW || roiYp < 0 || roiYp >= roiH)
                        continue;
                    const float* pt = MAP2(map, pitch, roiXp, roiYp);
                    if(pt[0] > 0.0f)
                    {
                        const float depthDistance = fabsf(dt[0] - pt[0]);
                        const float mn = maxThickness < depthDistance ? maxThickness : depthDistance;
                        sumCenterDepthDist += (minThickness > mn ? minThickness : mn);
                        ++nbValidPatchPixels;
                    }
                }
            if(nbValidPatchPixels < 3)
                continue;
            dt[1] = sumCenterDepthDist / (float)nbValidPatchPixels;
        }
}

/* mapKernels.cuh:212-274 (nearest) and :276-391 (bilinear), launch constants Map.cu:106-166 */
void avo_compute_sgm_upscaled_depth_pixsize_map(float* out, int out_pitch, const float* in, int in_pitch, const avdm_camera_t* rc,
                                                const avdm_pyramid_t* rcPyr, const avdm_refine_params_t* rp, float ratio, avdm_roi_t roi)
{
    (void)rc;
    const float rcMipmapLevel = pyr_level(rcPyr, rp->scale);
    const unsigned rcW = pyr_dim_w(rcPyr, rp->scale), rcH = pyr_dim_h(rcPyr, rp->scale);
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
#pragma omp parallel for schedule(static)
    for(int roiY = 0; roiY < roiH; ++roiY)
        for(int roiX = 0; roiX < roiW; ++roiX)
        {
            const unsigned x = (roi.x.begin + roiX) * (unsigned)rp->stepXY;
            const unsigned y = (roi.y.begin + roiY) * (unsigned)rp->stepXY;
            float* o = MAP2(out, out_pitch, roiX, roiY);
            const float alpha = tex2DLod(rcPyr, ((float)x + 0.5f) / (float)rcW, ((float)y + 0.5f) / (float)rcH, rcMipmapLevel).w;
            const float oy = ((float)roiY - 0.5f) * ratio;
            const float ox = ((float)roiX - 0.5f) * ratio;
            float dT[2];
            if(!rp->interpolateMiddleDepth)
            {
                if(alpha < 0.9f) /* sic: 0.9 on a 0..255 scale (mapKernels.cuh:238) */
                {
                    o[0] = -2.f;
                    o[1] = 0.f;
                    continue;
                }
                int xp = (int)floor(ox + 0.5);
                int yp = (int)floor(oy + 0.5);
                const int mx = (int)((float)roiW * ratio) - 1, my = (int)((float)roiH * ratio) - 1;
                xp = xp < mx ? xp : mx;
                yp = yp < my ? yp : my;
                const float* s = MAP2(in, in_pitch, xp, yp);
                dT[0] = s[0];
                dT[1] = s[1];
            }
            else
            {
                if(alpha < (255.f * 0.9f))
                {
                    o[0] = -2.f;
                    o[1] = 0.f;
                    continue;
                }
                int xp = (int)floorf(ox);
                int yp = (int)floorf(oy);
                const int mx = (int)((float)roiW * ratio) - 2, my = (int)((float)roiH * ratio) - 2;
                xp = xp < mx ? xp : mx;
                yp = yp < my ? yp : my;
                /* DEVIATION: the reference reads texel (-1, .) / (., -1) for roiX == 0 / roiY == 0 (floor(-0.5 * ratio) = -1,
                 * mapKernels.cuh:309-321), an out-of-bounds read; we clamp to 0 (off the default path: interpolateMiddleDepth = false) */
                xp = xp < 0 ? 0 : xp;
                yp = yp < 0 ? 0 : yp;
                const float* lu = MAP2(in, in_pitch, xp, yp);
                const float* ru = MAP2(in, in_pitch, xp + 1, yp);
                const float* rd = MAP2(in, in_pitch, xp + 1, yp + 1);
                const float* ld = MAP2(in, in_pitch, xp, yp + 1);
                if(lu[0] <= 0.0f || ru[0] <= 0.0f || rd[0] <= 0.0f || ld[0] <= 0.0f)
                {
                    float sx = 0.f, sy = 0.f;
                    int count = 0;
                    if(lu[0] > 0.0f) { sx = sx + lu[0]; sy = sy + lu[1]; ++count; }
                    if(ru[0] > 0.0f) { sx = sx + ru[0]; sy = sy + ru[1]; ++count; }
                    if(rd[0] > 0.0f) { sx = sx + rd[0]; sy = sy + rd[1]; ++count; }
                    if(ld[0] > 0.0f) { sx = sx + ld[0]; sy = sy + ld[1]; ++count; }
                    if(count != 0)
                    {
                        dT[0] = sx / (float)count;
                        dT[1] = sy / (float)count;
                    }
                    else
                    {
                        o[0] = -1.0f;
                        o[1] = 1.0f;
                        continue;
                    }
                }
                else
                {
                    const float ui = ox - (float)xp;
                    const float vi = oy - (float)yp;
                    const float ux = lu[0] + (ru[0] - lu[0]) * ui, uy = lu[1] + (ru[1] - lu[1]) * ui;
                    const float dx = ld[0] + (rd[0] - ld[0]) * ui, dy = ld[1] + (rd[1] - ld[1]) * ui;
                    dT[0] = ux + (dx - ux) * vi;
                    dT[1] = uy + (dy - uy) * vi;
                }
            }
            o[0] = dT[0];
            o[1] = dT[1] / (float)rp->halfNbDepths;
        }
}

/* mapKernels.cuh:479-515 */
static void optimize_varLofLABtoW(float* out, int out_pitch, const avdm_pyramid_t* rcPyr, unsigned rcW, unsigned rcH, float rcMipmapLevel,
                                  int stepXY, avdm_roi_t roi)
{
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
#pragma omp parallel for schedule(static)
    for(int roiY = 0; roiY < roiH; ++roiY)
        for(int roiX = 0; roiX < roiW; ++roiX)
        {
            const float x = (float)(roi.x.begin + roiX) * (float)stepXY;
            const float y = (float)(roi.y.begin + roiY) * (float)stepXY;
            const float iw = 1.f / (float)rcW, ih = 1.f / (float)rcH;
            const float xM1 = tex2DLod(rcPyr, ((x - 1.f) + 0.5f) * iw, ((y + 0.f) + 0.5f) * ih, rcMipmapLevel).x;
            const float xP1 = tex2DLod(rcPyr, ((x + 1.f) + 0.5f) * iw, ((y + 0.f) + 0.5f) * ih, rcMipmapLevel).x;
            const float yM1 = tex2DLod(rcPyr, ((x + 0.f) + 0.5f) * iw, ((y - 1.f) + 0.5f) * ih, rcMipmapLevel).x;
            const float yP1 = tex2DLod(rcPyr, ((x + 0.f) + 0.5f) * iw, ((y + 1.f) + 0.5f) * ih, rcMipmapLevel).x;
            *MAP1(out, out_pitch, roiX, roiY) = size2(mk2(xM1 - xP1, yM1 - yP1));
        }
}

/* point-sampled, unnormalised, clamped float texture over the whole tmp buffer (memory.hpp:886-916, Map.cu:228-229) */
static inline float tex2D_point(const float* buf, int pitch, int W, int H, float xf, float yf)
{
    int x = (int)floorf(xf), y = (int)floorf(yf);
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    return *MAP1(buf, pitch, x, y);
}

/* mapKernels.cuh:25-101 */
static f2 getCellSmoothStepEnergy(const avdm_camera_t* rc, const float* depthTex, int tex_pitch, int texW, int texH, f2 cell0, f2 offsetRoi)
{
    f2 out = mk2(0.0f, 180.0f);
    const float d0 = tex2D_point(depthTex, tex_pitch, texW, texH, cell0.x, cell0.y);
    if(d0 <= 0.0f)
        return out;
    const f2 cellL = mk2(cell0.x + 0.f, cell0.y + -1.f);
    const f2 cellR = mk2(cell0.x + 0.f, cell0.y + 1.f);
    const f2 cellU = mk2(cell0.x + -1.f, cell0.y + 0.f);
    const f2 cellB = mk2(cell0.x + 1.f, cell0.y + 0.f);
    const float dL = tex2D_point(depthTex, tex_pitch, texW, texH, cellL.x, cellL.y);
    const float dR = tex2D_point(depthTex, tex_pitch, texW, texH, cellR.x, cellR.y);
    const float dU = tex2D_point(depthTex, tex_pitch, texW, texH, cellU.x, cellU.y);
    const float dB = tex2D_point(depthTex, tex_pitch, texW, texH, cellB.x, cellB.y);
    const f3 p0 = get3DPointForPixelAndDepthFromRC(rc, mk2(cell0.x + offsetRoi.x, cell0.y + offsetRoi.y), d0);
    const f3 pL = get3DPointForPixelAndDepthFromRC(rc, mk2(cellL.x + offsetRoi.x, cellL.y + offsetRoi.y), dL);
    const f3 pR = get3DPointForPixelAndDepthFromRC(rc, mk2(cellR.x + offsetRoi.x, cellR.y + offsetRoi.y), dR);
    const f3 pU = get3DPointForPixelAndDepthFromRC(rc, mk2(cellU.x + offsetRoi.x, cellU.y + offsetRoi.y), dU);
    const f3 pB = get3DPointForPixelAndDepthFromRC(rc, mk2(cellB.x + offsetRoi.x, cellB.y + offsetRoi.y), dB);
    f3 cg = mk3(0.0f, 0.0f, 0.0f);
    float n = 0.0f;
    if(dL > 0.0f) { cg = add3(cg, pL); n++; }
    if(dR > 0.0f) { cg = add3(cg, pR); n++; }
    if(dU > 0.0f) { cg = add3(cg, pU); n++; }
    if(dB > 0.0f) { cg = add3(cg, pB); n++; }
    if(n > 1.0f)
    {
        cg = div3(cg, n);
        const f3 vcn = normalize3(sub3(cam3(rc->C), p0));
        const f3 pS = closestPointToLine3D(cg, p0, vcn);
        out.x = size3(sub3(cam3(rc->C), pS)) - d0;
    }
    float e = 0.0f;
    n = 0.0f;
    if(dL > 0.0f && dR > 0.0f)
    {
        e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pL, pR)));
        n++;
    }
    if(dU > 0.0f && dB > 0.0f)
    {
        e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pU, pB)));
        n++;
    }
    if(n > 0.0f)
        out.y = e;
    return out;
}

/* Map.cu:193-263 (cuda_depthSimMapOptimizeGradientDescent), mapKernels.cuh:517-608 */
void avo_depth_sim_map_optimize_gradient_descent(float* outOpt, int out_pitch, float* imgVariance, int var_pitch, float* tmpDepth, int tmp_pitch,
                                                 int tmpW, int tmpH, const float* sgmDepthPixSize, int sgm_pitch, const float* refineDepthSim,
                                                 int ref_pitch, const avdm_camera_t* rc, const avdm_pyramid_t* rcPyr,
                                                 const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    const float rcMipmapLevel = pyr_level(rcPyr, rp->scale);
    const unsigned rcW = pyr_dim_w(rcPyr, rp->scale), rcH = pyr_dim_h(rcPyr, rp->scale);
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);

    /* out_optimizeDepthSimMap_dmp.copyFrom(in_sgmDepthPixSizeMap_dmp) — over the ROI (callers allocate >= ROI) */
    for(int y = 0; y < roiH; ++y)
        memcpy(MAP2(outOpt, out_pitch, 0, y), MAP2(sgmDepthPixSize, sgm_pitch, 0, y), (size_t)roiW * 8);

    optimize_varLofLABtoW(imgVariance, var_pitch, rcPyr, rcW, rcH, rcMipmapLevel, rp->stepXY, roi);

    for(int iter = 0; iter < rp->optimizationNbIterations; ++iter)
    {
        /* optimize_getOptDeptMapFromOptDepthSimMap_kernel :517-529 */
#pragma omp parallel for schedule(static)
        for(int y = 0; y < roiH; ++y)
            for(int x = 0; x < roiW; ++x)
                *MAP1(tmpDepth, tmp_pitch, x, y) = MAP2(outOpt, out_pitch, x, y)[0];

            /* optimize_depthSimMap_kernel :531-608 */
#pragma omp parallel for schedule(dynamic, 4)
        for(int roiY = 0; roiY < roiH; ++roiY)
            for(int roiX = 0; roiX < roiW; ++roiX)
            {
                const float* sgm = MAP2(sgmDepthPixSize, sgm_pitch, roiX, roiY);
                const float sgmDepth = sgm[0], sgmPixSize = sgm[1];
                const float* rf = MAP2(refineDepthSim, ref_pitch, roiX, roiY);
                const float refineDepth = rf[0], refineSim = rf[1];
                float* o = MAP2(outOpt, out_pitch, roiX, roiY);
                f2 outDS = (iter == 0) ? mk2(sgmDepth, refineSim) : mk2(o[0], o[1]);
                const float depthOpt = outDS.x;
                if(depthOpt > 0.0f)
                {
                    const f2 se = getCellSmoothStepEnergy(rc, tmpDepth, tmp_pitch, tmpW, tmpH, mk2((float)roiX, (float)roiY),
                                                          mk2((float)roi.x.begin, (float)roi.y.begin));
                    float stepToSmoothDepth = se.x;
                    stepToSmoothDepth = copysignf(fminf(fabsf(stepToSmoothDepth), sgmPixSize / 10.0f), stepToSmoothDepth);
                    const float depthEnergy = se.y;
                    float stepToFineDM = refineDepth - depthOpt;
                    stepToFineDM = copysignf(fminf(fabsf(stepToFineDM), sgmPixSize / 10.0f), stepToFineDM);
                    const float stepToRoughDM = sgmDepth - depthOpt;
                    const float imgColorVariance = *MAP1(imgVariance, var_pitch, roiX, roiY);
                    const float colorVarianceThresholdForSmoothing = 20.0f;
                    const float angleThresholdForSmoothing = 30.0f;
                    const float weightedColorVariance =
                      sigmoid2f_(5.0f, angleThresholdForSmoothing, 40.0f, colorVarianceThresholdForSmoothing, imgColorVariance);
                    const float fineSimWeight = sigmoidf_(0.0f, 1.0f, 0.7f, -0.7f, refineSim);
                    const float energyLowerThanVarianceWeight = sigmoidf_(0.0f, 1.0f, 30.0f, weightedColorVariance, depthEnergy);
                    const float closeToRoughWeight = 1.0f - sigmoidf_(0.0f, 1.0f, 10.0f, 17.0f, fabsf(stepToRoughDM / sgmPixSize));
                    const float depthOptStep =
                      closeToRoughWeight * stepToRoughDM +
                      (1.0f - closeToRoughWeight) *
                        (energyLowerThanVarianceWeight * fineSimWeight * stepToFineDM + (1.0f - energyLowerThanVarianceWeight) * stepToSmoothDepth);
                    outDS.x = depthOpt + depthOptStep;
                    outDS.y = (1.0f - closeToRoughWeight) * (energyLowerThanVarianceWeight * fineSimWeight * refineSim +
                                                             (1.0f - energyLowerThanVarianceWeight) * (depthEnergy / 20.0f));
                }
                o[0] = outDS.x;
                o[1] = outDS.y;
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Test hooks: the static helpers above on arrays, so that tests/test_oracle_ref.py can hold them to the vectors the reference's own
 * device helpers produced (tests/golden/ref_helpers.npz, generated through oracle/_ref by tests/golden/make_golden.py).
 * ---------------------------------------------------------------------------------------------- */
void avo_test_rgb2lab(const float* rgb01, int n, float* lab)
{
    for(int i = 0; i < n; ++i)
    {
        const f3 l = xyz2lab(rgb2xyz(mk3(rgb01[3 * i], rgb01[3 * i + 1], rgb01[3 * i + 2])));
        lab[3 * i] = l.x; lab[3 * i + 1] = l.y; lab[3 * i + 2] = l.z;
    }
}
void avo_test_cost_yk_from_lab(const int* dxdy, const float* c1c2, int n, float invGammaC, float invGammaP, float* out)
{
    for(int i = 0; i < n; ++i)
    {
        const float* c = c1c2 + 8 * i;
        const f4 a = {c[0], c[1], c[2], c[3]}, b = {c[4], c[5], c[6], c[7]};
        out[i] = CostYKfromLab(dxdy[2 * i], dxdy[2 * i + 1], a, b, invGammaC, invGammaP);
    }
}
/* simStat::update(gx, gy, w) + computeWSim (SimStat.cuh:72-113,146-153): the statement sequence of compNCCby3DptsYK's fp32 branch */
void avo_test_sim_stat_wsim(const float* gxgyw, int m, int n, float* out)
{
    for(int i = 0; i < n; ++i)
    {
        float wsum = 0, xsum = 0, ysum = 0, xxsum = 0, yysum = 0, xysum = 0;
        for(int k = 0; k < m; ++k)
        {
            const float* g = gxgyw + 3 * ((size_t)i * m + k);
            const float gx = g[0], gy = g[1], w = g[2];
            wsum += w;
            xsum += w * gx;
            ysum += w * gy;
            xxsum += w * gx * gx;
            yysum += w * gy * gy;
            xysum += w * gx * gy;
        }
        const float varXW = (xxsum - xsum * xsum / wsum) / wsum;
        const float varYW = (yysum - ysum * ysum / wsum) / wsum;
        const float varXYW = (xysum - xsum * ysum / wsum) / wsum;
        const float rawSim = varXYW / sqrtf(varXW * varYW);
        out[i] = isfinite(rawSim) ? -rawSim : 1.0f;
    }
}
void avo_test_sigmoid(const float* zv, int n, float zeroVal, float endVal, float sigwidth, float sigMid, float* out, float* out2)
{
    for(int i = 0; i < n; ++i)
    {
        out[i] = sigmoidf_(zeroVal, endVal, sigwidth, sigMid, zv[i]);
        out2[i] = sigmoid2f_(zeroVal, endVal, sigwidth, sigMid, zv[i]);
    }
}
void avo_test_project3d(const float* P12, const float* pts, int n, float* out2)
{
    for(int i = 0; i < n; ++i)
    {
        const f2 r = project3DPoint(P12, mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        out2[2 * i] = r.x; out2[2 * i + 1] = r.y;
    }
}
