// avdm_device.h — device-side helpers shared by the gfx950 kernels:
//   small vector math, camera projection, the software texture unit over the fp16 Lab pyramid,
//   wave64 DPP reductions.  Written for CDNA4 only (wave = 64, no texture hardware path).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/avdm.h"

namespace avdm {

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
int set_error(hipError_t e, const char* where);
int set_error_msg(int code, const char* msg);
#define AVDM_LAUNCH_CHECK(name) return ::avdm::set_error(hipGetLastError(), name)

// Library-owned temporary of an entry point whose reference signature has no room for it: one buffer per (device, stream), grown on
// demand and reused by the later calls on that stream (stream order makes the reuse safe).  NOT the stream-ordered allocator: with two
// host threads on one device, hipMallocAsync in one and hipEventRecord / hipStreamWaitEvent in the other deadlocked inside the runtime
// about once in 25 two-worker runs (profiles/r02_multiworker_hang.md).  nullptr when the allocation fails.
// An entry point holds a StreamScratch object while it enqueues the work that uses the block: a block in use is never evicted or regrown
// under it by another host thread.  avdm_stream_release(stream) gives a stream's block back (call it before destroying the stream).
class StreamScratch
{
  public:
    StreamScratch(hipStream_t st, size_t bytes);
    ~StreamScratch();
    StreamScratch(const StreamScratch&) = delete;
    StreamScratch& operator=(const StreamScratch&) = delete;
    void* ptr() const { return _ptr; }

  private:
    hipStream_t _st;
    int _dev = 0;
    void* _ptr = nullptr;
    bool _private = false;
};
int stream_scratch_release(hipStream_t st);

static inline unsigned divUp(unsigned a, unsigned b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------------------------
struct f3
{
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float d) { return f3{a.x * d, a.y * d, a.z * d}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float size(f3 a) { return sqrtf(dot(a, a)); } // correctly rounded (hipcc default)
// reference: a * __fdividef(1, sqrtf(dot)) — one v_rsq_f32 here (1 ulp), same contract as the CUDA fast intrinsic
__device__ __forceinline__ f3 normalize(f3 a)
{
    const float dInv = __builtin_amdgcn_rsqf(dot(a, a)); // v_rsq_f32
    return f3{a.x * dInv, a.y * dInv, a.z * dInv};
}
__device__ __forceinline__ f3 ld3(const float* v) { return f3{v[0], v[1], v[2]}; }

__device__ __forceinline__ f3 M3x3mulV2(const float* M, float vx, float vy)
{
    return f3{M[0] * vx + M[3] * vy + M[6], M[1] * vx + M[4] * vy + M[7], M[2] * vx + M[5] * vy + M[8]};
}
__device__ __forceinline__ f3 M3x3mulV3(const float* M, f3 V)
{
    return f3{M[0] * V.x + M[3] * V.y + M[6] * V.z, M[1] * V.x + M[4] * V.y + M[7] * V.z, M[2] * V.x + M[5] * V.y + M[8] * V.z};
}
__device__ __forceinline__ f3 M3x4mulV3(const float* M, f3 V)
{
    return f3{M[0] * V.x + M[3] * V.y + M[6] * V.z + M[9], M[1] * V.x + M[4] * V.y + M[7] * V.z + M[10],
              M[2] * V.x + M[5] * V.y + M[8] * V.z + M[11]};
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); } // v_rcp_f32, 1 ulp (== __fdividef contract)
__device__ __forceinline__ float2 project3DPoint(const float* P, f3 V)
{
    const f3 p = M3x4mulV3(P, V);
    const float inv = fast_rcp(p.z);
    return make_float2(p.x * inv, p.y * inv);
}
__device__ __forceinline__ f3 linePlaneIntersect(f3 linePoint, f3 lineVect, f3 planePoint, f3 planeNormal)
{
    const float k = (dot(planePoint, planeNormal) - dot(planeNormal, linePoint)) / dot(planeNormal, lineVect);
    return linePoint + lineVect * k;
}
__device__ __forceinline__ f3 closestPointToLine3D(f3 point, f3 linePoint, f3 lineVectNormalized)
{
    return linePoint + lineVectNormalized * dot(lineVectNormalized, point - linePoint);
}
__device__ __forceinline__ float pointLineDistance3D(f3 point, f3 linePoint, f3 lineVectNormalized)
{
    return size(cross(lineVectNormalized, linePoint - point));
}
__device__ __forceinline__ float sigmoid(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
    return zeroVal + (endVal - zeroVal) * (1.0f / (1.0f + expf(10.0f * ((xval - sigMid) / sigwidth))));
}
__device__ __forceinline__ float sigmoid2(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
    return zeroVal + (endVal - zeroVal) * (1.0f / (1.0f + expf(10.0f * ((sigMid - xval) / sigwidth))));
}

// ---------------------------------------------------------------------------------------------
// camera helpers (Patch.cuh:137-170 of the reference, restated)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float computePixSize(const avdm_camera_t& cam, f3 p)
{
    const float2 rp = project3DPoint(cam.P, p);
    const f3 refvect = normalize(M3x3mulV2(cam.iP, rp.x + 1.0f, rp.y));
    return pointLineDistance3D(p, ld3(cam.C), refvect);
}
__device__ __forceinline__ f3 get3DPointForPixelAndFrontoParellePlaneRC(const avdm_camera_t& cam, float px, float py, float fpPlaneDepth)
{
    const f3 C = ld3(cam.C), Z = ld3(cam.ZVect);
    const f3 planep = C + Z * fpPlaneDepth;
    const f3 v = normalize(M3x3mulV2(cam.iP, px, py));
    return linePlaneIntersect(C, v, planep, Z);
}
__device__ __forceinline__ f3 get3DPointForPixelAndDepthFromRC(const avdm_camera_t& cam, float px, float py, float depth)
{
    const f3 rpv = normalize(M3x3mulV2(cam.iP, px, py));
    return ld3(cam.C) + rpv * depth;
}

// ---------------------------------------------------------------------------------------------
// software texture unit
// ---------------------------------------------------------------------------------------------
struct TexLevel
{
    const uint2* base; // fp16x4 texels
    int W, H;
    int pitch8; // row pitch in texels
};

struct Tex
{
    TexLevel lv[AVDM_MAX_LEVELS];
    int levels;
    int mode; // AVDM_FILTER_*
    int min_downscale, width0, height0;
};

inline Tex make_tex(const avdm_pyramid_t* p)
{
    Tex t;
    for(int l = 0; l < AVDM_MAX_LEVELS; ++l)
    {
        t.lv[l].base = (const uint2*)((const char*)p->base + p->offset[l]);
        t.lv[l].W = p->width[l];
        t.lv[l].H = p->height[l];
        t.lv[l].pitch8 = p->pitch[l] / 8;
    }
    t.levels = p->levels;
    t.mode = p->filter_mode;
    t.min_downscale = p->min_downscale;
    t.width0 = p->width0;
    t.height0 = p->height0;
    return t;
}
// DeviceMipmapImage::getLevel / getDimensions
inline float tex_level_of(const avdm_pyramid_t* p, int downscale) { return log2f((float)downscale / (float)p->min_downscale); }
inline int tex_dim_w(const avdm_pyramid_t* p, int downscale) { return (p->width0 + downscale - 1) / downscale; }
inline int tex_dim_h(const avdm_pyramid_t* p, int downscale) { return (p->height0 + downscale - 1) / downscale; }
inline bool lod_is_integral(const avdm_pyramid_t* p, int downscale, int* level)
{
    const float l = tex_level_of(p, downscale);
    const int li = (int)l;
    *level = li < 0 ? 0 : (li > p->levels - 1 ? p->levels - 1 : li);
    return (float)li == l;
}

__device__ __forceinline__ float4 unpack_h4(uint2 t)
{
    const __half2 lo = *reinterpret_cast<const __half2*>(&t.x);
    const __half2 hi = *reinterpret_cast<const __half2*>(&t.y);
    const float2 a = __half22float2(lo), b = __half22float2(hi);
    return make_float4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ uint2 pack_h4(float4 c)
{
    // The four values are fp32 RESULTS, rounded as such, before they are rounded to fp16 — like the reference's `__float2half(x * k)`.  Without
    // the fence the compiler folds a producing multiplication into the conversion (v_fma_mixlo_f16: ONE rounding of the exact product), which
    // moved the last bit of ~0.1 % of the texels of the Lab pyramid against the reference's (found in session r06_b by the ISA of rgb2lab_kernel).
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
    const __half2 lo = __floats2half2_rn(c.x, c.y);
    const __half2 hi = __floats2half2_rn(c.z, c.w);
    uint2 r;
    r.x = *reinterpret_cast<const unsigned*>(&lo);
    r.y = *reinterpret_cast<const unsigned*>(&hi);
    return r;
}

__device__ __forceinline__ float quant8(float a) { return floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f); }

__device__ __forceinline__ float4 texel_clamped(const TexLevel& L, int x, int y)
{
    x = min(max(x, 0), L.W - 1);
    y = min(max(y, 0), L.H - 1);
    return unpack_h4(L.base[(long long)y * L.pitch8 + x]);
}

// the bilinear blend of the four taps — ONE expression for every tap source (global memory or an LDS window)
__device__ __forceinline__ float4 bilinear_blend(float4 t00, float4 t10, float4 t01, float4 t11, float a, float b)
{
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}

// bilinear fetch at texel-space coordinates (x, y) = (u*W - 0.5, v*H - 0.5)
template <bool FIXED8>
__device__ __forceinline__ float4 tex_bilinear_px(const TexLevel& L, float x, float y)
{
    const float fx = floorf(x), fy = floorf(y);
    float a = x - fx, b = y - fy;
    if(FIXED8)
    {
        a = quant8(a);
        b = quant8(b);
    }
    const int i = (int)fx, j = (int)fy;
    float4 t00, t10, t01, t11;
    if(i >= 0 && j >= 0 && i < L.W - 1 && j < L.H - 1)
    {
        const uint2* r0 = L.base + (long long)j * L.pitch8 + i;
        const uint2* r1 = r0 + L.pitch8;
        t00 = unpack_h4(r0[0]);
        t10 = unpack_h4(r0[1]);
        t01 = unpack_h4(r1[0]);
        t11 = unpack_h4(r1[1]);
    }
    else
    {
        t00 = texel_clamped(L, i, j);
        t10 = texel_clamped(L, i + 1, j);
        t01 = texel_clamped(L, i, j + 1);
        t11 = texel_clamped(L, i + 1, j + 1);
    }
    return bilinear_blend(t00, t10, t01, t11, a, b);
}

// tex2DLod with normalised coordinates at ONE integral level
template <bool FIXED8>
__device__ __forceinline__ float4 tex2D_level(const TexLevel& L, float u, float v)
{
    return tex_bilinear_px<FIXED8>(L, u * (float)L.W - 0.5f, v * (float)L.H - 0.5f);
}

// full tex2DLod (fractional level of detail) — used only with useConsistentScale
__device__ __forceinline__ float4 tex2DLod(const Tex& T, float u, float v, float lod)
{
    const float maxl = (float)(T.levels - 1);
    lod = !(lod > 0.0f) ? 0.0f : (lod > maxl ? maxl : lod);
    const float fl = floorf(lod);
    float g = lod - fl;
    const bool fixed8 = T.mode == AVDM_FILTER_CUDA_FIXED8;
    if(fixed8)
        g = quant8(g);
    const int l0 = (int)fl;
    const float4 c0 = fixed8 ? tex2D_level<true>(T.lv[l0], u, v) : tex2D_level<false>(T.lv[l0], u, v);
    if(g == 0.0f || l0 + 1 >= T.levels)
        return c0;
    const float4 c1 = fixed8 ? tex2D_level<true>(T.lv[l0 + 1], u, v) : tex2D_level<false>(T.lv[l0 + 1], u, v);
    float4 r;
    r.x = (1.0f - g) * c0.x + g * c1.x;
    r.y = (1.0f - g) * c0.y + g * c1.y;
    r.z = (1.0f - g) * c0.z + g * c1.z;
    r.w = (1.0f - g) * c0.w + g * c1.w;
    return r;
}

// tex2DLod at ONE level of detail fixed for a whole launch (a stage's level, DeviceMipmapImage::getLevel): the clamp, the floor and the
// quantised blend fraction of tex2DLod are evaluated once on the host, the kernel blends two tex2D_level fetches — or fetches one when the
// level is integral (the default scale combinations).  Same operations as tex2DLod on the same operands: identical floats.  This is
// what lets the stages run at FRACTIONAL levels (sgmScale / refineScale that are not a power-of-two multiple of the pyramid's first level,
// e.g. --sgmScale 3 --refineScale 1; DeviceMipmapImage.cpp:92-99, deviceMipmappedArray.cu:348).
struct TexLod
{
    TexLevel l0, l1;
    float g;  // blend fraction towards l1 (quantised to 1/256 in FIXED8 mode)
    int two;  // 0: l0 alone
};
inline TexLod make_tex_lod(const avdm_pyramid_t* p, int downscale)
{
    const Tex t = make_tex(p);
    const float maxl = (float)(t.levels - 1);
    float lod = tex_level_of(p, downscale);
    lod = !(lod > 0.0f) ? 0.0f : (lod > maxl ? maxl : lod);
    const float fl = floorf(lod);
    float g = lod - fl;
    if(t.mode == AVDM_FILTER_CUDA_FIXED8)
        g = floorf(g * 256.0f + 0.5f) * (1.0f / 256.0f); // quant8
    const int l0 = (int)fl;
    TexLod L;
    L.l0 = t.lv[l0];
    L.two = (g != 0.0f && l0 + 1 < t.levels) ? 1 : 0;
    L.l1 = t.lv[L.two ? l0 + 1 : l0];
    L.g = g;
    return L;
}
template <bool FIXED8>
__device__ __forceinline__ float4 tex2D_lod(const TexLod& T, float u, float v)
{
    const float4 c0 = tex2D_level<FIXED8>(T.l0, u, v);
    if(!T.two)
        return c0;
    const float4 c1 = tex2D_level<FIXED8>(T.l1, u, v);
    const float g = T.g;
    float4 r;
    r.x = (1.0f - g) * c0.x + g * c1.x;
    r.y = (1.0f - g) * c0.y + g * c1.y;
    r.z = (1.0f - g) * c0.z + g * c1.z;
    r.w = (1.0f - g) * c0.w + g * c1.w;
    return r;
}

// ---------------------------------------------------------------------------------------------
// wave64 helpers (DPP)
// ---------------------------------------------------------------------------------------------
// dpp_ctrl encodings (GFX9): row_shl:n 0x100+n, row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138,
// row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float oldv, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned oldv, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)oldv, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}

// min over the 64 lanes of a wave; result valid in every lane (via readlane 63 -> SGPR)
__device__ __forceinline__ float wave_min_f32(float v)
{
    v = fminf(v, dpp_f32<0x111>(v, v));
    v = fminf(v, dpp_f32<0x112>(v, v));
    v = fminf(v, dpp_f32<0x114>(v, v));
    v = fminf(v, dpp_f32<0x118>(v, v));
    v = fminf(v, dpp_f32<0x142, 0xa>(v, v));
    v = fminf(v, dpp_f32<0x143, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0x111>(v, v));
    v = min(v, dpp_u32<0x112>(v, v));
    v = min(v, dpp_u32<0x114>(v, v));
    v = min(v, dpp_u32<0x118>(v, v));
    v = min(v, dpp_u32<0x142, 0xa>(v, v));
    v = min(v, dpp_u32<0x143, 0xc>(v, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

} // namespace avdm
