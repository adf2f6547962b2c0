// avdm_literal.hip — the reference's similarity arithmetic AS WRITTEN, on the GPU.  Compiled with -ffp-contract=off.
//
// Two users of one restatement (ncc_literal below):
//
//   * REFERENCE-ARITHMETIC MODE of the product (avdm_sgm_params_t::referenceArithmetic / avdm_refine_params_t::referenceArithmetic, the
//     CLI's --sgmReferenceArithmetic / --refineReferenceArithmetic): strict_sgm_kernel / strict_refine_kernel.  Every operation of
//     compNCCby3DptsYK in the reference's order — each patch sample a 3-D point through the 3 x 4 matrices with IEEE divisions, the border
//     test and the centre colour on the re-projected R pixel, two Yoon-Kweon weights with two exponentials, the six UNSHIFTED fp32 sums, no
//     FMA contraction — and the exponential evaluated to the bits of the C library the pinned reference build calls (avdm_libm.h): the
//     similarity volumes then equal those of the reference's own code compiled for the CPU (oracle/_ref) BIT FOR BIT, and with them the
//     winner-take-all decisions the default kernels' better-conditioned arithmetic re-draws on low-texture tiles.
//     Laid out for the machine like the default kernels: 16 x 16 pixels per workgroup, the R footprint and the hull of the workgroup's T
//     patches staged in LDS once per block of planes and used as a CACHE — a tap whose four texels lie in the window reads them from LDS,
//     any other tap takes the same texels from global memory through the same blend, so the result never depends on the window.
//     Cost: ~6 x the default sweep's instructions per sample (two 3 x 4 projections, two IEEE divisions, two IEEE square roots, two
//     double-precision exponentials per sample where the default path issues ~34 packed instructions); measured in bench.py's line.
//
//   * AVDM_SIM_LITERAL=1 (environment, read at each call): the ATTRIBUTION switch of rounds 3-5 — one lane per pixel, every tap from global
//     memory, AVDM_SIM_LITERAL_DEV=<bits> re-introducing the default kernels' deviations one at a time (scripts/deviation_report.py).
//
//   compNCCby3DptsYK        Patch.cuh:466-572      every patch sample as a 3-D point projected into both cameras; the border test and
//                                                  the centre colour on the RE-PROJECTED R pixel
//   simStat                 SimStat.cuh:72-153     the six UNSHIFTED fp32 sums of w, w L, w L^2 (L ~ 0 ... 255): variance = difference
//                                                  of numbers ~ 5e6
//   CostYKfromLab           color.cuh:167-210      two Yoon-Kweon weights, two exponentials, multiplied
//   volume_computeSimilarity_kernel / volume_refineSimilarity_kernel   deviceSimilarityVolumeKernels.cuh:109-233, 235-391
#include "avdm_device.h"
#include "avdm_libm.h"

#include <limits.h>
#include <math.h>
#include <mutex>
#include <stdlib.h>

namespace avdm {

namespace {

// matrix.cuh:66-75: a * __fdividef(1, sqrtf(dot)) — evaluated as an IEEE division like the reference compiled for the CPU
__device__ __forceinline__ f3 normalize_lit(f3 a)
{
    const float dInv = 1.0f / sqrtf(dot(a, a));
    return f3{a.x * dInv, a.y * dInv, a.z * dInv};
}
// matrix.cuh:117-126
__device__ __forceinline__ float2 project_lit(const float* P, f3 V)
{
    const f3 q = M3x4mulV3(P, V);
    const float inv = 1.0f / q.z;
    return make_float2(q.x * inv, q.y * inv);
}
// Patch.cuh:137-145
__device__ __forceinline__ float pix_size_lit(const avdm_camera_t& cam, f3 p)
{
    const float2 rp = project_lit(cam.P, p);
    const f3 refvect = normalize_lit(M3x3mulV2(cam.iP, rp.x + 1.0f, rp.y + 0.0f));
    return size(cross(refvect, ld3(cam.C) - p));
}
// the table of glibc's expf (avdm_libm.h); the strict kernels keep a copy in LDS (the index differs from lane to lane)
__device__ const uint64_t g_exp2f_tab[32] = AVDM_EXP2F_TAB;

// color.cuh:167-210; expf = the pinned build's C library, to its bits (avdm_libm.h)
__device__ __forceinline__ float cost_yk(int dx, int dy, float4 c1, float4 c2, float invGammaC, float invGammaP, const uint64_t* expTab)
{
    const float ex = c1.x - c2.x, ey = c1.y - c2.y, ez = c1.z - c2.z;
    float deltaC = sqrtf(ex * ex + ey * ey + ez * ez);
    deltaC *= invGammaC;
    float deltaP = sqrtf((float)(dx * dx + dy * dy));
    deltaP *= invGammaP;
    deltaC += deltaP;
    return glibc::expf_tab(-deltaC, expTab);
}
// the same with deltaP = sqrtf(dx^2 + dy^2) * invGammaP handed in (a correctly rounded square root and one fp32 product of two values that
// only depend on the patch position: evaluated once per launch on the host, the same two IEEE operations)
__device__ __forceinline__ float cost_yk_tab(float deltaP, float4 c1, float4 c2, float invGammaC, const uint64_t* expTab)
{
    const float ex = c1.x - c2.x, ey = c1.y - c2.y, ez = c1.z - c2.z;
    float deltaC = sqrtf(ex * ex + ey * ey + ez * ez);
    deltaC *= invGammaC;
    deltaC += deltaP;
    return glibc::expf_tab(-deltaC, expTab);
}
// matrix.cuh:334-337
__device__ __forceinline__ float sigmoid_lit(float zeroVal, float endVal, float sigwidth, float sigMid, float xval, const uint64_t* expTab)
{
    return zeroVal + (endVal - zeroVal) * (1.0f / (1.0f + glibc::expf_tab(10.0f * ((xval - sigMid) / sigwidth), expTab)));
}

// AVDM_SIM_LITERAL_DEV=<bit mask> (read at each call): the deviations of the default kernels (avdm_similarity.hip) from the reference's
// arithmetic, introduced into the literal evaluation one at a time, so that the distance default <-> reference can be attributed deviation by
// deviation (scripts/deviation_report.py, tests/test_gpu_parity.py::test_deviation_attribution):
enum
{
    DEV_SHIFTED_SUMS = 1,  // NCC statistics of L(centre) - L(sample) instead of L(sample) (shift invariant; the sums no longer cancel), FMA sums, rcp / rsq finish
    DEV_MERGED_EXP = 2,    // one hardware exp2 of the summed exponents instead of two expf multiplied; hardware sqrt of the colour distances
    DEV_HOMOGENEOUS = 4,   // sample positions as h0 + a (M x d) + b (M y d) with v_rcp_f32 instead of a 3-D point through the 3 x 4 matrix and a division
    DEV_EXACT_PIXEL = 8,   // centre colour of R fetched at the lane's own pixel instead of the re-projected patch centre
    DEV_SHARED_R = 16,     // R side of a sample (position, taps) from the patch of plane 1 of the aligned group of four planes
    DEV_EXACT_BORDER = 32, // rounds 1-3 only (NOT a deviation of the default kernels any more: they evaluate the reference's own test on the
                           // knife-edge rows, avdm_similarity.hip lit::): the R-side border test on the lane's own pixel
};

struct LitArgs
{
    float rcW, rcH, tcW, tcH; // nominal level dimensions (DeviceMipmapImage::getDimensions)
    float invGammaC, invGammaP;
    int wsh;
    int dev; // DEV_* bits (the attribution kernels only)
};

struct LitPatch
{
    f3 p, x, y;
    float d;
};

// ---- tap sources: tex2DLod(u, v) of one image --------------------------------------------------------------------------------------------
// the software texture unit straight from global memory, any level of detail (the attribution kernels; fractional levels)
struct TexTap
{
    Tex T;
    float lod;
    __device__ __forceinline__ float4 fetch(float u, float v) { return tex2DLod(T, u, v, lod); }
    __device__ __forceinline__ bool missed() const { return false; }
};
// One INTEGRAL level with an LDS window of its texels [x0, x0 + w) x [y0, y0 + h) (plain fp16 x 4 records, row pitch `pitch`).  tex2DLod at an
// integral level is tex2D_level (avdm_device.h: the level clamp, the floor and a zero blend fraction leave one bilinear fetch); the fetch below
// is tex_bilinear_px operation for operation — the texels come from the window when all four lie inside it, from global memory otherwise:
// same texels, same weights, same blend.
//   CHECKED = true : the window is a cache, every tap decides for itself (a divergent branch per tap);
//   CHECKED = false: branch-free — the tap's window address is clamped into the window and a tap that had to be clamped raises `miss`; the caller
//                    evaluates the plane once more with the checked form when any lane of its wave missed (rare: the window is the hull of the
//                    workgroup's patches).
template <bool FIXED8, bool CHECKED>
struct WinTap
{
    TexLevel L;
    const uint2* win;
    int pitch, x0, y0;
    unsigned wm1, hm1; // w - 1, h - 1 (0: no window)
    bool miss;
    __device__ __forceinline__ bool missed() const { return miss; }
    __device__ __forceinline__ float4 fetch(float u, float v)
    {
        const float x = u * (float)L.W - 0.5f, y = v * (float)L.H - 0.5f; // tex2D_level
        const float fx = floorf(x), fy = floorf(y);
        float a = x - fx, b = y - fy;
        if(FIXED8)
        {
            a = quant8(a);
            b = quant8(b);
        }
        const int i = (int)fx, j = (int)fy;
        const int wi = i - x0, wj = j - y0;
        float4 t00, t10, t01, t11;
        if(!CHECKED)
        {
            const int ci = min(max(wi, 0), (int)wm1 - 1), cj = min(max(wj, 0), (int)hm1 - 1);
            miss = miss || (ci != wi) || (cj != wj);
            const int o00 = cj * pitch + ci;
            int o10 = o00 + 1;
            asm volatile("" : "+v"(o10)); // two ds_read_b64, not one ds_read2_b64 (half the LDS rate)
            const int o01 = o00 + pitch;
            int o11 = o01 + 1;
            asm volatile("" : "+v"(o11));
            t00 = unpack_h4(win[o00]);
            t10 = unpack_h4(win[o10]);
            t01 = unpack_h4(win[o01]);
            t11 = unpack_h4(win[o11]);
        }
        else if((unsigned)wi < wm1 && (unsigned)wj < hm1)
        {
            const int o00 = wj * pitch + wi, o01 = o00 + pitch;
            t00 = unpack_h4(win[o00]);
            t10 = unpack_h4(win[o00 + 1]);
            t01 = unpack_h4(win[o01]);
            t11 = unpack_h4(win[o01 + 1]);
        }
        else if(i >= 0 && j >= 0 && i < L.W - 1 && j < L.H - 1)
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
};

// Patch.cuh:466-572 + SimStat.cuh; INFINITY when the patch is invalid.  T = the plane's own patch; Rp = the patch the R side is sampled
// from (the same one unless DEV_SHARED_R); (x, y) = the lane's pixel (DEV_EXACT_PIXEL).
// DEVS: the AVDM_SIM_LITERAL_DEV switches are live (attribution kernels); false = the reference's arithmetic, nothing else.
// dPtab: sqrtf(xp^2 + yp^2) * invGammaP per patch position, row-major (nullptr: evaluated per sample as written).
// WSH: the patch half-width as a compile-time constant (0: L.wsh) — what lets the sample loop of the strict kernels be unrolled by UNROLL
template <bool TInvert, bool DEVS, int UNROLL, int WSH, class RTap, class TTap>
__device__ __forceinline__ float ncc_literal(const avdm_camera_t& rc, const avdm_camera_t& tc, const LitArgs& L, const LitPatch& T, const LitPatch& Rp, float x,
                                             float y, RTap& rt, TTap& tt, const float* dPtab, const uint64_t* expTab)
{
    const int dev = DEVS ? L.dev : 0;
    const int wsh = WSH > 0 ? WSH : L.wsh;
    const f3 pp = T.p;
    float2 rp = project_lit(rc.P, pp);
    const float2 tp = project_lit(tc.P, pp);
    const float2 rpB = (dev & DEV_EXACT_BORDER) ? make_float2(x, y) : rp; // the border test
    if(dev & DEV_EXACT_PIXEL)
        rp = make_float2(x, y);                                           // the centre fetch
    const float dd = (float)wsh + 2.0f;
    if((rpB.x < dd) || (rpB.x > (L.rcW - 1.0f) - dd) || (tp.x < dd) || (tp.x > (L.tcW - 1.0f) - dd) || (rpB.y < dd) || (rpB.y > (L.rcH - 1.0f) - dd) ||
       (tp.y < dd) || (tp.y > (L.tcH - 1.0f) - dd))
        return INFINITY;
    const float rcIW = 1.f / L.rcW, rcIH = 1.f / L.rcH, tcIW = 1.f / L.tcW, tcIH = 1.f / L.tcH;
    const float4 rcCenter = rt.fetch((rp.x + 0.5f) * rcIW, (rp.y + 0.5f) * rcIH);
    const float4 tcCenter = tt.fetch((tp.x + 0.5f) * tcIW, (tp.y + 0.5f) * tcIH);
    if(rcCenter.w < (255.f * 0.9f) || tcCenter.w < (255.f * 0.4f))
        return INFINITY;
    // DEV_HOMOGENEOUS: P (p + a x d + b y d) = h0 + a (M x d) + b (M y d)
    [[maybe_unused]] f3 hr0, ht0, rax, ray, tax, tay;
    if(DEVS)
    {
        hr0 = M3x4mulV3(rc.P, Rp.p), ht0 = M3x4mulV3(tc.P, T.p);
        rax = M3x3mulV3(rc.P, Rp.x * Rp.d), ray = M3x3mulV3(rc.P, Rp.y * Rp.d);
        tax = M3x3mulV3(tc.P, T.x * T.d), tay = M3x3mulV3(tc.P, T.y * T.d);
    }
    const float log2e = 1.44269504088896340736f;
    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
    const int n = 2 * wsh + 1;
#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
#pragma unroll UNROLL
        for(int xp = -wsh; xp <= wsh; ++xp)
        {
            float2 rpc, tpc;
            if(DEVS && (dev & DEV_HOMOGENEOUS))
            {
                const float fx = (float)xp, fy = (float)yp;
                const float hrz = fmaf(fx, rax.z, fmaf(fy, ray.z, hr0.z)), htz = fmaf(fx, tax.z, fmaf(fy, tay.z, ht0.z));
                const float ir = fast_rcp(hrz), it = fast_rcp(htz);
                rpc = make_float2(fmaf(fx, rax.x, fmaf(fy, ray.x, hr0.x)) * ir, fmaf(fx, rax.y, fmaf(fy, ray.y, hr0.y)) * ir);
                tpc = make_float2(fmaf(fx, tax.x, fmaf(fy, tay.x, ht0.x)) * it, fmaf(fx, tax.y, fmaf(fy, tay.y, ht0.y)) * it);
            }
            else
            {
                const f3 pT = (T.p + T.x * (T.d * (float)xp)) + T.y * (T.d * (float)yp);
                const f3 pR = (Rp.p + Rp.x * (Rp.d * (float)xp)) + Rp.y * (Rp.d * (float)yp);
                rpc = project_lit(rc.P, pR);
                tpc = project_lit(tc.P, pT);
            }
            const float4 rcC = rt.fetch((rpc.x + 0.5f) * rcIW, (rpc.y + 0.5f) * rcIH);
            const float4 tcC = tt.fetch((tpc.x + 0.5f) * tcIW, (tpc.y + 0.5f) * tcIH);
            float w;
            if(DEVS && (dev & DEV_MERGED_EXP))
            {
                // exp(-(dCr / gC + dP / gP)) exp(-(dCt / gC + dP / gP)) = exp2((dCr + dCt) (-log2e / gC) - 2 dP log2e / gP)
                const float drx = rcCenter.x - rcC.x, dry = rcCenter.y - rcC.y, drz = rcCenter.z - rcC.z;
                const float dtx = tcCenter.x - tcC.x, dty = tcCenter.y - tcC.y, dtz = tcCenter.z - tcC.z;
                const float dcr = __builtin_amdgcn_sqrtf(fmaf(drx, drx, fmaf(dry, dry, drz * drz)));
                const float dct = __builtin_amdgcn_sqrtf(fmaf(dtx, dtx, fmaf(dty, dty, dtz * dtz)));
                const float tabv = 2.0f * sqrtf((float)(xp * xp + yp * yp)) * L.invGammaP * log2e;
                w = __builtin_amdgcn_exp2f(fmaf(dcr + dct, -L.invGammaC * log2e, -tabv));
            }
            else if(dPtab != nullptr)
            {
                const float dP = dPtab[(yp + wsh) * n + (xp + wsh)];
                const float wr = cost_yk_tab(dP, rcCenter, rcC, L.invGammaC, expTab);
                const float wt = cost_yk_tab(dP, tcCenter, tcC, L.invGammaC, expTab);
                w = wr * wt;
            }
            else
            {
                const float wr = cost_yk(xp, yp, rcCenter, rcC, L.invGammaC, L.invGammaP, expTab);
                const float wt = cost_yk(xp, yp, tcCenter, tcC, L.invGammaC, L.invGammaP, expTab);
                w = wr * wt;
            }
            if(DEVS && (dev & DEV_SHIFTED_SUMS))
            {
                const float gx = rcCenter.x - rcC.x, gy = tcCenter.x - tcC.x;
                const float wgx = w * gx, wgy = w * gy;
                wsum += w;
                xsum += wgx;
                ysum += wgy;
                xxsum = fmaf(wgx, gx, xxsum);
                yysum = fmaf(wgy, gy, yysum);
                xysum = fmaf(wgx, gy, xysum);
            }
            else
            {
                const float gx = rcC.x, gy = tcC.x; // simStat::update
                wsum += w;
                xsum += w * gx;
                ysum += w * gy;
                xxsum += w * gx * gx;
                yysum += w * gy * gy;
                xysum += w * gx * gy;
            }
        }
    float rawSim;
    if(DEVS && (dev & DEV_SHIFTED_SUMS))
    {
        const float iw = fast_rcp(wsum);
        const float varXW = (xxsum - xsum * xsum * iw) * iw;
        const float varYW = (yysum - ysum * ysum * iw) * iw;
        const float varXYW = (xysum - xsum * ysum * iw) * iw;
        rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
    }
    else
    {
        // simStat::computeWSim
        const float varXW = (xxsum - xsum * xsum / wsum) / wsum;
        const float varYW = (yysum - ysum * ysum / wsum) / wsum;
        const float varXYW = (xysum - xsum * ysum / wsum) / wsum;
        rawSim = varXYW / sqrtf(varXW * varYW);
    }
    const float sim = isfinite(rawSim) ? -rawSim : 1.0f;
    if(TInvert)
        return sigmoid_lit(0.0f, 1.0f, 0.7f, -0.7f, sim, expTab);
    return sim;
}

// computeRotCSEpip (Patch.cuh:111-135), with the SGM normal when given (kernels.cuh:316-333)
__device__ __forceinline__ void patch_axes_lit(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 p, const float* nn, f3& ax, f3& ay)
{
    const f3 v1 = normalize_lit(ld3(rc.C) - p);
    const f3 v2 = normalize_lit(ld3(tc.C) - p);
    ay = normalize_lit(cross(v1, v2));
    const f3 s = v1 + v2;
    const f3 n = nn != nullptr ? f3{nn[0], nn[1], nn[2]} : normalize_lit(f3{s.x / 2.0f, s.y / 2.0f, s.z / 2.0f});
    ax = normalize_lit(cross(ay, n));
}
// volume_computePatch (kernels.cuh:26-35): the patch of pixel (x, y) on the fronto-parallel plane at `depthPlane`
__device__ __forceinline__ LitPatch sgm_patch_lit(const avdm_camera_t& rc, const avdm_camera_t& tc, float x, float y, float depthPlane)
{
    const f3 C = ld3(rc.C), Z = ld3(rc.ZVect);
    const f3 planep = C + Z * depthPlane;
    const f3 v = normalize_lit(M3x3mulV2(rc.iP, x, y));
    LitPatch q;
    q.p = linePlaneIntersect(C, v, planep, Z);
    q.d = pix_size_lit(rc, q.p);
    patch_axes_lit(rc, tc, q.p, nullptr, q.x, q.y);
    return q;
}
// kernels.cuh:285-333: the patch of pixel (x, y) on Refine plane z around the SGM depth
__device__ __forceinline__ LitPatch refine_patch_lit(const avdm_camera_t& rc, const avdm_camera_t& tc, float x, float y, float2 dps, int rel, const float* nn)
{
    const f3 C = ld3(rc.C);
    LitPatch q;
    q.p = C + normalize_lit(M3x3mulV2(rc.iP, x, y)) * dps.x; // get3DPointForPixelAndDepthFromRC
    if(rel != 0)
        q.p = q.p + normalize_lit(q.p - C) * ((float)rel * dps.y); // move3DPointByRcPixSize (kernels.cuh:17-24)
    q.d = pix_size_lit(rc, q.p);
    patch_axes_lit(rc, tc, q.p, nn, q.x, q.y);
    return q;
}
// kernels.cuh:197-231: similarity -> the uint8 scale, then best / second best
__device__ __forceinline__ float to_u8_scale(float fsim)
{
    if(fsim == INFINITY)
        return 255.0f;
    fsim = (fsim - (-1.0f)) * (1.0f / (1.0f - (-1.0f)));
    fsim = fminf(1.0f, fmaxf(0.0f, fsim));
    return fsim * 254.0f;
}
__device__ __forceinline__ void commit_best_second(unsigned& wb, unsigned& ws, int k, float fsim)
{
    const unsigned sh8 = 8u * k;
    const unsigned b1 = (wb >> sh8) & 0xffu, b2 = (ws >> sh8) & 0xffu;
    if(fsim < (float)b1)
    {
        ws = (ws & ~(0xffu << sh8)) | (b1 << sh8);
        wb = (wb & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
    }
    else if(fsim < (float)b2)
        ws = (ws & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
}

// ==========================================================================================================================================
// the attribution kernels (AVDM_SIM_LITERAL=1): one lane per pixel, global-memory taps, AVDM_SIM_LITERAL_DEV switches
// ==========================================================================================================================================
struct LitTex
{
    Tex rcT, tcT;
    float mipmapLevel;
};

__global__ void __launch_bounds__(256)
  literal_similarity_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, long long pitch_y, int pitch_x, const float* __restrict__ depths,
                            avdm_camera_t rc, avdm_camera_t tc, LitArgs L, LitTex X, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    const unsigned vx = blockIdx.x * 64 + (threadIdx.x & 63), vy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const unsigned z0 = ((zBegin >> 2) + blockIdx.z) << 2;
    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    uint8_t* const pb = best + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    uint8_t* const ps = second + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    unsigned wb = *reinterpret_cast<const unsigned*>(pb), ws = *reinterpret_cast<const unsigned*>(ps);
    TexTap rt{X.rcT, X.mipmapLevel}, tt{X.tcT, X.mipmapLevel};
#pragma unroll 1
    for(int k = 0; k < 4; ++k)
    {
        const unsigned vz = z0 + k;
        if(vz < zBegin || vz >= zEnd)
            continue;
        const LitPatch T = sgm_patch_lit(rc, tc, x, y, depths[vz]);
        // DEV_SHARED_R: the R side from plane 1 of the aligned group of four (clamped to the T camera's range), like the four-plane pass
        const unsigned zr = min(max(z0 + 1u, zBegin), zEnd - 1u);
        const LitPatch Rp = (L.dev & DEV_SHARED_R) ? sgm_patch_lit(rc, tc, x, y, depths[zr]) : T;
        commit_best_second(wb, ws, k, to_u8_scale(ncc_literal<false, true, 1, 0>(rc, tc, L, T, Rp, x, y, rt, tt, nullptr, g_exp2f_tab)));
    }
    *reinterpret_cast<unsigned*>(pb) = wb;
    *reinterpret_cast<unsigned*>(ps) = ws;
}

__global__ void __launch_bounds__(256)
  literal_refine_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize, int map_pitch,
                        const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, LitArgs L, LitTex X, int stepXY,
                        unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    const unsigned vx = blockIdx.x * 64 + (threadIdx.x & 63), vy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const float2 dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    if(dps.x <= 0.0f) // kernels.cuh:266-270
        return;
    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const float* nn = sgmNormal != nullptr ? (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx : nullptr;
    __half* const pv = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2;
    TexTap rt{X.rcT, X.mipmapLevel}, tt{X.tcT, X.mipmapLevel};
#pragma unroll 1
    for(unsigned vz = zBegin; vz < zEnd; ++vz)
    {
        const LitPatch T = refine_patch_lit(rc, tc, x, y, dps, (int)vz - ((volDimZ - 1) / 2), nn);
        const unsigned zr = min(max((vz & ~3u) + 1u, zBegin), zEnd - 1u);
        const LitPatch Rp = (L.dev & DEV_SHARED_R) ? refine_patch_lit(rc, tc, x, y, dps, (int)zr - ((volDimZ - 1) / 2), nn) : T;
        const float fsim = ncc_literal<true, true, 1, 0>(rc, tc, L, T, Rp, x, y, rt, tt, nullptr, g_exp2f_tab);
        if(fsim == INFINITY)
            continue;
        pv[vz] = __float2half(__half2float(pv[vz]) + fsim);
    }
}

// ==========================================================================================================================================
// the reference-arithmetic mode of the product: LDS windows as a cache
// ==========================================================================================================================================
struct StrictArgs
{
    TexLevel rcL, tcL; // the integral level both images are sampled at
    int rcap, tcap;    // LDS capacities in texels: R tile, T window
    int rpitch;        // row pitch of the R tile (texels)
    float dP[81];      // sqrtf(xp^2 + yp^2) * invGammaP, row-major over (yp, xp)
};

// LDS row pitch (texels) for a window of w texels: smallest value = 8 (mod 16) that is >= w (the 8 x 8-pixel gathers of a wave at the
// minimum bank-conflict degree, as in avdm_similarity.hip)
__host__ __device__ __forceinline__ int strict_pitch_for(int w) { return (((w + 7) >> 4) << 4) + 8; }

__device__ __forceinline__ float wave_max_f32_(float v)
{
    v = fmaxf(v, dpp_f32<0x111>(v, v));
    v = fmaxf(v, dpp_f32<0x112>(v, v));
    v = fmaxf(v, dpp_f32<0x114>(v, v));
    v = fmaxf(v, dpp_f32<0x118>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f32<0x143, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// cooperative copy of the window [x0, x0 + w) x [y0, y0 + h) of level L (inside the image) into LDS, coalesced rows
__device__ __forceinline__ void stage_plain(uint2* dst, int pitch, const TexLevel& L, int x0, int y0, int w, int h)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int r = wave; r < h; r += 4)
    {
        const uint2* src = L.base + (long long)(y0 + r) * L.pitch8 + x0;
        uint2* d = dst + r * pitch;
        for(int c = lane; c < w; c += 64)
            d[c] = src[c];
    }
}

struct StrictWindows
{
    int rx0, ry0, rw, rh, rpitch; // rw = 0: no R tile
    int tx0, ty0, tw, th, tpitch; // tw = 0: no T window
};

// The two windows of a workgroup (16 x 16 stage pixels x a block of planes), staged.  They only decide where a tap's texels are READ from:
//   R: the footprint of the workgroup's pixels + the patch halo (the margin of the border test, wsh + 2 pixels);
//   T: the hull of the lanes' projected patch corners on the first and the last plane of the block (a point moving along a ray projects to a
//      monotone path in T: the planes in between lie inside), one texel of slack, clipped to the image; a hull larger than the LDS budget is
//      cut down around its centre.
// patch_of(z, ok): the lane's patch on plane z (ok = false: none).  sbox: 4 shared ints.
template <class PatchOf>
__device__ __forceinline__ StrictWindows strict_stage_windows(uint2* smem, int* sbox, const StrictArgs& A, const LitArgs& L, const avdm_camera_t& tc, int stepXY,
                                                              avdm_roi_t roi, bool active, unsigned zFirst, unsigned zLast, PatchOf patch_of)
{
    StrictWindows W;
    // ---- R tile ----
    {
        const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
        const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
        const float pxMin = (float)((int)roi.x.begin + bx) * (float)stepXY, pxMax = (float)((int)roi.x.begin + min(bx + 15, roiW - 1)) * (float)stepXY;
        const float pyMin = (float)((int)roi.y.begin + by) * (float)stepXY, pyMax = (float)((int)roi.y.begin + min(by + 15, roiH - 1)) * (float)stepXY;
        const float m = (float)L.wsh + 2.0f;
        const float sx = (float)A.rcL.W / L.rcW, sy = (float)A.rcL.H / L.rcH; // pixel of the nominal level -> texel of the actual one
        int x0 = (int)floorf((pxMin - m + 0.5f) * sx - 0.5f) - 1, x1 = (int)floorf((pxMax + m + 0.5f) * sx - 0.5f) + 2;
        int y0 = (int)floorf((pyMin - m + 0.5f) * sy - 0.5f) - 1, y1 = (int)floorf((pyMax + m + 0.5f) * sy - 0.5f) + 2;
        x0 = max(x0, 0), y0 = max(y0, 0), x1 = min(x1, A.rcL.W - 1), y1 = min(y1, A.rcL.H - 1);
        W.rx0 = x0, W.ry0 = y0, W.rw = x1 - x0 + 1, W.rh = y1 - y0 + 1, W.rpitch = A.rpitch;
        if(W.rw < 2 || W.rh < 2 || W.rw > W.rpitch || W.rpitch * W.rh > A.rcap)
            W.rw = W.rh = 0;
        else
            stage_plain(smem, W.rpitch, A.rcL, W.rx0, W.ry0, W.rw, W.rh);
    }
    // ---- T window: the hull of the lanes' patch corners on the two extreme planes ----
    if(threadIdx.x < 4)
        sbox[threadIdx.x] = (threadIdx.x & 2) ? INT_MIN : INT_MAX;
    __syncthreads();
    float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
    if(active)
    {
        const float sx = (float)A.tcL.W / L.tcW, sy = (float)A.tcL.H / L.tcH;
#pragma unroll 1
        for(int e = 0; e < 2; ++e)
        {
            const unsigned vz = e == 0 ? zFirst : zLast;
            if(e == 1 && zLast == zFirst)
                break;
            bool ok = false;
            const LitPatch q = patch_of(vz, ok);
            if(!ok)
                continue;
#pragma unroll
            for(int c = 0; c < 4; ++c)
            {
                const float cx = (float)((c & 1) ? L.wsh : -L.wsh), cy = (float)((c & 2) ? L.wsh : -L.wsh);
                const f3 pT = (q.p + q.x * (q.d * cx)) + q.y * (q.d * cy);
                const float2 tp = project_lit(tc.P, pT);
                const float X = (tp.x + 0.5f) * sx - 0.5f, Y = (tp.y + 0.5f) * sy - 0.5f;
                if(fabsf(X) < 1.0e8f && fabsf(Y) < 1.0e8f) // (also refuses NaN)
                {
                    bx0 = fminf(bx0, X), bx1 = fmaxf(bx1, X);
                    by0 = fminf(by0, Y), by1 = fmaxf(by1, Y);
                }
            }
        }
    }
    {
        const float mnx = wave_min_f32(bx0), mny = wave_min_f32(by0), mxx = wave_max_f32_(bx1), mxy = wave_max_f32_(by1);
        if((threadIdx.x & 63) == 0 && mnx <= mxx && mny <= mxy)
        {
            atomicMin(&sbox[0], (int)floorf(mnx));
            atomicMin(&sbox[1], (int)floorf(mny));
            atomicMax(&sbox[2], (int)floorf(mxx));
            atomicMax(&sbox[3], (int)floorf(mxy));
        }
    }
    __syncthreads();
    {
        int x0 = max(sbox[0] - 1, 0), y0 = max(sbox[1] - 1, 0), x1 = min(sbox[2] + 2, A.tcL.W - 1), y1 = min(sbox[3] + 2, A.tcL.H - 1);
        W.tx0 = W.ty0 = W.tw = W.th = W.tpitch = 0;
        if(sbox[0] <= sbox[2] && x1 - x0 >= 1 && y1 - y0 >= 1)
        {
            // larger than the budget: cut down around the centre (uniform; the taps outside read global memory)
            while(strict_pitch_for(x1 - x0 + 1) * (y1 - y0 + 1) > A.tcap && (x1 - x0 > 8 || y1 - y0 > 8))
            {
                if(x1 - x0 >= y1 - y0)
                    x0 += 1, x1 -= 1;
                else
                    y0 += 1, y1 -= 1;
            }
            if(strict_pitch_for(x1 - x0 + 1) * (y1 - y0 + 1) <= A.tcap)
            {
                W.tx0 = x0, W.ty0 = y0, W.tw = x1 - x0 + 1, W.th = y1 - y0 + 1, W.tpitch = strict_pitch_for(W.tw);
                stage_plain(smem + A.rcap, W.tpitch, A.tcL, W.tx0, W.ty0, W.tw, W.th);
            }
        }
    }
    __syncthreads();
    return W;
}

__device__ __forceinline__ void strict_pixel_of_lane(int& tx, int& ty)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    tx = (w & 1) * 8 + (lane & 7);
    ty = (w >> 1) * 8 + (lane >> 3);
}

constexpr unsigned kStrictSgmChunksPerWg = 4; // 16 planes per workgroup share the two windows

// one plane of one lane: the branch-free window taps, and once more with the checked taps when any lane of the wave had a tap outside its window
template <bool TInvert, bool FIXED8, int UNROLL, int WSH>
__device__ __forceinline__ float strict_plane(const avdm_camera_t& rc, const avdm_camera_t& tc, const LitArgs& L, const StrictArgs& A, const StrictWindows& W,
                                              const uint2* smem, const uint64_t* expTab, const LitPatch& T, float x, float y, bool active)
{
    float fsim = INFINITY;
    bool miss = false;
    const bool haveWindows = W.rw > 0 && W.tw > 0; // uniform
    if(haveWindows)
    {
        if(active)
        {
            WinTap<FIXED8, false> rt{A.rcL, smem, W.rpitch, W.rx0, W.ry0, (unsigned)(W.rw - 1), (unsigned)(W.rh - 1), false};
            WinTap<FIXED8, false> tt{A.tcL, smem + A.rcap, W.tpitch, W.tx0, W.ty0, (unsigned)(W.tw - 1), (unsigned)(W.th - 1), false};
            fsim = ncc_literal<TInvert, false, UNROLL, WSH>(rc, tc, L, T, T, x, y, rt, tt, A.dP, expTab);
            miss = rt.missed() || tt.missed();
        }
        if(!__any(miss)) // wave-uniform
            return fsim;
    }
    if(active)
    {
        WinTap<FIXED8, true> rt{A.rcL, smem, W.rpitch, W.rx0, W.ry0, W.rw > 0 ? (unsigned)(W.rw - 1) : 0u, W.rw > 0 ? (unsigned)(W.rh - 1) : 0u, false};
        WinTap<FIXED8, true> tt{A.tcL, smem + A.rcap, W.tpitch, W.tx0, W.ty0, W.tw > 0 ? (unsigned)(W.tw - 1) : 0u, W.tw > 0 ? (unsigned)(W.th - 1) : 0u, false};
        fsim = ncc_literal<TInvert, false, 1, 0>(rc, tc, L, T, T, x, y, rt, tt, A.dP, expTab);
    }
    return fsim;
}

template <bool FIXED8, int WSH>
__global__ void __launch_bounds__(256, 3)
  strict_sgm_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, long long pitch_y, int pitch_x, const float* __restrict__ depths, avdm_camera_t rc,
                    avdm_camera_t tc, LitArgs L, StrictArgs A, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    extern __shared__ __attribute__((aligned(16))) uint2 smem[];
    __shared__ int sbox[4];
    __shared__ uint64_t sExp[32];
    if(threadIdx.x < 32)
        sExp[threadIdx.x] = g_exp2f_tab[threadIdx.x]; // (visible after the barriers of strict_stage_windows)
    int tx, ty;
    strict_pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    const bool inRoi = vx < roi.x.end - roi.x.begin && vy < roi.y.end - roi.y.begin;
    const unsigned z0 = ((zBegin >> 2) + blockIdx.z * kStrictSgmChunksPerWg) << 2;
    const unsigned za = max(z0, zBegin), zbEnd = min(z0 + 4u * kStrictSgmChunksPerWg, zEnd);
    if(za >= zbEnd) // uniform
        return;
    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const StrictWindows W = strict_stage_windows(smem, sbox, A, L, tc, stepXY, roi, inRoi, za, zbEnd - 1u, [&](unsigned vz, bool& ok) {
        ok = true;
        return sgm_patch_lit(rc, tc, x, y, depths[vz]);
    });
    uint8_t* const pb0 = best + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    uint8_t* const ps0 = second + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
#pragma unroll 1
    for(unsigned c = 0; c < kStrictSgmChunksPerWg; ++c)
    {
        const unsigned zc = z0 + 4u * c;
        if(zc >= zEnd) // uniform
            break;
        if(zc + 4u <= zBegin) // uniform
            continue;
        unsigned wb = 0, ws = 0;
        if(inRoi)
        {
            wb = *reinterpret_cast<const unsigned*>(pb0 + 4u * c);
            ws = *reinterpret_cast<const unsigned*>(ps0 + 4u * c);
        }
#pragma unroll 1
        for(int k = 0; k < 4; ++k)
        {
            const unsigned vz = zc + k;
            if(vz < zBegin || vz >= zEnd) // uniform
                continue;
            LitPatch T;
            if(inRoi)
                T = sgm_patch_lit(rc, tc, x, y, depths[vz]);
            const float fsim = strict_plane<false, FIXED8, WSH == 4 ? 3 : 1, WSH>(rc, tc, L, A, W, smem, sExp, T, x, y, inRoi);
            commit_best_second(wb, ws, k, to_u8_scale(fsim));
        }
        if(inRoi)
        {
            *reinterpret_cast<unsigned*>(pb0 + 4u * c) = wb;
            *reinterpret_cast<unsigned*>(ps0 + 4u * c) = ws;
        }
    }
}

template <bool FIXED8, int WSH>
__global__ void __launch_bounds__(256, 3)
  strict_refine_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize, int map_pitch,
                       const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, LitArgs L, StrictArgs A, int stepXY,
                       unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    extern __shared__ __attribute__((aligned(16))) uint2 smem[];
    __shared__ int sbox[4];
    __shared__ uint64_t sExp[32];
    if(threadIdx.x < 32)
        sExp[threadIdx.x] = g_exp2f_tab[threadIdx.x]; // (visible after the barriers of strict_stage_windows)
    int tx, ty;
    strict_pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    const bool inRoi = vx < roi.x.end - roi.x.begin && vy < roi.y.end - roi.y.begin;
    float2 dps = make_float2(-1.f, 0.f);
    if(inRoi)
        dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    const bool active = inRoi && !(dps.x <= 0.0f); // kernels.cuh:266-270
    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const float* nn = sgmNormal != nullptr && inRoi ? (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx : nullptr;
    const int mid = (volDimZ - 1) / 2;
    const StrictWindows W = strict_stage_windows(smem, sbox, A, L, tc, stepXY, roi, active, zBegin, zEnd - 1u, [&](unsigned vz, bool& ok) {
        ok = true;
        return refine_patch_lit(rc, tc, x, y, dps, (int)vz - mid, nn);
    });
    __half* const pv = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2;
#pragma unroll 1
    for(unsigned vz = zBegin; vz < zEnd; ++vz)
    {
        LitPatch T;
        if(active)
            T = refine_patch_lit(rc, tc, x, y, dps, (int)vz - mid, nn);
        const float fsim = strict_plane<true, FIXED8, 1, WSH>(rc, tc, L, A, W, smem, sExp, T, x, y, active);
        if(active && fsim != INFINITY)
            pv[vz] = __float2half(__half2float(pv[vz]) + fsim);
    }
}

static LitArgs make_args(const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale, int wsh, double gammaC, double gammaP, bool devs)
{
    LitArgs L;
    L.rcW = (float)tex_dim_w(rcPyr, scale);
    L.rcH = (float)tex_dim_h(rcPyr, scale);
    L.tcW = (float)tex_dim_w(tcPyr, scale);
    L.tcH = (float)tex_dim_h(tcPyr, scale);
    L.invGammaC = 1.f / (float)gammaC;
    L.invGammaP = 1.f / (float)gammaP;
    L.wsh = wsh;
    const char* e = devs ? getenv("AVDM_SIM_LITERAL_DEV") : nullptr;
    L.dev = e != nullptr ? atoi(e) : 0;
    return L;
}
static LitTex make_tex_args(const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale)
{
    LitTex X;
    X.rcT = make_tex(rcPyr);
    X.tcT = make_tex(tcPyr);
    X.mipmapLevel = tex_level_of(rcPyr, scale);
    return X;
}

// the LDS budget of the strict kernels: a third of the compute unit's 160 KiB in allocation granules of 1280 B (three workgroups per compute
// unit, like the general instantiations of avdm_similarity.hip), minus the static shared state
constexpr int kStrictLdsBytes = 42 * 1280 - 512;

// false: the windows cannot serve this call (a fractional level of detail, a patch too large for the budget): the caller runs the
// one-lane-per-pixel kernels, which evaluate the same arithmetic with every tap from global memory
static bool make_strict_args(StrictArgs& A, const LitArgs& L, const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale, int stepXY)
{
    int rl = 0, tl = 0;
    if(!lod_is_integral(rcPyr, scale, &rl) || !lod_is_integral(tcPyr, scale, &tl) || rl != tl)
        return false;
    // tex2DLod samples BOTH images at the R image's level (Patch.cuh:499-505); with pyramids of one layout that is the T image's own level
    if(rcPyr->levels != tcPyr->levels || rcPyr->min_downscale != tcPyr->min_downscale)
        return false;
    A.rcL = make_tex(rcPyr).lv[rl];
    A.tcL = make_tex(tcPyr).lv[tl];
    const int rw = 15 * stepXY + 2 * (L.wsh + 2) + 5;
    A.rpitch = strict_pitch_for(rw);
    A.rcap = A.rpitch * rw;
    A.tcap = kStrictLdsBytes / 8 - A.rcap;
    if(A.tcap < A.rcap)
        return false;
    const int n = 2 * L.wsh + 1;
    for(int i = 0; i < 81; ++i)
        A.dP[i] = 0.0f;
    for(int yp = -L.wsh; yp <= L.wsh; ++yp)
        for(int xp = -L.wsh; xp <= L.wsh; ++xp)
        {
            volatile float deltaP = sqrtf((float)(xp * xp + yp * yp)); // CostYKfromLab (color.cuh:167-210): one correctly rounded root, one fp32 product
            deltaP = deltaP * L.invGammaP;
            A.dP[(yp + L.wsh) * n + (xp + L.wsh)] = deltaP;
        }
    return true;
}

} // namespace

// AVDM_SIM_LITERAL=1: the attribution kernels
int literal_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                               const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_sgm_params_t* sp, avdm_range_t dr,
                               avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, sp->scale, sp->wsh, sp->gammaC, sp->gammaP, true);
    const LitTex X = make_tex_args(rc_pyr, tc_pyr, sp->scale);
    const unsigned nchunks = ((dr.end + 3) >> 2) - (dr.begin >> 2);
    hipLaunchKernelGGL(literal_similarity_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), nchunks), dim3(256), 0,
                       (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, L, X, sp->stepXY, dr.begin, dr.end, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity(literal)");
}

int literal_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal,
                              int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                              const avdm_refine_params_t* rp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, rp->scale, rp->wsh, rp->gammaC, rp->gammaP, true);
    const LitTex X = make_tex_args(rc_pyr, tc_pyr, rp->scale);
    hipLaunchKernelGGL(literal_refine_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), 1), dim3(256), 0, (hipStream_t)stream,
                       (__half*)vol_f16, pitch_y, pitch_x, dimZ, (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, L, X, rp->stepXY,
                       dr.begin, dr.end, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity(literal)");
}

// avdm_sgm_params_t::referenceArithmetic / avdm_refine_params_t::referenceArithmetic: the product's reference-arithmetic mode
// (AVDM_STRICT_WINDOWS=0, read at each call: every tap from global memory — the A/B that shows the windows change no bit)
static bool strict_windows_enabled()
{
    const char* e = getenv("AVDM_STRICT_WINDOWS");
    return !(e != nullptr && e[0] == '0');
}

int strict_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                              const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_sgm_params_t* sp, avdm_range_t dr,
                              avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, sp->scale, sp->wsh, sp->gammaC, sp->gammaP, false);
    StrictArgs A;
    const unsigned nchunks = ((dr.end + 3) >> 2) - (dr.begin >> 2);
    if(!strict_windows_enabled() || !make_strict_args(A, L, rc_pyr, tc_pyr, sp->scale, sp->stepXY))
    {
        const LitTex X = make_tex_args(rc_pyr, tc_pyr, sp->scale);
        hipLaunchKernelGGL(literal_similarity_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), nchunks), dim3(256), 0,
                           (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, L, X, sp->stepXY, dr.begin, dr.end, roi);
        AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity(reference arithmetic, global taps)");
    }
    const dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), divUp(nchunks, kStrictSgmChunksPerWg));
    const size_t lds = (size_t)(A.rcap + A.tcap) * sizeof(uint2);
#define AVDM_STRICT_SGM(F8, W)                                                                                                                            \
    hipLaunchKernelGGL((strict_sgm_kernel<F8, W>), grid, dim3(256), lds, (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, L, A,     \
                       sp->stepXY, dr.begin, dr.end, roi)
    const bool f8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
    if(f8 && sp->wsh == 4)
        AVDM_STRICT_SGM(true, 4); // the default patch: its sample loop unrolled
    else if(f8)
        AVDM_STRICT_SGM(true, 0);
    else
        AVDM_STRICT_SGM(false, 0);
#undef AVDM_STRICT_SGM
    AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity(reference arithmetic)");
}

int strict_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal,
                             int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                             const avdm_refine_params_t* rp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, rp->scale, rp->wsh, rp->gammaC, rp->gammaP, false);
    StrictArgs A;
    if(!strict_windows_enabled() || !make_strict_args(A, L, rc_pyr, tc_pyr, rp->scale, rp->stepXY))
    {
        const LitTex X = make_tex_args(rc_pyr, tc_pyr, rp->scale);
        hipLaunchKernelGGL(literal_refine_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), 1), dim3(256), 0,
                           (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x, dimZ, (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc,
                           *tc, L, X, rp->stepXY, dr.begin, dr.end, roi);
        AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity(reference arithmetic, global taps)");
    }
    const dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), 1);
    const size_t lds = (size_t)(A.rcap + A.tcap) * sizeof(uint2);
#define AVDM_STRICT_REFINE(F8, W)                                                                                                                         \
    hipLaunchKernelGGL((strict_refine_kernel<F8, W>), grid, dim3(256), lds, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x, dimZ,               \
                       (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, L, A, rp->stepXY, dr.begin, dr.end, roi)
    const bool f8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
    if(f8 && rp->wsh == 3)
        AVDM_STRICT_REFINE(true, 3);
    else if(f8)
        AVDM_STRICT_REFINE(true, 0);
    else
        AVDM_STRICT_REFINE(false, 0);
#undef AVDM_STRICT_REFINE
    AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity(reference arithmetic)");
}

} // namespace avdm
