// avdm_literal.hip — AVDM_SIM_LITERAL=1: the reference's similarity arithmetic AS WRITTEN, on the GPU.  Compiled with -ffp-contract=off.
//
// NOT a product path (one lane per pixel, every tap through the software texture unit from global memory, ~10 x slower than the default
// kernels).  It exists so that the distance between the default kernels (avdm_similarity.hip: shifted NCC sums, exact R pixel in the
// border test, homogeneous patch projection) and the reference's own code compiled for the CPU (oracle/_ref) can be ATTRIBUTED by
// measurement: this form differs from the reference on the CPU only through the device library's expf, whereas the default kernels
// differ from both by the conditioning of the fp32 sums (tests/test_gpu_parity.py::test_literal_mode_*).
//
//   compNCCby3DptsYK        Patch.cuh:466-572      every patch sample as a 3-D point projected into both cameras; the border test and
//                                                  the centre colour on the RE-PROJECTED R pixel
//   simStat                 SimStat.cuh:72-153     the six UNSHIFTED fp32 sums of w, w L, w L^2 (L ~ 0 ... 255): variance = difference
//                                                  of numbers ~ 5e6
//   CostYKfromLab           color.cuh:167-210      two Yoon-Kweon weights, two exponentials, multiplied
//   volume_computeSimilarity_kernel / volume_refineSimilarity_kernel   deviceSimilarityVolumeKernels.cuh:109-233, 235-391
#include "avdm_device.h"

#include <math.h>
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
// color.cuh:167-210
__device__ __forceinline__ float cost_yk(int dx, int dy, float4 c1, float4 c2, float invGammaC, float invGammaP)
{
    const float ex = c1.x - c2.x, ey = c1.y - c2.y, ez = c1.z - c2.z;
    float deltaC = sqrtf(ex * ex + ey * ey + ez * ez);
    deltaC *= invGammaC;
    float deltaP = sqrtf((float)(dx * dx + dy * dy));
    deltaP *= invGammaP;
    deltaC += deltaP;
    return expf(-deltaC);
}

// AVDM_SIM_LITERAL_DEV=<bit mask> (read at each call): the deviations of the default kernels (avdm_similarity.hip) from the reference's
// arithmetic, introduced into THIS literal evaluation one at a time, so that the distance default <-> reference can be attributed deviation by
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
    Tex rcT, tcT;
    float rcW, rcH, tcW, tcH; // nominal level dimensions (DeviceMipmapImage::getDimensions)
    float mipmapLevel;
    float invGammaC, invGammaP;
    int wsh;
    int dev; // DEV_* bits
};

struct LitPatch
{
    f3 p, x, y;
    float d;
};

// Patch.cuh:466-572 + SimStat.cuh; INFINITY when the patch is invalid.  T = the plane's own patch; Rp = the patch the R side is sampled
// from (the same one unless DEV_SHARED_R); (x, y) = the lane's pixel (DEV_EXACT_PIXEL).
template <bool TInvert>
__device__ float ncc_literal(const avdm_camera_t& rc, const avdm_camera_t& tc, const LitArgs& L, const LitPatch& T, const LitPatch& Rp, float x, float y)
{
    const int dev = L.dev;
    const f3 pp = T.p;
    float2 rp = project_lit(rc.P, pp);
    const float2 tp = project_lit(tc.P, pp);
    const float2 rpB = (dev & DEV_EXACT_BORDER) ? make_float2(x, y) : rp; // the border test
    if(dev & DEV_EXACT_PIXEL)
        rp = make_float2(x, y);                                           // the centre fetch
    const float dd = (float)L.wsh + 2.0f;
    if((rpB.x < dd) || (rpB.x > (L.rcW - 1.0f) - dd) || (tp.x < dd) || (tp.x > (L.tcW - 1.0f) - dd) || (rpB.y < dd) || (rpB.y > (L.rcH - 1.0f) - dd) ||
       (tp.y < dd) || (tp.y > (L.tcH - 1.0f) - dd))
        return INFINITY;
    const float rcIW = 1.f / L.rcW, rcIH = 1.f / L.rcH, tcIW = 1.f / L.tcW, tcIH = 1.f / L.tcH;
    const float4 rcCenter = tex2DLod(L.rcT, (rp.x + 0.5f) * rcIW, (rp.y + 0.5f) * rcIH, L.mipmapLevel);
    const float4 tcCenter = tex2DLod(L.tcT, (tp.x + 0.5f) * tcIW, (tp.y + 0.5f) * tcIH, L.mipmapLevel);
    if(rcCenter.w < (255.f * 0.9f) || tcCenter.w < (255.f * 0.4f))
        return INFINITY;
    // DEV_HOMOGENEOUS: P (p + a x d + b y d) = h0 + a (M x d) + b (M y d)
    const f3 hr0 = M3x4mulV3(rc.P, Rp.p), ht0 = M3x4mulV3(tc.P, T.p);
    const f3 rax = M3x3mulV3(rc.P, Rp.x * Rp.d), ray = M3x3mulV3(rc.P, Rp.y * Rp.d);
    const f3 tax = M3x3mulV3(tc.P, T.x * T.d), tay = M3x3mulV3(tc.P, T.y * T.d);
    const float log2e = 1.44269504088896340736f;
    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
#pragma unroll 1
    for(int yp = -L.wsh; yp <= L.wsh; ++yp)
#pragma unroll 1
        for(int xp = -L.wsh; xp <= L.wsh; ++xp)
        {
            float2 rpc, tpc;
            if(dev & DEV_HOMOGENEOUS)
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
            const float4 rcC = tex2DLod(L.rcT, (rpc.x + 0.5f) * rcIW, (rpc.y + 0.5f) * rcIH, L.mipmapLevel);
            const float4 tcC = tex2DLod(L.tcT, (tpc.x + 0.5f) * tcIW, (tpc.y + 0.5f) * tcIH, L.mipmapLevel);
            float w;
            if(dev & DEV_MERGED_EXP)
            {
                // exp(-(dCr / gC + dP / gP)) exp(-(dCt / gC + dP / gP)) = exp2((dCr + dCt) (-log2e / gC) - 2 dP log2e / gP)
                const float drx = rcCenter.x - rcC.x, dry = rcCenter.y - rcC.y, drz = rcCenter.z - rcC.z;
                const float dtx = tcCenter.x - tcC.x, dty = tcCenter.y - tcC.y, dtz = tcCenter.z - tcC.z;
                const float dcr = __builtin_amdgcn_sqrtf(fmaf(drx, drx, fmaf(dry, dry, drz * drz)));
                const float dct = __builtin_amdgcn_sqrtf(fmaf(dtx, dtx, fmaf(dty, dty, dtz * dtz)));
                const float tabv = 2.0f * sqrtf((float)(xp * xp + yp * yp)) * L.invGammaP * log2e;
                w = __builtin_amdgcn_exp2f(fmaf(dcr + dct, -L.invGammaC * log2e, -tabv));
            }
            else
            {
                const float wr = cost_yk(xp, yp, rcCenter, rcC, L.invGammaC, L.invGammaP);
                const float wt = cost_yk(xp, yp, tcCenter, tcC, L.invGammaC, L.invGammaP);
                w = wr * wt;
            }
            if(dev & DEV_SHIFTED_SUMS)
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
    if(dev & DEV_SHIFTED_SUMS)
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
        return sigmoid(0.0f, 1.0f, 0.7f, -0.7f, sim);
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

__global__ void __launch_bounds__(256)
  literal_similarity_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, long long pitch_y, int pitch_x, const float* __restrict__ depths,
                            avdm_camera_t rc, avdm_camera_t tc, LitArgs L, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
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
#pragma unroll 1
    for(int k = 0; k < 4; ++k)
    {
        const unsigned vz = z0 + k;
        if(vz < zBegin || vz >= zEnd)
            continue;
        // volume_computePatch (kernels.cuh:26-35)
        auto patch_of = [&](unsigned z) -> LitPatch {
            const f3 C = ld3(rc.C), Z = ld3(rc.ZVect);
            const f3 planep = C + Z * depths[z];
            const f3 v = normalize_lit(M3x3mulV2(rc.iP, x, y));
            LitPatch q;
            q.p = linePlaneIntersect(C, v, planep, Z);
            q.d = pix_size_lit(rc, q.p);
            patch_axes_lit(rc, tc, q.p, nullptr, q.x, q.y);
            return q;
        };
        const LitPatch T = patch_of(vz);
        // DEV_SHARED_R: the R side from plane 1 of the aligned group of four (clamped to the T camera's range), like the four-plane pass
        const unsigned zr = min(max(z0 + 1u, zBegin), zEnd - 1u);
        const LitPatch Rp = (L.dev & DEV_SHARED_R) ? patch_of(zr) : T;
        float fsim = ncc_literal<false>(rc, tc, L, T, Rp, x, y);
        if(fsim == INFINITY)
            fsim = 255.0f;
        else
        {
            fsim = (fsim - (-1.0f)) * (1.0f / (1.0f - (-1.0f)));
            fsim = fminf(1.0f, fmaxf(0.0f, fsim));
            fsim *= 254.0f;
        }
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
    *reinterpret_cast<unsigned*>(pb) = wb;
    *reinterpret_cast<unsigned*>(ps) = ws;
}

__global__ void __launch_bounds__(256)
  literal_refine_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize, int map_pitch,
                        const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, LitArgs L, int stepXY, unsigned zBegin,
                        unsigned zEnd, avdm_roi_t roi)
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
#pragma unroll 1
    for(unsigned vz = zBegin; vz < zEnd; ++vz)
    {
        auto patch_of = [&](unsigned z) -> LitPatch {
            const f3 C = ld3(rc.C);
            LitPatch q;
            q.p = C + normalize_lit(M3x3mulV2(rc.iP, x, y)) * dps.x; // get3DPointForPixelAndDepthFromRC
            const int rel = (int)z - ((volDimZ - 1) / 2);
            if(rel != 0)
                q.p = q.p + normalize_lit(q.p - C) * ((float)rel * dps.y); // move3DPointByRcPixSize (kernels.cuh:17-24)
            q.d = pix_size_lit(rc, q.p);
            patch_axes_lit(rc, tc, q.p, nn, q.x, q.y);
            return q;
        };
        const LitPatch T = patch_of(vz);
        const unsigned zr = min(max((vz & ~3u) + 1u, zBegin), zEnd - 1u);
        const LitPatch Rp = (L.dev & DEV_SHARED_R) ? patch_of(zr) : T;
        const float fsim = ncc_literal<true>(rc, tc, L, T, Rp, x, y);
        if(fsim == INFINITY)
            continue;
        pv[vz] = __float2half(__half2float(pv[vz]) + fsim);
    }
}

static LitArgs make_args(const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale, int wsh, double gammaC, double gammaP)
{
    LitArgs L;
    L.rcT = make_tex(rcPyr);
    L.tcT = make_tex(tcPyr);
    L.rcW = (float)tex_dim_w(rcPyr, scale);
    L.rcH = (float)tex_dim_h(rcPyr, scale);
    L.tcW = (float)tex_dim_w(tcPyr, scale);
    L.tcH = (float)tex_dim_h(tcPyr, scale);
    L.mipmapLevel = tex_level_of(rcPyr, scale);
    L.invGammaC = 1.f / (float)gammaC;
    L.invGammaP = 1.f / (float)gammaP;
    L.wsh = wsh;
    const char* e = getenv("AVDM_SIM_LITERAL_DEV");
    L.dev = e != nullptr ? atoi(e) : 0;
    return L;
}

} // namespace

int literal_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                               const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_sgm_params_t* sp, avdm_range_t dr,
                               avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, sp->scale, sp->wsh, sp->gammaC, sp->gammaP);
    const unsigned nchunks = ((dr.end + 3) >> 2) - (dr.begin >> 2);
    hipLaunchKernelGGL(literal_similarity_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), nchunks), dim3(256), 0,
                       (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, L, sp->stepXY, dr.begin, dr.end, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity(literal)");
}

int literal_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal,
                              int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                              const avdm_refine_params_t* rp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    const LitArgs L = make_args(rc_pyr, tc_pyr, rp->scale, rp->wsh, rp->gammaC, rp->gammaP);
    hipLaunchKernelGGL(literal_refine_kernel, dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), 1), dim3(256), 0, (hipStream_t)stream,
                       (__half*)vol_f16, pitch_y, pitch_x, dimZ, (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, L, rp->stepXY,
                       dr.begin, dr.end, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity(literal)");
}

} // namespace avdm
