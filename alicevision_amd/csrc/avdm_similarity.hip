// avdm_similarity.hip — plane-sweep weighted-NCC similarity volumes for gfx950.
//   avdm_volume_compute_similarity  <-> cuda_volumeComputeSimilarity  (planeSweeping/deviceSimilarityVolume.cu:155-206,
//                                       kernel planeSweeping/deviceSimilarityVolumeKernels.cuh:109-233)
//   avdm_volume_refine_similarity   <-> cuda_volumeRefineSimilarity   (deviceSimilarityVolume.cu:208-259, kernels.cuh:235-391)
//   NCC core                        <-> compNCCby3DptsYK              (cuda/device/Patch.cuh:466-572, SimStat.cuh, color.cuh:167-210)
//
// CDNA4 design (not the CUDA one):
//   * one lane per PIXEL, 8x8 pixels per wave64, 16x16 per workgroup: neighbouring lanes sample neighbouring texels of R and
//     of T for the same plane, so the L1/L2 serve the gathers; each lane walks a chunk of consecutive planes and writes its
//     results as one packed word into the z-fastest volume (4 x u8 = one dword RMW; 8 x fp16 = one 16-byte RMW).
//   * patch samples are projected in homogeneous form:  P*(p + a*x + b*y) = h0 + a*(M*x) + b*(M*y)  — 3 FMA + 1 v_rcp per
//     camera and sample instead of a 3-D point + a 3x4 product; the two Yoon–Kweon exponentials are merged into one v_exp.
//   * camera parameters and the per-offset proximity table live in the kernarg segment (SGPR / scalar loads), no __constant__.
#include "avdm_device.h"

#include <math.h>

namespace avdm {

struct PatchTable
{
    float c[81]; // 2 * sqrt(xp^2 + yp^2) * invGammaP * log2(e), row-major over (yp, xp), for wsh <= 4
};

struct NccArgs
{
    TexLevel rcL, tcL;         // integral-level fast path
    float rcSx, rcOx, rcSy, rcOy; // nominal-level pixel -> texel space of the actual level: x = px*S + O
    float tcSx, tcOx, tcSy, tcOy;
    float rcW1, rcH1, tcW1, tcH1; // float(levelDim - 1) of the nominal dims (border test)
    float negInvGammaC_log2e;
    float invGammaC, invGammaP;
    float mipmapLevel;
    int wsh;
};

__device__ __forceinline__ float4 fetch(const TexLevel& L, bool fixed8, float x, float y)
{
    return fixed8 ? tex_bilinear_px<true>(L, x, y) : tex_bilinear_px<false>(L, x, y);
}

// returns +INF when invalid / masked.  TInvert: sigmoid-filtered positive similarity (Refine) else raw NCC in [-1, 1].
template <bool FIXED8, int WSH, bool TInvert>
__device__ __forceinline__ float ncc_patch(const avdm_camera_t& rc, const avdm_camera_t& tc, const NccArgs& A, const PatchTable& tab, f3 pp, f3 px,
                                           f3 py, float pd, float rpx, float rpy)
{
    const int wsh = WSH > 0 ? WSH : A.wsh;
    // homogeneous image coordinates of the patch centre and of the two scaled patch axes
    const f3 hr0 = M3x4mulV3(rc.P, pp);
    const f3 ht0 = M3x4mulV3(tc.P, pp);
    // (rpx, rpy): the patch centre lies on the ray of pixel (x, y), so its R projection IS (x, y); using the exact pixel makes
    // the border test deterministic on the knife-edge rows where x == wsh + 2 (DESIGN.md "knife-edge rows")
    const float it0 = fast_rcp(ht0.z);
    const float tpx = ht0.x * it0, tpy = ht0.y * it0;

    const float dd = (float)wsh + 2.0f;
    if((rpx < dd) || (rpx > A.rcW1 - dd) || (tpx < dd) || (tpx > A.tcW1 - dd) || (rpy < dd) || (rpy > A.rcH1 - dd) || (tpy < dd) ||
       (tpy > A.tcH1 - dd))
        return INFINITY;

    const float4 rcCenter = tex_bilinear_px<FIXED8>(A.rcL, fmaf(rpx, A.rcSx, A.rcOx), fmaf(rpy, A.rcSy, A.rcOy));
    const float4 tcCenter = tex_bilinear_px<FIXED8>(A.tcL, fmaf(tpx, A.tcSx, A.tcOx), fmaf(tpy, A.tcSy, A.tcOy));
    if(rcCenter.w < (255.f * 0.9f) || tcCenter.w < (255.f * 0.4f))
        return INFINITY;

    const f3 ax = px * pd, ay = py * pd; // patch.x * patch.d, patch.y * patch.d
    const f3 rax = M3x3mulV3(rc.P, ax), ray = M3x3mulV3(rc.P, ay);
    const f3 tax = M3x3mulV3(tc.P, ax), tay = M3x3mulV3(tc.P, ay);

    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
    const int n = 2 * wsh + 1;

#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
    {
        const float fy = (float)yp;
        const f3 hrRow = f3{fmaf(fy, ray.x, hr0.x), fmaf(fy, ray.y, hr0.y), fmaf(fy, ray.z, hr0.z)};
        const f3 htRow = f3{fmaf(fy, tay.x, ht0.x), fmaf(fy, tay.y, ht0.y), fmaf(fy, tay.z, ht0.z)};
        const float* trow = tab.c + (yp + wsh) * n + wsh;
#pragma unroll
        for(int xp = -(WSH > 0 ? WSH : 4); xp <= (WSH > 0 ? WSH : 4); ++xp)
        {
            if(WSH <= 0 && (xp < -wsh || xp > wsh))
                continue;
            const float fx = (float)xp;
#ifdef AVDM_DBG_REFPROJ
            const f3 p3 = pp + px * (pd * fx) + py * (pd * fy);
            const float2 rq = project3DPoint(rc.P, p3), tq = project3DPoint(tc.P, p3);
            const float rx = rq.x, ry = rq.y, tx = tq.x, ty = tq.y;
#else
            const float hrz = fmaf(fx, rax.z, hrRow.z), htz = fmaf(fx, tax.z, htRow.z);
            const float ir = fast_rcp(hrz), it = fast_rcp(htz);
            const float rx = fmaf(fx, rax.x, hrRow.x) * ir, ry = fmaf(fx, rax.y, hrRow.y) * ir;
            const float tx = fmaf(fx, tax.x, htRow.x) * it, ty = fmaf(fx, tax.y, htRow.y) * it;
#endif

            const float4 rcC = tex_bilinear_px<FIXED8>(A.rcL, fmaf(rx, A.rcSx, A.rcOx), fmaf(ry, A.rcSy, A.rcOy));
            const float4 tcC = tex_bilinear_px<FIXED8>(A.tcL, fmaf(tx, A.tcSx, A.tcOx), fmaf(ty, A.tcSy, A.tcOy));

            // w = exp(-(dC_r/gC + dP/gP)) * exp(-(dC_t/gC + dP/gP)) = exp2((dC_r + dC_t) * (-log2e/gC) - 2*dP*log2e/gP)
            const float drx = rcCenter.x - rcC.x, dry = rcCenter.y - rcC.y, drz = rcCenter.z - rcC.z;
            const float dtx = tcCenter.x - tcC.x, dty = tcCenter.y - tcC.y, dtz = tcCenter.z - tcC.z;
            const float dcr = __builtin_amdgcn_sqrtf(fmaf(drx, drx, fmaf(dry, dry, drz * drz)));
            const float dct = __builtin_amdgcn_sqrtf(fmaf(dtx, dtx, fmaf(dty, dty, dtz * dtz)));
            const float w = __builtin_amdgcn_exp2f(fmaf(dcr + dct, A.negInvGammaC_log2e, -trow[xp]));

            // NCC statistics on L shifted by the centre values (gx = L_r(centre) - L_r(sample), same for T): variances and the
            // covariance are shift- and (joint) sign-invariant, and the shifted sums do not cancel catastrophically in fp32
            // the way sum(w L^2) - sum(w L)^2 / sum(w) does with L ~ 200 (DESIGN.md "NCC conditioning").
            const float gx = drx, gy = dtx;
            const float wgx = w * gx, wgy = w * gy;
            wsum += w;
            xsum += wgx;
            ysum += wgy;
            xxsum = fmaf(wgx, gx, xxsum);
            yysum = fmaf(wgy, gy, yysum);
            xysum = fmaf(wgx, gy, xysum);
        }
    }

    const float iw = fast_rcp(wsum);
    const float varXW = (xxsum - xsum * xsum * iw) * iw;
    const float varYW = (yysum - ysum * ysum * iw) * iw;
    const float varXYW = (xysum - xsum * ysum * iw) * iw;
    const float rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
    const float sim = isfinite(rawSim) ? -rawSim : 1.0f;
    if(TInvert)
        return sigmoid(0.0f, 1.0f, 0.7f, -0.7f, sim);
    return sim;
}

__device__ __forceinline__ void patch_axes(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 p, f3& ax, f3& ay)
{
    // computeRotCSEpip (Patch.cuh:111-135); only x and y are used downstream
    const f3 v1 = normalize(ld3(rc.C) - p);
    const f3 v2 = normalize(ld3(tc.C) - p);
    ay = normalize(cross(v1, v2));
    const f3 n = normalize((v1 + v2) * 0.5f);
    ax = normalize(cross(ay, n));
}

__device__ __forceinline__ void pixel_of_lane(int& tx, int& ty)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    tx = (w & 1) * 8 + (lane & 7);
    ty = (w >> 1) * 8 + (lane >> 3);
}

// ---------------------------------------------------------------------------------------------
// SGM similarity: best / second-best uint8 volumes, 4 planes per lane per launch-z
// ---------------------------------------------------------------------------------------------
template <bool FIXED8, int WSH>
__global__ void __launch_bounds__(256)
  similarity_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, long long pitch_y, int pitch_x, const float* __restrict__ depths,
                    avdm_camera_t rc, avdm_camera_t tc, NccArgs A, PatchTable tab, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    int tx, ty;
    pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const unsigned z0 = ((zBegin >> 2) + blockIdx.z) << 2;

    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;

    // pixel ray (shared by the 4 planes): get3DPointForPixelAndFrontoParellePlaneRC restated
    const f3 C = ld3(rc.C), Z = ld3(rc.ZVect);
    const f3 v = normalize(M3x3mulV2(rc.iP, x, y));
    const float dnC = dot(Z, C), dnv = dot(Z, v);

    uint8_t* pb = best + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    uint8_t* ps = second + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    unsigned wb = *reinterpret_cast<const unsigned*>(pb);
    unsigned ws = *reinterpret_cast<const unsigned*>(ps);

#pragma unroll 1
    for(int k = 0; k < 4; ++k)
    {
        const unsigned vz = z0 + k;
        if(vz < zBegin || vz >= zEnd)
            continue;
        const float depthPlane = depths[vz];
        const f3 planep = C + Z * depthPlane;
        const float kk = (dot(planep, Z) - dnC) / dnv;
        const f3 p = C + v * kk;
        const float pd = computePixSize(rc, p);
        f3 ax, ay;
        patch_axes(rc, tc, p, ax, ay);
        float fsim = ncc_patch<FIXED8, WSH, false>(rc, tc, A, tab, p, ax, ay, pd, x, y);
        if(fsim == INFINITY)
            fsim = 255.0f;
        else
        {
            fsim = (fsim + 1.0f) * 0.5f;
            fsim = fminf(1.0f, fmaxf(0.0f, fsim));
            fsim *= 254.0f;
        }
        const unsigned sh = 8u * k;
        const unsigned b1 = (wb >> sh) & 0xffu, b2 = (ws >> sh) & 0xffu;
        if(fsim < (float)b1)
        {
            ws = (ws & ~(0xffu << sh)) | (b1 << sh);
            wb = (wb & ~(0xffu << sh)) | ((unsigned)fsim << sh);
        }
        else if(fsim < (float)b2)
            ws = (ws & ~(0xffu << sh)) | ((unsigned)fsim << sh);
    }
    *reinterpret_cast<unsigned*>(pb) = wb;
    *reinterpret_cast<unsigned*>(ps) = ws;
}

// ---------------------------------------------------------------------------------------------
// Refine similarity: fp16 volume += sigmoid-filtered NCC, 8 planes per lane per launch-z
// ---------------------------------------------------------------------------------------------
template <bool FIXED8, int WSH>
__global__ void __launch_bounds__(256)
  refine_similarity_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize,
                           int map_pitch, const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, NccArgs A,
                           PatchTable tab, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    int tx, ty;
    pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const float2 dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    if(dps.x <= 0.0f)
        return;
    const unsigned z0 = ((zBegin >> 3) + blockIdx.z) << 3;

    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const f3 C = ld3(rc.C);
    const f3 rpv = normalize(M3x3mulV2(rc.iP, x, y));
    const f3 pMid = C + rpv * dps.x;
    // move3DPointByRcPixSize direction: normalize(p - C) recomputed from the mid point like the reference (kernels.cuh:17-24)
    const f3 dir = normalize(pMid - C);

    __half* pv = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2 + z0;
    uint4 packed = *reinterpret_cast<const uint4*>(pv);
    __half* hv = reinterpret_cast<__half*>(&packed);

#pragma unroll 1
    for(int k = 0; k < 8; ++k)
    {
        const unsigned vz = z0 + k;
        if(vz < zBegin || vz >= zEnd)
            continue;
        const int rel = (int)vz - ((volDimZ - 1) / 2);
        f3 p = pMid;
        if(rel != 0)
            p = pMid + dir * ((float)rel * dps.y);
        const float pd = computePixSize(rc, p);
        f3 ax, ay;
        {
            const f3 v1 = normalize(C - p);
            const f3 v2 = normalize(ld3(tc.C) - p);
            ay = normalize(cross(v1, v2));
            f3 n;
            if(sgmNormal != nullptr)
            {
                const float* nn = (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx;
                n = f3{nn[0], nn[1], nn[2]};
            }
            else
                n = normalize((v1 + v2) * 0.5f);
            ax = normalize(cross(ay, n));
        }
        const float fsim = ncc_patch<FIXED8, WSH, true>(rc, tc, A, tab, p, ax, ay, pd, x, y);
        if(fsim == INFINITY)
            continue;
        hv[k] = __float2half(__half2float(hv[k]) + fsim);
    }
    *reinterpret_cast<uint4*>(pv) = packed;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool fill_ncc_args(NccArgs& A, PatchTable& tab, const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale, int wsh, double gammaC,
                          double gammaP)
{
    int rl, tl;
    if(!lod_is_integral(rcPyr, scale, &rl) || !lod_is_integral(tcPyr, scale, &tl))
        return false;
    const Tex rt = make_tex(rcPyr), tt = make_tex(tcPyr);
    A.rcL = rt.lv[rl];
    A.tcL = tt.lv[tl];
    const int rcW = tex_dim_w(rcPyr, scale), rcH = tex_dim_h(rcPyr, scale), tcW = tex_dim_w(tcPyr, scale), tcH = tex_dim_h(tcPyr, scale);
    // tex2DLod((p + .5)/Wnominal) -> texel space of the actual level: (p + .5) * Wactual/Wnominal - .5
    A.rcSx = (float)A.rcL.W / (float)rcW;
    A.rcOx = 0.5f * A.rcSx - 0.5f;
    A.rcSy = (float)A.rcL.H / (float)rcH;
    A.rcOy = 0.5f * A.rcSy - 0.5f;
    A.tcSx = (float)A.tcL.W / (float)tcW;
    A.tcOx = 0.5f * A.tcSx - 0.5f;
    A.tcSy = (float)A.tcL.H / (float)tcH;
    A.tcOy = 0.5f * A.tcSy - 0.5f;
    A.rcW1 = (float)(rcW - 1);
    A.rcH1 = (float)(rcH - 1);
    A.tcW1 = (float)(tcW - 1);
    A.tcH1 = (float)(tcH - 1);
    A.invGammaC = 1.f / (float)gammaC;
    A.invGammaP = 1.f / (float)gammaP;
    const float log2e = 1.44269504088896340736f;
    A.negInvGammaC_log2e = -A.invGammaC * log2e;
    A.mipmapLevel = (float)rl;
    A.wsh = wsh;
    const int n = 2 * wsh + 1;
    for(int yp = -wsh; yp <= wsh; ++yp)
        for(int xp = -wsh; xp <= wsh; ++xp)
            tab.c[(yp + wsh) * n + (xp + wsh)] = 2.0f * sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP * log2e;
    return true;
}

} // namespace avdm

using namespace avdm;

extern "C" {

int avdm_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                                   const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                                   const avdm_sgm_params_t* sp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    if(dr.end <= dr.begin || roi.x.end <= roi.x.begin || roi.y.end <= roi.y.begin)
        return 0;
    if(sp->wsh < 1 || sp->wsh > 4)
        return set_error_msg(1, "avdm_volume_compute_similarity: wsh must be in [1, 4]");
    if(sp->useConsistentScale)
        return set_error_msg(1, "avdm_volume_compute_similarity: useConsistentScale is not supported yet (SURVEY §8f.4)");
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)best & 3) || ((uintptr_t)second & 3))
        return set_error_msg(1, "avdm_volume_compute_similarity: volume base / pitches must be multiples of 4 bytes");
    NccArgs A;
    PatchTable tab;
    if(!fill_ncc_args(A, tab, rc_pyr, tc_pyr, sp->scale, sp->wsh, sp->gammaC, sp->gammaP))
        return set_error_msg(1, "avdm_volume_compute_similarity: non-integral mip level");
    const unsigned nchunks = ((dr.end + 3) >> 2) - (dr.begin >> 2);
    if(((dr.end + 3) & ~3u) > (unsigned)pitch_x)
        return set_error_msg(1, "avdm_volume_compute_similarity: pitch_x too small for the depth range (must cover the 4-aligned range)");
    dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), nchunks);
    const bool fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
#define LAUNCH(F8, W)                                                                                                                                 \
    hipLaunchKernelGGL((similarity_kernel<F8, W>), grid, dim3(256), 0, (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, A, tab, \
                       sp->stepXY, dr.begin, dr.end, roi)
    if(fixed8)
    {
        if(sp->wsh == 4) LAUNCH(true, 4);
        else if(sp->wsh == 3) LAUNCH(true, 3);
        else LAUNCH(true, 0);
    }
    else
    {
        if(sp->wsh == 4) LAUNCH(false, 4);
        else if(sp->wsh == 3) LAUNCH(false, 3);
        else LAUNCH(false, 0);
    }
#undef LAUNCH
    AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity");
}

int avdm_volume_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch,
                                  const float* sgm_normal, int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc,
                                  const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_refine_params_t* rp, avdm_range_t dr,
                                  avdm_roi_t roi, void* stream)
{
    if(dr.end <= dr.begin || roi.x.end <= roi.x.begin || roi.y.end <= roi.y.begin)
        return 0;
    if(rp->wsh < 1 || rp->wsh > 4)
        return set_error_msg(1, "avdm_volume_refine_similarity: wsh must be in [1, 4]");
    if(rp->useConsistentScale)
        return set_error_msg(1, "avdm_volume_refine_similarity: useConsistentScale is not supported yet (SURVEY §8f.4)");
    if((pitch_x & 15) || (pitch_y & 15) || ((uintptr_t)vol_f16 & 15))
        return set_error_msg(1, "avdm_volume_refine_similarity: volume base / pitches must be multiples of 16 bytes");
    if((int)(((dr.end + 7) & ~7u) * 2) > pitch_x)
        return set_error_msg(1, "avdm_volume_refine_similarity: pitch_x too small (must cover the 8-aligned depth range)");
    NccArgs A;
    PatchTable tab;
    if(!fill_ncc_args(A, tab, rc_pyr, tc_pyr, rp->scale, rp->wsh, rp->gammaC, rp->gammaP))
        return set_error_msg(1, "avdm_volume_refine_similarity: non-integral mip level");
    const unsigned nchunks = ((dr.end + 7) >> 3) - (dr.begin >> 3);
    dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), nchunks);
    const bool fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
#define LAUNCH(F8, W)                                                                                                                              \
    hipLaunchKernelGGL((refine_similarity_kernel<F8, W>), grid, dim3(256), 0, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x, dimZ,         \
                       (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, A, tab, rp->stepXY, dr.begin, dr.end, roi)
    if(fixed8)
    {
        if(rp->wsh == 3) LAUNCH(true, 3);
        else if(rp->wsh == 4) LAUNCH(true, 4);
        else LAUNCH(true, 0);
    }
    else
    {
        if(rp->wsh == 3) LAUNCH(false, 3);
        else if(rp->wsh == 4) LAUNCH(false, 4);
        else LAUNCH(false, 0);
    }
#undef LAUNCH
    AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity");
}

} // extern "C"
