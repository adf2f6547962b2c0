// avdm_sgm.hip — SGM path aggregation of the uint8 cost volume + winner-take-all depth retrieval, for gfx950.
//   avdm_volume_optimize            <-> cuda_volumeOptimize / cuda_volumeAggregatePath (planeSweeping/deviceSimilarityVolume.cu:262-425;
//                                       kernels planeSweeping/deviceSimilarityVolumeKernels.cuh:596-744)
//   avdm_volume_retrieve_best_depth <-> cuda_volumeRetrieveBestDepth (deviceSimilarityVolume.cu:427-467; kernels.cuh:393-512)
//
// This translation unit is compiled with -ffp-contract=off: the aggregation is integer-valued except for the adaptive P2,
// and every fp32 operation below is written in the reference's order so that the stage is BIT-EXACT against the oracle.
//
// CDNA4 design (the reference issues ~3 tiny kernels per slice, ~10^4 launches per volume, through uint32 slice copies):
//   * ONE launch per path.  One wave64 owns one column (fixed position on the non-scanned image axis) and walks the scanned
//     axis as a persistent loop; the previous-slice path costs L(z) never leave VGPRs.
//   * z-fastest volume: a lane owns 4*NW consecutive planes, so a step is one coalesced 256*NW-byte read of the input
//     volume, (for paths 1..3) one of the output volume, and one coalesced write.  Algorithmic traffic only: 11 B/voxel total.
//   * min over z  = lane-local min + 6 DPP v_min steps (row_shr / row_bcast) + v_readlane; z±1 neighbours = wave_shr/shl DPP.
//   * the colour-adaptive P2 of 64 consecutive steps is evaluated at once, one step per lane, and read back with v_readlane.
//   * loads of the next PF steps are issued before the current PF steps are processed (software prefetch ring in VGPRs):
//     the recurrence only depends on registers, never on the loads of the step being issued.
#include "avdm_device.h"

#include <math.h>

namespace avdm {

// Fully specified exp of the P2 sigmoid — identical operation sequence to oracle/avdm_oracle.c:avo_exp_p2.
__device__ __forceinline__ float exp_p2(float x)
{
    x = x > 88.0f ? 88.0f : x;
    x = x < -80.0f ? -80.0f : x;
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

struct SgmArgs
{
    const uint8_t* in;
    uint8_t* out;
    long long strideA, strideB; // bytes between consecutive columns / consecutive steps
    int A, B, Z;
    int rev;
    float P1, P2w;
    TexLevel L;  // R image at the SGM mip level
    float rcW, rcH; // nominal level dims (DeviceMipmapImage::getDimensions)
    int beginX, beginY; // ROI offsets as the reference applies them to (v.x, v.y)
    int scanIsX;        // 1: the scanned axis is volume x (axisT.y == 0), 0: volume y
    float step;
    int fixed8;
};

#define SGM_BIG 3.0e38f

__device__ __forceinline__ float p2_of_step(const SgmArgs& S, int a, int b)
{
    if(S.P2w < 0)
        return fabsf(S.P2w);
    const int vx = S.scanIsX ? b : a, vy = S.scanIsX ? a : b;
    const int ySign = S.rev ? -1 : 1;
    const int imX0 = (int)((float)(S.beginX + vx) * S.step);
    const int imY0 = (int)((float)(S.beginY + vy) * S.step);
    const int imX1 = (int)((float)imX0 - (float)ySign * S.step * (float)(S.scanIsX ? 1 : 0));
    const int imY1 = (int)((float)imY0 - (float)ySign * S.step * (float)(S.scanIsX ? 0 : 1));
    const float u0 = ((float)imX0 + 0.5f) / S.rcW, v0 = ((float)imY0 + 0.5f) / S.rcH;
    const float u1 = ((float)imX1 + 0.5f) / S.rcW, v1 = ((float)imY1 + 0.5f) / S.rcH;
    const float4 c0 = S.fixed8 ? tex2D_level<true>(S.L, u0, v0) : tex2D_level<false>(S.L, u0, v0);
    const float4 c1 = S.fixed8 ? tex2D_level<true>(S.L, u1, v1) : tex2D_level<false>(S.L, u1, v1);
    const float dx = c0.x - c1.x, dy = c0.y - c1.y, dz = c0.z - c1.z;
    const float deltaC = sqrtf(dx * dx + dy * dy + dz * dz);
    return 80.f + (255.f - 80.f) * (1.0f / (1.0f + exp_p2(10.0f * ((deltaC - S.P2w) / 80.f))));
}

template <int NW, int K>
__global__ void __launch_bounds__(256) sgm_path_kernel(SgmArgs S)
{
    constexpr int ZL = 4 * NW;
    constexpr int SGM_PF = NW == 1 ? 8 : (NW == 2 ? 4 : 2); // prefetch depth (steps in flight), bounded by VGPR budget
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if(a >= S.A)
        return; // whole wave
    const int z0 = lane * ZL;
    const int Z = S.Z;
    const bool tailBytes = (Z & 3) != 0;

    const uint8_t* inCol = S.in + (long long)a * S.strideA + z0;
    uint8_t* outCol = S.out + (long long)a * S.strideA + z0;

    // which of my dwords hold at least one valid plane / are fully valid
    bool wAny[NW], wFull[NW];
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        wAny[w] = z0 + 4 * w < Z;
        wFull[w] = z0 + 4 * w + 3 < Z;
    }

    float prev[ZL];

    // ---- slice 0: prev = in(b = 0) (always slice 0, also for the reverse path), out(b = 0) = 255 ----
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        unsigned v = 0;
        if(wAny[w])
            v = *reinterpret_cast<const unsigned*>(inCol + 4 * w);
#pragma unroll
        for(int j = 0; j < 4; ++j)
            prev[4 * w + j] = (z0 + 4 * w + j < Z) ? (float)((v >> (8 * j)) & 0xffu) : SGM_BIG;
        if(wFull[w])
            *reinterpret_cast<unsigned*>(outCol + 4 * w) = 0xffffffffu;
        else if(wAny[w])
            for(int j = 0; j < 4; ++j)
                if(z0 + 4 * w + j < Z)
                    outCol[4 * w + j] = 255;
    }

    const int nSteps = S.B - 1; // ib = 1 .. B-1
    unsigned ringIn[SGM_PF][NW], ringOut[SGM_PF][NW];

    auto issue = [&](int ib, unsigned (&ri)[NW], unsigned (&ro)[NW]) {
        if(ib > nSteps)
            return;
        const int b = S.rev ? S.B - 1 - ib : ib;
        const long long off = (long long)b * S.strideB;
#pragma unroll
        for(int w = 0; w < NW; ++w)
        {
            ri[w] = 0;
            ro[w] = 0;
            if(wAny[w])
            {
                ri[w] = *reinterpret_cast<const unsigned*>(inCol + off + 4 * w);
                if(K > 0)
                    ro[w] = *reinterpret_cast<const unsigned*>(outCol + off + 4 * w);
            }
        }
    };

    // prologue: loads of steps 1..PF
#pragma unroll
    for(int s = 0; s < SGM_PF; ++s)
        issue(1 + s, ringIn[s], ringOut[s]);

    float p2vec = 0.f;

    for(int ib0 = 1; ib0 <= nSteps; ib0 += SGM_PF)
    {
        unsigned curIn[SGM_PF][NW], curOut[SGM_PF][NW];
#pragma unroll
        for(int s = 0; s < SGM_PF; ++s)
#pragma unroll
            for(int w = 0; w < NW; ++w)
            {
                curIn[s][w] = ringIn[s][w];
                curOut[s][w] = ringOut[s][w];
            }
            // next group's loads go out before this group's arithmetic
#pragma unroll
        for(int s = 0; s < SGM_PF; ++s)
            issue(ib0 + SGM_PF + s, ringIn[s], ringOut[s]);

#pragma unroll
        for(int s = 0; s < SGM_PF; ++s)
        {
            const int ib = ib0 + s;
            if(ib > nSteps)
                break;
            const int b = S.rev ? S.B - 1 - ib : ib;

            // P2 of 64 consecutive steps at once (one per lane), refreshed every 64 steps
            if(((ib - 1) & 63) == 0)
            {
                const int myIb = ib + lane;
                const int myB = S.rev ? S.B - 1 - myIb : myIb;
                p2vec = (myIb <= nSteps) ? p2_of_step(S, a, myB) : 0.f;
            }
            const float P2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2vec), (ib - 1) & 63));

            // best cost of the previous slice over all planes
            float m = prev[0];
#pragma unroll
            for(int j = 1; j < ZL; ++j)
                m = fminf(m, prev[j]);
            const float best = wave_min_f32(m);
            const float bestP2 = best + P2;

            // z-1 / z+1 neighbours across lanes
            const float left = dpp_f32<0x138>(SGM_BIG, prev[ZL - 1]); // from lane-1
            const float right = dpp_f32<0x130>(SGM_BIG, prev[0]);     // from lane+1

            float nprev[ZL];
#pragma unroll
            for(int w = 0; w < NW; ++w)
            {
                unsigned outw = 0;
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    const int i = 4 * w + j;
                    const int z = z0 + i;
                    const float cur = (float)((curIn[s][w] >> (8 * j)) & 0xffu);
                    const float pm1 = (i == 0) ? left : prev[i - 1];
                    const float pp1 = (i == ZL - 1) ? right : prev[i + 1];
                    float pathCost = 255.0f;
                    if(z >= 1 && z < Z - 1)
                    {
                        const float minCost = fminf(fminf(fminf(prev[i], pm1 + S.P1), pp1 + S.P1), bestP2);
                        pathCost = cur + minCost - best;
                    }
                    nprev[i] = (z < Z) ? truncf(pathCost) : SGM_BIG; // TSimAcc(pathCost): float -> uint32 truncation
                    pathCost = fminf(255.0f, fmaxf(0.0f, pathCost));
                    float val;
                    if(K == 0)
                        val = pathCost;
                    else
                    {
                        const float o = (float)((curOut[s][w] >> (8 * j)) & 0xffu);
                        val = (o * (float)K + pathCost) / (float)(K + 1);
                    }
                    outw |= ((unsigned)val & 0xffu) << (8 * j);
                }
                uint8_t* po = outCol + (long long)b * S.strideB + 4 * w;
                if(wFull[w])
                    *reinterpret_cast<unsigned*>(po) = outw;
                else if(tailBytes && wAny[w])
                {
#pragma unroll
                    for(int j = 0; j < 4; ++j)
                        if(z0 + 4 * w + j < Z)
                            po[j] = (uint8_t)(outw >> (8 * j));
                }
            }
#pragma unroll
            for(int i = 0; i < ZL; ++i)
                prev[i] = nprev[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// retrieve best depth: one wave scans one pixel's planes per iteration (coalesced 256*NW-byte read, DPP arg-min),
// 64 pixels per wave; then one lane per pixel converts plane -> ray distance / thickness and writes float2 coalesced.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float depthPlaneToDepth(const avdm_camera_t& cam, float fpPlaneDepth, float px, float py)
{
    const f3 C = ld3(cam.C), Zv = ld3(cam.ZVect);
    const f3 planep = C + Zv * fpPlaneDepth;
    // normalize() of the reference uses the fast reciprocal square root; the oracle restates it exactly (1/sqrtf)
    f3 v = M3x3mulV2(cam.iP, px, py);
    const float dInv = 1.0f / sqrtf(dot(v, v));
    v = f3{v.x * dInv, v.y * dInv, v.z * dInv};
    const f3 p = linePlaneIntersect(C, v, planep, Zv);
    return size(C - p);
}

__global__ void __launch_bounds__(256)
  retrieve_best_depth_kernel(float2* outDT, int dt_pitch, float2* outDS, int ds_pitch, const float* __restrict__ depths,
                             const uint8_t* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, avdm_camera_t rc, int scaleStep,
                             float thicknessMultFactor, float maxSimilarity, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    const int lane = threadIdx.x & 63;
    const unsigned roiW = roi.x.end - roi.x.begin;
    const unsigned vy = blockIdx.y;
    const unsigned x0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; // first pixel of this wave
    if(x0 >= roiW)
        return;

    unsigned myKey = 0xffffffffu; // (sim << 16 | z) of the pixel x0 + lane
    const unsigned nPix = min(64u, roiW - x0);
    for(unsigned i = 0; i < nPix; ++i)
    {
        const uint8_t* col = vol + (long long)vy * pitch_y + (long long)(x0 + i) * pitch_x;
        unsigned key = 0xffffffffu;
        for(unsigned zb = (zBegin & ~3u) + 4u * lane; zb < zEnd; zb += 256u)
        {
            const unsigned w = *reinterpret_cast<const unsigned*>(col + zb);
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                const unsigned z = zb + j;
                const unsigned s = (w >> (8 * j)) & 0xffu;
                // strict '<' against 255 and first-minimum-wins == min over (sim, z) keys restricted to sim < 255
                if(z >= zBegin && z < zEnd && s < 255u)
                    key = min(key, (s << 16) | z);
            }
        }
        key = wave_min_u32(key);
        if((unsigned)lane == i)
            myKey = key;
    }

    const unsigned vx = x0 + lane;
    if(vx >= roiW)
        return;
    float2* dt = (float2*)((char*)outDT + (long long)vy * dt_pitch) + vx;
    float2* ds = outDS ? (float2*)((char*)outDS + (long long)vy * ds_pitch) + vx : nullptr;

    const float bestSim = (myKey == 0xffffffffu) ? 255.f : (float)(myKey >> 16);
    const int bestZIdx = (myKey == 0xffffffffu) ? -1 : (int)(myKey & 0xffffu);
    if((bestZIdx == -1) || (bestSim > maxSimilarity))
    {
        *dt = make_float2(-1.f, -1.f);
        if(ds)
            *ds = make_float2(-1.f, 1.f);
        return;
    }
    const float px = (float)((roi.x.begin + vx) * scaleStep), py = (float)((roi.y.begin + vy) * scaleStep);
    const int m1 = max(0, bestZIdx - 1);
    const int p1 = min(volDimZ - 1, bestZIdx + 1);
    const float bestDepth = depthPlaneToDepth(rc, depths[bestZIdx], px, py);
    const float bestDepth_m1 = depthPlaneToDepth(rc, depths[m1], px, py);
    const float bestDepth_p1 = depthPlaneToDepth(rc, depths[p1], px, py);
    const float out_bestSim = (bestSim / 255.0f) * 2.0f - 1.0f;
    const float thick = fmaxf(bestDepth_p1 - bestDepth, bestDepth - bestDepth_m1) * thicknessMultFactor;
    *dt = make_float2(bestDepth, thick);
    if(ds)
        *ds = make_float2(bestDepth, out_bestSim);
}

template <int NW>
static void launch_path(const SgmArgs& S, int K, hipStream_t st)
{
    dim3 grid(divUp(S.A, 4));
    switch(K)
    {
        case 0: hipLaunchKernelGGL((sgm_path_kernel<NW, 0>), grid, dim3(256), 0, st, S); break;
        case 1: hipLaunchKernelGGL((sgm_path_kernel<NW, 1>), grid, dim3(256), 0, st, S); break;
        case 2: hipLaunchKernelGGL((sgm_path_kernel<NW, 2>), grid, dim3(256), 0, st, S); break;
        default: hipLaunchKernelGGL((sgm_path_kernel<NW, 3>), grid, dim3(256), 0, st, S); break;
    }
}

} // namespace avdm

using namespace avdm;

extern "C" {

size_t avdm_volume_optimize_scratch_bytes(int dimX, int dimY, int dimZ)
{
    (void)dimX;
    (void)dimY;
    (void)dimZ;
    return 0; // the path costs of the previous slice live in registers; no slice buffers (Sgm.hpp:144-148 of the reference)
}

int avdm_volume_optimize(uint8_t* out_vol, const uint8_t* in_vol, long long pitch_y, int pitch_x, void* scratch, const avdm_pyramid_t* rc_pyr,
                         const avdm_sgm_params_t* sp, int last_depth_index, avdm_roi_t roi, void* stream)
{
    (void)scratch;
    const int dimX = (int)(roi.x.end - roi.x.begin), dimY = (int)(roi.y.end - roi.y.begin), Z = last_depth_index;
    if(dimX <= 0 || dimY <= 0 || Z <= 0)
        return 0;
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)out_vol & 3) || ((uintptr_t)in_vol & 3))
        return set_error_msg(1, "avdm_volume_optimize: volume base / pitches must be multiples of 4 bytes");
    if(((Z + 3) & ~3) > pitch_x)
        return set_error_msg(1, "avdm_volume_optimize: pitch_x must cover the 4-aligned depth count");
    if(Z > 1536)
        return set_error_msg(1, "avdm_volume_optimize: more than 1536 depth planes are not supported");
    int level;
    if(!lod_is_integral(rc_pyr, sp->scale, &level))
        return set_error_msg(1, "avdm_volume_optimize: non-integral mip level");
    const Tex t = make_tex(rc_pyr);

    SgmArgs S;
    S.in = in_vol;
    S.out = out_vol;
    S.Z = Z;
    S.P1 = (float)sp->p1;
    S.P2w = (float)sp->p2Weighting;
    S.L = t.lv[level];
    S.rcW = (float)tex_dim_w(rc_pyr, sp->scale);
    S.rcH = (float)tex_dim_h(rc_pyr, sp->scale);
    S.step = (float)sp->stepXY;
    S.fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;

    const int NW = (Z + 255) / 256;
    int npaths = 0;
    for(const char* ax = sp->filteringAxes; *ax; ++ax)
    {
        if(*ax != 'X' && *ax != 'Y')
            continue;
        const bool scanX = (*ax == 'X');
        S.scanIsX = scanX ? 1 : 0;
        S.A = scanX ? dimY : dimX;
        S.B = scanX ? dimX : dimY;
        S.strideA = scanX ? pitch_y : (long long)pitch_x;
        S.strideB = scanX ? (long long)pitch_x : pitch_y;
        // deviceSimilarityVolumeKernels.cuh:688-689: beginX = (axisT.x == 0) ? roi.x.begin : roi.y.begin, applied to v.x (sic)
        const bool swap = sp->strictRoiQuirk && scanX;
        S.beginX = swap ? (int)roi.y.begin : (int)roi.x.begin;
        S.beginY = swap ? (int)roi.x.begin : (int)roi.y.begin;
        for(int rev = 0; rev < 2; ++rev)
        {
            S.rev = rev;
            const int K = npaths++;
            if(K > 3)
                return set_error_msg(1, "avdm_volume_optimize: at most 2 filtering axes");
            switch(NW)
            {
                case 1: launch_path<1>(S, K, (hipStream_t)stream); break;
                case 2: launch_path<2>(S, K, (hipStream_t)stream); break;
                case 3: launch_path<3>(S, K, (hipStream_t)stream); break;
                case 4: launch_path<4>(S, K, (hipStream_t)stream); break;
                case 5: launch_path<5>(S, K, (hipStream_t)stream); break;
                default: launch_path<6>(S, K, (hipStream_t)stream); break;
            }
        }
    }
    AVDM_LAUNCH_CHECK("avdm_volume_optimize");
}

int avdm_volume_retrieve_best_depth(float* out_depth_thickness, int dt_pitch, float* out_depth_sim, int ds_pitch, const float* depths,
                                    const uint8_t* vol, long long pitch_y, int pitch_x, int vol_dimZ, const avdm_camera_t* rc_scale1,
                                    const avdm_sgm_params_t* sp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)vol & 3))
        return set_error_msg(1, "avdm_volume_retrieve_best_depth: volume base / pitches must be multiples of 4 bytes");
    if(dr.end > 65535u)
        return set_error_msg(1, "avdm_volume_retrieve_best_depth: too many depth planes");
    const int scaleStep = sp->scale * sp->stepXY;
    const float thicknessMultFactor = 1.f + (float)sp->depthThicknessInflate;
    const float maxSimilarity = (float)sp->maxSimilarity * 254.f;
    dim3 grid(divUp(roiW, 256), roiH);
    hipLaunchKernelGGL(retrieve_best_depth_kernel, grid, dim3(256), 0, (hipStream_t)stream, (float2*)out_depth_thickness, dt_pitch,
                       (float2*)out_depth_sim, ds_pitch, depths, vol, pitch_y, pitch_x, vol_dimZ, *rc_scale1, scaleStep, thicknessMultFactor,
                       maxSimilarity, dr.begin, dr.end, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_retrieve_best_depth");
}

} // extern "C"
