// avdm_sgm.hip — SGM path aggregation of the uint8 cost volume + winner-take-all depth retrieval, for gfx950.
//   avdm_volume_optimize            <-> cuda_volumeOptimize / cuda_volumeAggregatePath (planeSweeping/deviceSimilarityVolume.cu:262-425;
//                                       kernels planeSweeping/deviceSimilarityVolumeKernels.cuh:596-744)
//   avdm_volume_retrieve_best_depth <-> cuda_volumeRetrieveBestDepth (deviceSimilarityVolume.cu:427-467; kernels.cuh:393-512)
//
// This translation unit is compiled with -ffp-contract=off: the aggregation is integer-valued except for the adaptive P2,
// and every fp32 operation below is written in the reference's order so that the stage is BIT-EXACT against the oracle.
//
// CDNA4 design (the reference issues ~3 tiny kernels per slice, ~10^4 launches per volume, through uint32 slice copies):
//   * ONE launch per path.  One wave64 (= one workgroup) owns one column (fixed position on the non-scanned image axis) and
//     walks the scanned axis as a persistent loop; the previous-slice path costs L(z) never leave VGPRs.
//   * z-fastest volume: a lane owns 4*NW consecutive planes, so a step is one coalesced 256*NW-byte read of the input
//     volume, (for paths 1..3) one of the output volume, and one coalesced write.  Algorithmic traffic only: 11 B/voxel total.
//   * min over z  = lane-local min + 6 DPP v_min steps (row_shr / row_bcast) + v_readlane; z±1 neighbours = wave_shr/shl DPP.
//   * the colour-adaptive P2 is evaluated once per (column, slice) by a small map kernel per axis (one map serves the forward
//     and the reverse path); the path kernel reads 64 consecutive steps of it per lane-coalesced load and picks with v_readlane.
//   * only ~1000 columns exist per path (one wave per SIMD), so latency is hidden by ILP, not occupancy: a 4-slot register ring
//     keeps the loads of the next 24-32 slices in flight; the recurrence depends on registers only.
//   * the running average (out*K + L)/(K+1) -> uint8 is evaluated without an IEEE division: one exact FMA, then the integer
//     quotient through exact fp32 scalings (equality with the reference's float expression checked over every fp32 input).
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

// ---- adaptive P2 (kernels.cuh:696-720), evaluated once per (column, slice) into a float map ----------------------------
struct SgmP2Args
{
    TexLevel L;         // R image at the SGM mip level
    float rcW, rcH;     // nominal level dims (DeviceMipmapImage::getDimensions)
    int beginX, beginY; // ROI offsets as the reference applies them to (v.x, v.y)
    int scanIsX;        // 1: the scanned axis is volume x (axisT.y == 0), 0: volume y
    float step;
    float P2w;
    int A, B;
};

#define SGM_BIG 3.0e38f

// P2 of the FORWARD path at slice b of column a: colour step between stage pixel b and b - 1 along the scanned axis.
// The reverse path at slice b compares pixel b with b + 1 — the same texel pair as the forward path at b + 1, with the
// two fetches swapped; deltaC is a sum of squared differences, so the value is bit-identical and ONE map serves both
// directions of an axis (reverse reads entry b + 1).
template <bool FIXED8>
__global__ void __launch_bounds__(256) sgm_p2_map_kernel(float* __restrict__ p2, SgmP2Args S)
{
    const int b = blockIdx.x * 16 + (threadIdx.x & 15);
    const int a = blockIdx.y * 16 + (threadIdx.x >> 4);
    if(a >= S.A || b >= S.B)
        return;
    float P2;
    if(S.P2w < 0)
        P2 = fabsf(S.P2w);
    else if(b == 0)
        P2 = 0.f; // never read
    else
    {
        const int vx = S.scanIsX ? b : a, vy = S.scanIsX ? a : b;
        const int imX0 = (int)((float)(S.beginX + vx) * S.step);
        const int imY0 = (int)((float)(S.beginY + vy) * S.step);
        const int imX1 = (int)((float)imX0 - S.step * (float)(S.scanIsX ? 1 : 0));
        const int imY1 = (int)((float)imY0 - S.step * (float)(S.scanIsX ? 0 : 1));
        const float u0 = ((float)imX0 + 0.5f) / S.rcW, v0 = ((float)imY0 + 0.5f) / S.rcH;
        const float u1 = ((float)imX1 + 0.5f) / S.rcW, v1 = ((float)imY1 + 0.5f) / S.rcH;
        const float4 c0 = tex2D_level<FIXED8>(S.L, u0, v0);
        const float4 c1 = tex2D_level<FIXED8>(S.L, u1, v1);
        const float dx = c0.x - c1.x, dy = c0.y - c1.y, dz = c0.z - c1.z;
        const float deltaC = sqrtf(dx * dx + dy * dy + dz * dz);
        P2 = 80.f + (255.f - 80.f) * (1.0f / (1.0f + exp_p2(10.0f * ((deltaC - S.P2w) / 80.f))));
    }
    p2[(long long)a * S.B + b] = P2;
}

// ---- one aggregation path ------------------------------------------------------------------------------------------
struct SgmPathArgs
{
    const uint8_t* in;
    uint8_t* out;
    const float* p2;            // [A][B], see sgm_p2_map_kernel
    long long strideA, strideB; // bytes between consecutive columns / consecutive slices
    int A, B, Z;
    int rev;
    float P1;
};

__device__ __forceinline__ float ubyte_f32(unsigned w, int j)
{
    // v_cvt_f32_ubyteN
    return (float)((w >> (8 * j)) & 0xffu);
}

// Path costs are non-negative, finite fp32 values: their bit patterns order like unsigned integers, so every min of the
// recurrence is an integer min on the bits — no NaN canonicalisation, and the DPP lane permutes fold into v_min_u32_dpp
// (old = 0xffffffff is the identity of min_u32, which is what lets the compiler fold row_mask-ed broadcasts too).
__device__ __forceinline__ unsigned fbits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __uint_as_float(v); }
__device__ __forceinline__ unsigned wave_min_bits(unsigned v)
{
    v = min(v, dpp_u32<0x111>(0xffffffffu, v));
    v = min(v, dpp_u32<0x112>(0xffffffffu, v));
    v = min(v, dpp_u32<0x114>(0xffffffffu, v));
    v = min(v, dpp_u32<0x118>(0xffffffffu, v));
    v = min(v, dpp_u32<0x142, 0xa>(0xffffffffu, v));
    v = min(v, dpp_u32<0x143, 0xc>(0xffffffffu, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// NW dwords (4 planes each) per lane; K = index of the path (the running average weight); FULL: Z == 256 * NW, i.e. every
// lane owns 4 * NW valid planes and every access is a whole dword (the production shapes); otherwise ragged tails are
// handled with byte masks (read-modify-write of the partially valid dword, padding planes z >= Z are left untouched).
template <int NW, int K, bool FULL>
__global__ void __launch_bounds__(64) sgm_path_kernel(SgmPathArgs S)
{
    constexpr int ZL = 4 * NW;
    constexpr int PF = NW == 1 ? 8 : (NW == 2 ? 4 : 2); // slices per ring slot
    constexpr int NSETS = 4;                             // ring slots: loads run (NSETS - 1) * PF .. NSETS * PF slices ahead
    constexpr bool LOAD_OUT = (K > 0) || !FULL;
    const int lane = threadIdx.x;
    const int a = blockIdx.x;
    const int Z = S.Z;
    const int z0 = lane * ZL;
    const int nSteps = S.B - 1; // ib = 1 .. B-1

    unsigned offw[NW];
    bool wAny[NW];
    unsigned vmask[NW];
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const int zw = z0 + 4 * w;
        const int nv = FULL ? 4 : min(max(Z - zw, 0), 4);
        wAny[w] = nv > 0;
        vmask[w] = nv >= 4 ? 0xffffffffu : ((1u << (8 * (nv & 3))) - 1u);
        offw[w] = (unsigned)(wAny[w] ? zw : ((Z - 1) & ~3)); // lanes past the last plane re-read the last valid dword (never stored)
    }

    const uint8_t* __restrict__ inCol = S.in + (long long)a * S.strideA;
    uint8_t* __restrict__ outCol = S.out + (long long)a * S.strideA;

    // ---- slice 0: prev = in(b = 0) (always slice 0, also for the reverse path), out(b = 0) = 255 ----
    unsigned prev[ZL]; // bit patterns of non-negative floats
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const unsigned v = *reinterpret_cast<const unsigned*>(inCol + offw[w]);
#pragma unroll
        for(int j = 0; j < 4; ++j)
            prev[4 * w + j] = (FULL || ((vmask[w] >> (8 * j)) & 1u)) ? fbits(ubyte_f32(v, j)) : fbits(SGM_BIG);
        if(FULL)
            *reinterpret_cast<unsigned*>(outCol + offw[w]) = 0xffffffffu;
        else if(wAny[w])
        {
            const unsigned old = *reinterpret_cast<const unsigned*>(outCol + offw[w]);
            *reinterpret_cast<unsigned*>(outCol + offw[w]) = old | vmask[w];
        }
    }
    if(nSteps <= 0)
        return;

    // uniform slice pointers, advanced by one slice per step (scalar 64-bit adds; the lane part is a 32-bit offset)
    const long long dirStride = S.rev ? -S.strideB : S.strideB;
    const long long firstOff = (long long)(S.rev ? S.B - 2 : 1) * S.strideB; // slice of ib = 1
    const uint8_t* inLoad = inCol + firstOff;
    const uint8_t* outLoad = outCol + firstOff;
    uint8_t* outStore = outCol + firstOff;
    int ibLoad = 1;

    unsigned rin[NSETS][PF][NW], rout[NSETS][PF][NW];

    auto load_group = [&](unsigned (&ri)[PF][NW], unsigned (&ro)[PF][NW]) {
#pragma unroll
        for(int t = 0; t < PF; ++t)
        {
#pragma unroll
            for(int w = 0; w < NW; ++w)
            {
                ri[t][w] = *reinterpret_cast<const unsigned*>(inLoad + offw[w]);
                if(LOAD_OUT)
                    ro[t][w] = *reinterpret_cast<const unsigned*>(outLoad + offw[w]);
            }
            if(ibLoad < nSteps) // past the end: keep re-reading the last slice (harmless)
            {
                inLoad += dirStride;
                outLoad += dirStride;
                ++ibLoad;
            }
        }
    };

    auto load_p2 = [&](int blk) -> float {
        const int ib = min(blk * 64 + 1 + lane, nSteps);
        return S.p2[(long long)a * S.B + (S.rev ? S.B - ib : ib)];
    };

    auto step = [&](int ib, const unsigned (&inw)[NW], const unsigned (&oldw)[NW], float p2vec) {
        const float P2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2vec), (ib - 1) & 63));

        // best cost of the previous slice over all planes (computeBestZInSlice)
        unsigned m = prev[0];
#pragma unroll
        for(int i = 1; i < ZL; ++i)
            m = min(m, prev[i]);
        const float best = bitsf(wave_min_bits(m));
        const unsigned bestP2 = fbits(best + P2);

        // z-1 / z+1 neighbours across lanes (wave_shr:1 / wave_shl:1), folded into the min with the in-lane neighbour
        const unsigned nbLo = min(dpp_u32<0x138>(0xffffffffu, prev[ZL - 1]), prev[1]);      // min(prev[z0-1], prev[z0+1])
        const unsigned nbHi = min(dpp_u32<0x130>(0xffffffffu, prev[0]), prev[ZL - 2]);      // min(prev[z0+ZL], prev[z0+ZL-2])

        unsigned nprev[ZL];
#pragma unroll
        for(int w = 0; w < NW; ++w)
        {
            unsigned neww = 0;
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                const int i = 4 * w + j;
                const float cur = ubyte_f32(inw[w], j);
                const unsigned nb = (i == 0) ? nbLo : ((i == ZL - 1) ? nbHi : min(prev[i - 1], prev[i + 1]));
                // fminf(fminf(fminf(p, pm1 + P1), pp1 + P1), best + P2): x -> x + P1 is monotone, so the two middle terms are
                // min(pm1, pp1) + P1; all operands are non-negative floats -> integer min on the bits
                const unsigned minCost = min(min(prev[i], fbits(bitsf(nb) + S.P1)), bestP2);
                float pathCost = (cur + bitsf(minCost)) - best;
                // planes 0 and Z-1 are forced to 255 (kernels.cuh:692-730)
                if(FULL)
                {
                    if(i == 0)
                        pathCost = (lane == 0) ? 255.0f : pathCost;
                    if(i == ZL - 1)
                        pathCost = (lane == 63) ? 255.0f : pathCost;
                }
                else
                    pathCost = ((z0 + i == 0) || (z0 + i >= Z - 1)) ? 255.0f : pathCost;
                const float tr = truncf(pathCost); // TSimAcc(pathCost): float -> uint32 truncation (pathCost >= 0)
                nprev[i] = (FULL || ((vmask[w] >> (8 * j)) & 1u)) ? fbits(tr) : fbits(SGM_BIG);
                float q; // integer-valued float in [0, 255]: the byte to store
                if(K == 0)
                    q = __builtin_amdgcn_fmed3f(tr, 0.0f, 255.0f); // trunc(clamp(x)) == clamp(trunc(x))
                else
                {
                    const float lc = __builtin_amdgcn_fmed3f(pathCost, 0.0f, 255.0f);
                    const float n = fmaf(ubyte_f32(oldw[w], j), (float)K, lc); // o*K is exact: == fl(fl(o*K) + lc)
                    // (uint8)(n / (K+1)) == floor(floor(n) / (K+1)) for every fp32 n in [0, 1021) (exhaustively checked)
                    if(K == 1)
                        q = truncf(n * 0.5f);
                    else if(K == 3)
                        q = truncf(n * 0.25f);
                    else
                        q = truncf(truncf(n) * 0.33333334f);
                }
                neww = __builtin_amdgcn_cvt_pk_u8_f32(q, j, neww);
            }
            unsigned* po = reinterpret_cast<unsigned*>(outStore + offw[w]);
            if(FULL)
                *po = neww;
            else if(wAny[w])
                *po = (neww & vmask[w]) | (oldw[w] & ~vmask[w]);
        }
        outStore += dirStride;
#pragma unroll
        for(int i = 0; i < ZL; ++i)
            prev[i] = nprev[i];
    };

#pragma unroll
    for(int s = 0; s < NSETS; ++s)
        load_group(rin[s], rout[s]);
    float p2vec = load_p2(0);
    float p2next = load_p2(1);

    const int nGroups = (nSteps + PF - 1) / PF;
    for(int G = 0; G < nGroups; G += NSETS)
    {
        if(G > 0 && ((G * PF) & 63) == 0)
        {
            p2vec = p2next;
            p2next = load_p2((G * PF) / 64 + 1);
        }
#pragma unroll
        for(int s = 0; s < NSETS; ++s)
        {
            const int g = G + s;
            if(g >= nGroups)
                break;
            if(g * PF + PF <= nSteps)
            {
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    step(g * PF + 1 + t, rin[s][t], rout[s][t], p2vec);
            }
            else
            {
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    if(g * PF + 1 + t <= nSteps)
                        step(g * PF + 1 + t, rin[s][t], rout[s][t], p2vec);
            }
            load_group(rin[s], rout[s]);
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
static void launch_path(const SgmPathArgs& S, int K, bool full, hipStream_t st)
{
    dim3 grid(S.A);
#define AVDM_SGM_LAUNCH(KK)                                                                                                                           \
    if(full)                                                                                                                                          \
        hipLaunchKernelGGL((sgm_path_kernel<NW, KK, true>), grid, dim3(64), 0, st, S);                                                                \
    else                                                                                                                                              \
        hipLaunchKernelGGL((sgm_path_kernel<NW, KK, false>), grid, dim3(64), 0, st, S)
    switch(K)
    {
        case 0: AVDM_SGM_LAUNCH(0); break;
        case 1: AVDM_SGM_LAUNCH(1); break;
        case 2: AVDM_SGM_LAUNCH(2); break;
        default: AVDM_SGM_LAUNCH(3); break;
    }
#undef AVDM_SGM_LAUNCH
}

} // namespace avdm

using namespace avdm;

extern "C" {

size_t avdm_volume_optimize_scratch_bytes(int dimX, int dimY, int dimZ)
{
    (void)dimZ;
    if(dimX <= 0 || dimY <= 0)
        return 0;
    // one fp32 adaptive-P2 map per axis pass (the path costs of the previous slice live in registers: no uint32 slice
    // buffers like Sgm.hpp:144-148 of the reference)
    return ((size_t)dimX * (size_t)dimY * sizeof(float) + 255) & ~(size_t)255;
}

int avdm_volume_optimize(uint8_t* out_vol, const uint8_t* in_vol, long long pitch_y, int pitch_x, void* scratch, const avdm_pyramid_t* rc_pyr,
                         const avdm_sgm_params_t* sp, int last_depth_index, avdm_roi_t roi, void* stream)
{
    const int dimX = (int)(roi.x.end - roi.x.begin), dimY = (int)(roi.y.end - roi.y.begin), Z = last_depth_index;
    if(dimX <= 0 || dimY <= 0 || Z <= 0)
        return 0;
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)out_vol & 3) || ((uintptr_t)in_vol & 3))
        return set_error_msg(1, "avdm_volume_optimize: volume base / pitches must be multiples of 4 bytes");
    if(((Z + 3) & ~3) > pitch_x)
        return set_error_msg(1, "avdm_volume_optimize: pitch_x must cover the 4-aligned depth count");
    if(Z > 1536)
        return set_error_msg(1, "avdm_volume_optimize: more than 1536 depth planes are not supported");
    if(scratch == nullptr || ((uintptr_t)scratch & 3))
        return set_error_msg(1, "avdm_volume_optimize: scratch of avdm_volume_optimize_scratch_bytes() bytes (4-byte aligned) is required");
    int level;
    if(!lod_is_integral(rc_pyr, sp->scale, &level))
        return set_error_msg(1, "avdm_volume_optimize: non-integral mip level");
    const Tex t = make_tex(rc_pyr);
    hipStream_t st = (hipStream_t)stream;

    SgmP2Args Q;
    Q.L = t.lv[level];
    Q.rcW = (float)tex_dim_w(rc_pyr, sp->scale);
    Q.rcH = (float)tex_dim_h(rc_pyr, sp->scale);
    Q.step = (float)sp->stepXY;
    Q.P2w = (float)sp->p2Weighting;
    const bool fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;

    SgmPathArgs S;
    S.in = in_vol;
    S.out = out_vol;
    S.p2 = (const float*)scratch;
    S.Z = Z;
    S.P1 = (float)sp->p1;

    const int NW = (Z + 255) / 256;
    const bool full = (Z == 256 * NW);
    int npaths = 0;
    for(const char* ax = sp->filteringAxes; *ax; ++ax)
    {
        if(*ax != 'X' && *ax != 'Y')
            continue;
        if(npaths > 2)
            return set_error_msg(1, "avdm_volume_optimize: at most 2 filtering axes");
        const bool scanX = (*ax == 'X');
        S.A = scanX ? dimY : dimX;
        S.B = scanX ? dimX : dimY;
        S.strideA = scanX ? pitch_y : (long long)pitch_x;
        S.strideB = scanX ? (long long)pitch_x : pitch_y;
        // deviceSimilarityVolumeKernels.cuh:688-689: beginX = (axisT.x == 0) ? roi.x.begin : roi.y.begin, applied to v.x (sic)
        const bool swap = sp->strictRoiQuirk && scanX;
        Q.scanIsX = scanX ? 1 : 0;
        Q.beginX = swap ? (int)roi.y.begin : (int)roi.x.begin;
        Q.beginY = swap ? (int)roi.x.begin : (int)roi.y.begin;
        Q.A = S.A;
        Q.B = S.B;
        const dim3 pgrid(divUp(S.B, 16), divUp(S.A, 16));
        if(fixed8)
            hipLaunchKernelGGL(sgm_p2_map_kernel<true>, pgrid, dim3(256), 0, st, (float*)scratch, Q);
        else
            hipLaunchKernelGGL(sgm_p2_map_kernel<false>, pgrid, dim3(256), 0, st, (float*)scratch, Q);
        for(int rev = 0; rev < 2; ++rev)
        {
            S.rev = rev;
            const int K = npaths++;
            switch(NW)
            {
                case 1: launch_path<1>(S, K, full, st); break;
                case 2: launch_path<2>(S, K, full, st); break;
                case 3: launch_path<3>(S, K, full, st); break;
                case 4: launch_path<4>(S, K, full, st); break;
                case 5: launch_path<5>(S, K, full, st); break;
                default: launch_path<6>(S, K, full, st); break;
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
