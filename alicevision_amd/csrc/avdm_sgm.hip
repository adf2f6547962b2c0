// avdm_sgm.hip — SGM path aggregation of the uint8 cost volume + winner-take-all depth retrieval, for gfx950.
//   avdm_volume_optimize[_tiles]    <-> cuda_volumeOptimize / cuda_volumeAggregatePath (planeSweeping/deviceSimilarityVolume.cu:262-425;
//                                       kernels planeSweeping/deviceSimilarityVolumeKernels.cuh:596-744)
//   avdm_volume_retrieve_best_depth <-> cuda_volumeRetrieveBestDepth (deviceSimilarityVolume.cu:427-467; kernels.cuh:393-512)
//
// This translation unit is compiled with -ffp-contract=off: the aggregation is integer-valued except for the adaptive P2,
// and every fp32 operation below is written in the reference's order so that the stage is BIT-EXACT against the oracle.
//
// CDNA4 design (the reference issues ~3 tiny kernels per slice, ~10^4 launches per volume, through uint32 slice copies):
//   * ONE launch per path for ALL tiles of a batch.  One wave64 (= one workgroup) owns one column (fixed position on the
//     non-scanned image axis of one tile) and walks the scanned axis as a persistent loop; the previous-slice path costs L(z)
//     never leave VGPRs.  Columns are the only parallelism of this recurrence (~1000 per path for an undivided 12 MP frame, one
//     wave per SIMD); batching the tiles of a frame (DepthMapEstimator's tile list) is what puts several waves on every SIMD.
//   * z-fastest volume: a lane owns 4*NW consecutive planes, so a step is one coalesced 256*NW-byte read of the input
//     volume, (for paths 1..3) one of the output volume, and one coalesced write.  Algorithmic traffic only: 11 B/voxel total.
//   * the plain VALU issues one wave instruction per 4 cycles per SIMD, so the step is written for instruction count: path
//     costs are carried as PACKED uint16 pairs (v_pk_min_u16 / v_pk_add_u16 / v_pk_mad_u16: two planes per instruction).
//     Everything in the recurrence is an integer except the adaptive P2 (a float in [80, 255]): with frac(P2) < 1 - 2^-13
//     no fp32 rounding of the reference's expression can carry into the integer part (proof in DESIGN.md), so
//         trunc(L) = cur + min(min(prev, min(prev[z-1], prev[z+1]) + P1), best + floor(P2)) - best
//         out      = (out*K + min(trunc(L), 255)) div (K + 1)
//     exactly; the (rare, wave-uniform) steps whose P2 fraction is closer to 1 run the fp32 restatement instead.
//   * min over z = in-lane packed min + 6 v_min_u32_dpp on the replicated pair (row_shr / row_bcast) + v_readlane;
//     z±1 neighbours = v_alignbit on adjacent pairs, wave_shr/shl DPP across lanes.
//   * the colour-adaptive P2 is evaluated once per (column, slice) by a small map kernel per axis (one map serves the forward
//     and the reverse path); the path kernel reads 64 consecutive steps of it per lane-coalesced load and picks with v_readlane.
//   * a 4-slot register ring keeps the loads of the next 24-32 slices in flight; the recurrence depends on registers only.
#include "avdm_device.h"

#include <hip/hip_ext.h>

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <mutex>
#include <vector>

namespace avdm {

#define AVDM_SGM_MAX_TILES 24 // tiles per launch (kernarg budget); larger batches are split

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
struct SgmP2Tile
{
    TexLod L;           // R image at the SGM mip level (one level, or two blended when the level is fractional)
    float rcW, rcH;     // nominal level dims (DeviceMipmapImage::getDimensions)
    int beginX, beginY; // ROI offsets as the reference applies them to (v.x, v.y)
    int A, B;
    int scanIsX;        // 1: the scanned axis is volume x (axisT.y == 0), 0: volume y
    float* p2;          // [A][B]
};
struct SgmP2Batch
{
    float step;
    float P2w;
    int fixed8;
    SgmP2Tile t[AVDM_SGM_MAX_TILES];
};

#define SGM_BIG 3.0e38f

// P2 of the FORWARD path at slice b of column a: colour step between stage pixel b and b - 1 along the scanned axis.
// The reverse path at slice b compares pixel b with b + 1 — the same texel pair as the forward path at b + 1, with the
// two fetches swapped; deltaC is a sum of squared differences, so the value is bit-identical and ONE map serves both
// directions of an axis (reverse reads entry b + 1).
__global__ void __launch_bounds__(256) sgm_p2_map_kernel(SgmP2Batch S)
{
    const SgmP2Tile& T = S.t[blockIdx.z];
    const int b = blockIdx.x * 16 + (threadIdx.x & 15);
    const int a = blockIdx.y * 16 + (threadIdx.x >> 4);
    if(a >= T.A || b >= T.B)
        return;
    float P2;
    if(S.P2w < 0)
        P2 = fabsf(S.P2w);
    else if(b == 0)
        P2 = 80.0f; // never read
    else
    {
        const int vx = T.scanIsX ? b : a, vy = T.scanIsX ? a : b;
        const int imX0 = (int)((float)(T.beginX + vx) * S.step);
        const int imY0 = (int)((float)(T.beginY + vy) * S.step);
        const int imX1 = (int)((float)imX0 - S.step * (float)(T.scanIsX ? 1 : 0));
        const int imY1 = (int)((float)imY0 - S.step * (float)(T.scanIsX ? 0 : 1));
        const float u0 = ((float)imX0 + 0.5f) / T.rcW, v0 = ((float)imY0 + 0.5f) / T.rcH;
        const float u1 = ((float)imX1 + 0.5f) / T.rcW, v1 = ((float)imY1 + 0.5f) / T.rcH;
        const float4 c0 = S.fixed8 ? tex2D_lod<true>(T.L, u0, v0) : tex2D_lod<false>(T.L, u0, v0);
        const float4 c1 = S.fixed8 ? tex2D_lod<true>(T.L, u1, v1) : tex2D_lod<false>(T.L, u1, v1);
        const float dx = c0.x - c1.x, dy = c0.y - c1.y, dz = c0.z - c1.z;
        const float deltaC = sqrtf(dx * dx + dy * dy + dz * dz);
        P2 = 80.f + (255.f - 80.f) * (1.0f / (1.0f + exp_p2(10.0f * ((deltaC - S.P2w) / 80.f))));
    }
    T.p2[(long long)a * T.B + b] = P2;
}

// The maps of BOTH filtering axes of a tile in one pass over its pixels.  A lane owns stage pixel (x, y): the colour step to its left
// neighbour is the X-axis map's entry [y][x], the one to the neighbour above the Y-axis map's [x][y] — three texel fetches instead of
// four (the centre is shared whenever both axes apply the same ROI offsets, i.e. always but in strictRoiQuirk mode on an offset tile),
// lanes along x for both fetch patterns, and the Y-axis map — whose rows run along y — stored through a 16 x 16 LDS transpose so that
// its stores are runs of 16 floats too.  Same expressions as sgm_p2_map_kernel, entry by entry (the A/B is AVDM_SGM_P2_MAP=legacy).
struct SgmP2Tile2
{
    TexLod L;
    float rcW, rcH;
    int dimX, dimY;
    int beginX[2], beginY[2]; // per filtering axis (slot): ROI offsets as the reference applies them to (v.x, v.y)
    int scanIsX[2];
    float* p2[2];             // slot's map: [dimY][dimX] when it scans along x, [dimX][dimY] when along y; nullptr: no such slot
};
struct SgmP2Batch2
{
    float step;
    float P2w;
    int fixed8;
    SgmP2Tile2 t[AVDM_SGM_MAX_TILES];
};

__device__ __forceinline__ float4 sgm_p2_texel(const SgmP2Batch2& S, const SgmP2Tile2& T, int imX, int imY)
{
    const float u = ((float)imX + 0.5f) / T.rcW, v = ((float)imY + 0.5f) / T.rcH;
    return S.fixed8 ? tex2D_lod<true>(T.L, u, v) : tex2D_lod<false>(T.L, u, v);
}

__global__ void __launch_bounds__(256) sgm_p2_map2_kernel(SgmP2Batch2 S)
{
    __shared__ float tile[16][17];
    const SgmP2Tile2& T = S.t[blockIdx.z];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + tx, y = blockIdx.y * 16 + ty;
    if((int)(blockIdx.x * 16) >= T.dimX || (int)(blockIdx.y * 16) >= T.dimY)
        return; // (whole workgroup outside this tile: the grid is sized for the largest tile of the batch)
    const bool inside = x < T.dimX && y < T.dimY;
    const bool shared0 = T.p2[0] != nullptr && T.p2[1] != nullptr && T.beginX[0] == T.beginX[1] && T.beginY[0] == T.beginY[1];
    float4 c0shared = make_float4(0.f, 0.f, 0.f, 0.f);
    if(inside && shared0 && !(S.P2w < 0))
        c0shared = sgm_p2_texel(S, T, (int)((float)(T.beginX[0] + x) * S.step), (int)((float)(T.beginY[0] + y) * S.step));
    for(int s = 0; s < 2; ++s)
    {
        if(T.p2[s] == nullptr)
            continue;
        float P2 = 80.0f;
        if(inside)
        {
            const int b = T.scanIsX[s] ? x : y;
            if(S.P2w < 0)
                P2 = fabsf(S.P2w);
            else if(b == 0)
                P2 = 80.0f; // never read
            else
            {
                const int imX0 = (int)((float)(T.beginX[s] + x) * S.step);
                const int imY0 = (int)((float)(T.beginY[s] + y) * S.step);
                const int imX1 = (int)((float)imX0 - S.step * (float)(T.scanIsX[s] ? 1 : 0));
                const int imY1 = (int)((float)imY0 - S.step * (float)(T.scanIsX[s] ? 0 : 1));
                const float4 c0 = shared0 ? c0shared : sgm_p2_texel(S, T, imX0, imY0);
                const float4 c1 = sgm_p2_texel(S, T, imX1, imY1);
                const float dx = c0.x - c1.x, dy = c0.y - c1.y, dz = c0.z - c1.z;
                const float deltaC = sqrtf(dx * dx + dy * dy + dz * dz);
                P2 = 80.f + (255.f - 80.f) * (1.0f / (1.0f + exp_p2(10.0f * ((deltaC - S.P2w) / 80.f))));
            }
        }
        if(T.scanIsX[s])
        {
            if(inside)
                T.p2[s][(long long)y * T.dimX + x] = P2;
        }
        else
        {
            __syncthreads(); // (the tile may still be read by the previous slot)
            tile[ty][tx] = P2;
            __syncthreads();
            const int ox = blockIdx.x * 16 + ty, oy = blockIdx.y * 16 + tx; // roles swapped: consecutive lanes = consecutive y
            if(ox < T.dimX && oy < T.dimY)
                T.p2[s][(long long)ox * T.dimY + oy] = tile[tx][ty];
        }
    }
}

// ---- one aggregation path over a batch of tiles -----------------------------------------------------------------------
struct SgmPathTile
{
    const uint8_t* in;
    uint8_t* out;
    const float* p2;            // [A][B], see sgm_p2_map_kernel
    long long strideA, strideB; // bytes between consecutive columns / consecutive slices
    int A, B, Z;
    int colEnd;                 // exclusive prefix sum of A over the batch: workgroups [colEnd[t-1], colEnd[t]) belong to tile t
    uint8_t* tmp;               // pair kernel only: scratch volume (tightly packed, z-fastest) for the reverse path's costs
    long long tStrideA, tStrideB;
};
struct SgmPathBatch
{
    int rev;
    float P1;
    SgmPathTile t[AVDM_SGM_MAX_TILES];
};

__device__ __forceinline__ float ubyte_f32(unsigned w, int j)
{
    // v_cvt_f32_ubyteN
    return (float)((w >> (8 * j)) & 0xffu);
}

// Path costs are non-negative, finite fp32 values: their bit patterns order like unsigned integers, so every min of the
// fp32 restatement is an integer min on the bits — no NaN canonicalisation, and the DPP lane permutes fold into v_min_u32_dpp
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

// per-lane constants of a column walk
template <int NW>
struct SgmLane
{
    unsigned offw[NW];  // byte offset of each of my dwords inside a (column, slice) run of planes
    bool wAny[NW];      // dword holds at least one valid plane
    unsigned vmask[NW]; // 0xff per valid plane of the dword
    int lane, z0, Z;
};

// One step of the fp32 restatement (the reference's expression, operation by operation).  prev = fp32 bit patterns.
template <int NW, int K, bool FULL>
__device__ __forceinline__ void sgm_step_f32(unsigned (&prev)[4 * NW], const unsigned (&inw)[NW], const unsigned (&oldw)[NW], float P2, float P1,
                                             const SgmLane<NW>& Ln, uint8_t* outSlice)
{
    constexpr int ZL = 4 * NW;
    // best cost of the previous slice over all planes (computeBestZInSlice)
    unsigned m = prev[0];
#pragma unroll
    for(int i = 1; i < ZL; ++i)
        m = min(m, prev[i]);
    const float best = bitsf(wave_min_bits(m));
    const unsigned bestP2 = fbits(best + P2);

    // z-1 / z+1 neighbours across lanes (wave_shr:1 / wave_shl:1), folded into the min with the in-lane neighbour
    const unsigned nbLo = min(dpp_u32<0x138>(0xffffffffu, prev[ZL - 1]), prev[1]); // min(prev[z0-1], prev[z0+1])
    const unsigned nbHi = min(dpp_u32<0x130>(0xffffffffu, prev[0]), prev[ZL - 2]); // min(prev[z0+ZL], prev[z0+ZL-2])

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
            const unsigned minCost = min(min(prev[i], fbits(bitsf(nb) + P1)), bestP2);
            float pathCost = (cur + bitsf(minCost)) - best;
            // planes 0 and Z-1 are forced to 255 (kernels.cuh:692-730)
            if(FULL)
            {
                if(i == 0)
                    pathCost = (Ln.lane == 0) ? 255.0f : pathCost;
                if(i == ZL - 1)
                    pathCost = (Ln.lane == 63) ? 255.0f : pathCost;
            }
            else
                pathCost = ((Ln.z0 + i == 0) || (Ln.z0 + i >= Ln.Z - 1)) ? 255.0f : pathCost;
            const float tr = truncf(pathCost); // TSimAcc(pathCost): float -> uint32 truncation (pathCost >= 0)
            nprev[i] = (FULL || ((Ln.vmask[w] >> (8 * j)) & 1u)) ? fbits(tr) : fbits(SGM_BIG);
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
        unsigned* po = reinterpret_cast<unsigned*>(outSlice + Ln.offw[w]);
        if(FULL)
            *po = neww;
        else if(Ln.wAny[w])
            *po = (neww & Ln.vmask[w]) | (oldw[w] & ~Ln.vmask[w]);
    }
#pragma unroll
    for(int i = 0; i < ZL; ++i)
        prev[i] = nprev[i];
}

// ---- packed uint16 helpers (one VGPR = two planes) ----
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return as_u32(__builtin_elementwise_min(as_pk(a), as_pk(b))); }
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return as_u32(as_pk(a) + as_pk(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return as_u32(as_pk(a) - as_pk(b)); }
#define SGM_BIG16 0x3fffu
#define SGM_BIG16_PAIR 0x3fff3fffu

// One step in packed uint16 arithmetic (see the header comment for the equivalence).  P = plane pairs (lo = even plane).
template <int NW, int K, bool FULL>
__device__ __forceinline__ void sgm_step_u16(unsigned (&P)[2 * NW], const unsigned (&inw)[NW], const unsigned (&oldw)[NW], unsigned iP2Pair,
                                             unsigned P1Pair, const unsigned (&keepM)[2 * NW], const unsigned (&forceV)[2 * NW],
                                             const SgmLane<NW>& Ln, uint8_t* outSlice)
{
    constexpr int NR = 2 * NW;
    // best cost of the previous slice: in-lane packed min, both halves replicated, then a u32 min over the wave
    unsigned m = P[0];
#pragma unroll
    for(int r = 1; r < NR; ++r)
        m = pk_min(m, P[r]);
    m = pk_min(m, __builtin_amdgcn_alignbit(m, m, 16));
    const unsigned bestPair = wave_min_bits(m); // (best, best)
    const unsigned FPair = bestPair + iP2Pair;  // (best + floor(P2)) in both halves: no carry between halves (values < 2^15)

    // Lp[r] = (plane 2r-1, plane 2r): adjacent pairs shifted by one plane; the ends come from the neighbour lanes
    unsigned Lp[NR + 1];
    Lp[0] = __builtin_amdgcn_alignbit(P[0], dpp_u32<0x138>(SGM_BIG16_PAIR, P[NR - 1]), 16);
#pragma unroll
    for(int r = 1; r < NR; ++r)
        Lp[r] = __builtin_amdgcn_alignbit(P[r], P[r - 1], 16);
    Lp[NR] = __builtin_amdgcn_alignbit(dpp_u32<0x130>(SGM_BIG16_PAIR, P[0]), P[NR - 1], 16);

    unsigned q[NR];
#pragma unroll
    for(int r = 0; r < NR; ++r)
    {
        const unsigned nb = pk_min(Lp[r], Lp[r + 1]); // min(prev[z-1], prev[z+1]) for both planes of the pair
        const unsigned mF = pk_min(pk_min(P[r], pk_add(nb, P1Pair)), FPair);
        const unsigned cur = __builtin_amdgcn_perm(0u, inw[r >> 1], (r & 1) ? 0x0c030c02u : 0x0c010c00u); // two bytes -> two uint16
        unsigned L = pk_add(cur, pk_sub(mF, bestPair));
        L = (L & keepM[r]) | forceV[r]; // planes 0 / Z-1 -> 255, planes past Z -> BIG
        P[r] = L;
        const unsigned Lc = pk_min(L, 0x00ff00ffu);
        if(K == 0)
            q[r] = Lc;
        else
        {
            const unsigned o = __builtin_amdgcn_perm(0u, oldw[r >> 1], (r & 1) ? 0x0c030c02u : 0x0c010c00u);
            const unsigned n = as_u32(as_pk(o) * (unsigned short)K + as_pk(Lc)); // v_pk_mad_u16, <= 1020
            if(K == 1)
                q[r] = as_u32(as_pk(n) >> (unsigned short)1);
            else if(K == 3)
                q[r] = as_u32(as_pk(n) >> (unsigned short)2);
            else
            { // n div 3 == (n * 683) >> 11 for n <= 1020 (checked exhaustively); 32-bit products
                const unsigned lo = ((n & 0xffffu) * 683u) >> 11, hi = ((n >> 16) * 683u) >> 11;
                q[r] = lo | (hi << 16);
            }
        }
    }
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const unsigned neww = __builtin_amdgcn_perm(q[2 * w + 1], q[2 * w], 0x06040200u); // low bytes of the four uint16
        unsigned* po = reinterpret_cast<unsigned*>(outSlice + Ln.offw[w]);
        if(FULL)
            *po = neww;
        else if(Ln.wAny[w])
            *po = (neww & Ln.vmask[w]) | (oldw[w] & ~Ln.vmask[w]);
    }
}

// NW dwords (4 planes each) per lane; K = index of the path (the running average weight); FULL: Z == 256 * NW, i.e. every
// lane owns 4 * NW valid planes and every access is a whole dword (the production shapes); otherwise ragged tails are
// handled with byte masks (read-modify-write of the partially valid dword, padding planes z >= Z are left untouched).
// INT16: packed uint16 recurrence (integer P1 required) with the fp32 step for the rare P2 fractions close to 1.
#define AVDM_SGM_WPB 4 // waves (= adjacent columns) per workgroup: their loads of a slice form one contiguous run in memory
template <int NW, int K, bool FULL, bool INT16>
__global__ void __launch_bounds__(64 * AVDM_SGM_WPB) sgm_path_kernel(SgmPathBatch S)
{
    constexpr int ZL = 4 * NW;
    constexpr int NR = 2 * NW;
    constexpr int PF = NW == 1 ? 8 : (NW == 2 ? 4 : 2); // slices per ring slot
    constexpr int NSETS = 4;                             // ring slots: loads run (NSETS - 1) * PF .. NSETS * PF slices ahead
    constexpr bool LOAD_OUT = (K > 0) || !FULL;

    // colEnd counts workgroups (AVDM_SGM_WPB columns each, never straddling two tiles)
    int ti = 0;
    while(ti < AVDM_SGM_MAX_TILES - 1 && (int)blockIdx.x >= S.t[ti].colEnd)
        ++ti;
    const SgmPathTile& T = S.t[ti];
    const int a = ((int)blockIdx.x - (ti > 0 ? S.t[ti - 1].colEnd : 0)) * AVDM_SGM_WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if(a >= T.A)
        return; // whole wave
    const int rev = S.rev;
    const int B = T.B;
    const long long strideB = T.strideB;

    SgmLane<NW> Ln;
    Ln.lane = threadIdx.x & 63;
    Ln.Z = T.Z;
    Ln.z0 = Ln.lane * ZL;
    const int nSteps = B - 1; // ib = 1 .. B-1

#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const int zw = Ln.z0 + 4 * w;
        const int nv = FULL ? 4 : min(max(Ln.Z - zw, 0), 4);
        Ln.wAny[w] = nv > 0;
        Ln.vmask[w] = nv >= 4 ? 0xffffffffu : ((1u << (8 * (nv & 3))) - 1u);
        Ln.offw[w] = (unsigned)(Ln.wAny[w] ? zw : ((Ln.Z - 1) & ~3)); // lanes past the last plane re-read the last valid dword (never stored)
    }

    const uint8_t* __restrict__ inCol = T.in + (long long)a * T.strideA;
    uint8_t* __restrict__ outCol = T.out + (long long)a * T.strideA;

    // ---- slice 0: prev = in(b = 0) (always slice 0, also for the reverse path), out(b = 0) = 255 ----
    unsigned prevF[ZL]; // fp32 bit patterns (fp32 kernel)
    unsigned P[NR];     // packed uint16 pairs (INT16 kernel)
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const unsigned v = *reinterpret_cast<const unsigned*>(inCol + Ln.offw[w]);
#pragma unroll
        for(int j = 0; j < 4; ++j)
            prevF[4 * w + j] = (FULL || ((Ln.vmask[w] >> (8 * j)) & 1u)) ? fbits(ubyte_f32(v, j)) : fbits(SGM_BIG);
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
            unsigned pr = __builtin_amdgcn_perm(0u, v, h ? 0x0c030c02u : 0x0c010c00u);
            if(!FULL)
            {
                if(!((Ln.vmask[w] >> (16 * h)) & 1u))
                    pr = (pr & 0xffff0000u) | SGM_BIG16;
                if(!((Ln.vmask[w] >> (16 * h + 8)) & 1u))
                    pr = (pr & 0x0000ffffu) | (SGM_BIG16 << 16);
            }
            P[2 * w + h] = pr;
        }
        if(FULL)
            *reinterpret_cast<unsigned*>(outCol + Ln.offw[w]) = 0xffffffffu;
        else if(Ln.wAny[w])
        {
            const unsigned old = *reinterpret_cast<const unsigned*>(outCol + Ln.offw[w]);
            *reinterpret_cast<unsigned*>(outCol + Ln.offw[w]) = old | Ln.vmask[w];
        }
    }
    if(nSteps <= 0)
        return;

    // packed kernel: per pair, what survives of the computed L (keep) and what is forced (255 on planes 0 / Z-1, BIG past Z)
    unsigned keepM[NR], forceV[NR];
#pragma unroll
    for(int r = 0; r < NR; ++r)
    {
        unsigned keep = 0, force = 0;
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
            const int z = Ln.z0 + 2 * r + h;
            const bool valid = FULL || z < Ln.Z;
            const bool border = (z == 0) || (z >= Ln.Z - 1);
            const unsigned k16 = (valid && !border) ? 0xffffu : 0u;
            const unsigned f16 = !valid ? SGM_BIG16 : (border ? 255u : 0u);
            keep |= k16 << (16 * h);
            force |= f16 << (16 * h);
        }
        keepM[r] = keep;
        forceV[r] = force;
    }
    const unsigned P1Pair = (unsigned)(int)S.P1 * 0x00010001u;

    // uniform slice pointers, advanced by one slice per step (scalar 64-bit adds; the lane part is a 32-bit offset)
    const long long dirStride = rev ? -strideB : strideB;
    const long long firstOff = (long long)(rev ? B - 2 : 1) * strideB; // slice of ib = 1
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
                ri[t][w] = *reinterpret_cast<const unsigned*>(inLoad + Ln.offw[w]);
                if(LOAD_OUT)
                    ro[t][w] = *reinterpret_cast<const unsigned*>(outLoad + Ln.offw[w]);
            }
            // past the end: keep re-reading the last slice (harmless); selects, not a branch (see sgm_pair_kernel)
            const bool more = ibLoad < nSteps;
            const long long dIn = more ? dirStride : 0ll;
            inLoad += dIn;
            outLoad += dIn;
            ibLoad += more ? 1 : 0;
        }
    };

    auto load_p2 = [&](int blk) -> float {
        const int ib = min(blk * 64 + 1 + Ln.lane, nSteps);
        return T.p2[(long long)a * B + (rev ? B - ib : ib)];
    };

    // per 64-step block: P2 of my step, its integer part replicated in both halves, and which steps need the fp32 step
    float p2vec = 0.f;
    unsigned ip2vec = 0;
    unsigned long long riskyMask = 0;
    auto set_p2_block = [&](float v) {
        p2vec = v;
        if(INT16)
        {
            const float fl = floorf(v);
            ip2vec = (unsigned)(int)fl * 0x00010001u;
            // frac(P2) >= 1 - 2^-13 (or a value outside the uint16 budget): fp32 roundings of the reference's expression may carry
            riskyMask = __ballot(!((v - fl) < (1.0f - 1.0f / 8192.0f)) || !(v >= 0.0f) || !(v < 8192.0f));
        }
    };

    // one slice; FAST = packed uint16 step (only when no step of the current 32-step span needs fp32), else the fp32 step
    auto step = [&](auto fastTag, int ib, const unsigned (&inw)[NW], const unsigned (&oldw)[NW]) {
        constexpr bool FAST = decltype(fastTag)::value;
        const int idx = (ib - 1) & 63;
        if(FAST)
        {
            const unsigned iP2Pair = (unsigned)__builtin_amdgcn_readlane((int)ip2vec, idx);
            sgm_step_u16<NW, K, FULL>(P, inw, oldw, iP2Pair, P1Pair, keepM, forceV, Ln, outStore);
        }
        else
        {
            const float P2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2vec), idx));
            if(INT16)
            { // state converted both ways (values are integers <= 510 or BIG)
#pragma unroll
                for(int r = 0; r < NR; ++r)
                {
                    const unsigned lo = P[r] & 0xffffu, hi = P[r] >> 16;
                    prevF[2 * r] = lo >= SGM_BIG16 ? fbits(SGM_BIG) : fbits((float)lo);
                    prevF[2 * r + 1] = hi >= SGM_BIG16 ? fbits(SGM_BIG) : fbits((float)hi);
                }
            }
            sgm_step_f32<NW, K, FULL>(prevF, inw, oldw, P2, S.P1, Ln, outStore);
            if(INT16)
            {
#pragma unroll
                for(int r = 0; r < NR; ++r)
                {
                    const float lo = bitsf(prevF[2 * r]), hi = bitsf(prevF[2 * r + 1]);
                    const unsigned ulo = lo > 60000.0f ? SGM_BIG16 : (unsigned)lo, uhi = hi > 60000.0f ? SGM_BIG16 : (unsigned)hi;
                    P[r] = ulo | (uhi << 16);
                }
            }
        }
        outStore += dirStride;
    };

    const int nGroups = (nSteps + PF - 1) / PF;
    // NSETS * PF consecutive slices, statically unrolled so that every ring register is addressed by name
    auto span = [&](auto fastTag, int G) {
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
                    step(fastTag, g * PF + 1 + t, rin[s][t], rout[s][t]);
            }
            else
            {
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    if(g * PF + 1 + t <= nSteps)
                        step(fastTag, g * PF + 1 + t, rin[s][t], rout[s][t]);
            }
            load_group(rin[s], rout[s]);
        }
    };

#pragma unroll
    for(int s = 0; s < NSETS; ++s)
        load_group(rin[s], rout[s]);
    set_p2_block(load_p2(0));
    float p2next = load_p2(1);

    for(int G = 0; G < nGroups; G += NSETS)
    {
        if(G > 0 && ((G * PF) & 63) == 0)
        {
            set_p2_block(p2next);
            p2next = load_p2((G * PF) / 64 + 1);
        }
        // which of the NSETS * PF (<= 32) steps of this span need the fp32 step (wave-uniform; almost never any)
        const unsigned riskyBits = (unsigned)(riskyMask >> ((G * PF) & 63)) & (NSETS * PF >= 32 ? 0xffffffffu : ((1u << (NSETS * PF)) - 1u));
        if(INT16 && riskyBits == 0u)
            span(std::true_type{}, G);
        else
            span(std::false_type{}, G);
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

// =====================================================================================================================
// PAIR kernel: the forward and the reverse path of one axis in ONE launch, two waves per column.
//
// The four path recurrences only read the INPUT volume; what is ordered is the running average into the output volume:
//     out_k = (out_{k-1} * k + min(L_k, 255)) div (k + 1),   k = 0..3   (k = 2 * axis + direction)
// A voxel of slice s is reached by the forward wave at step s and by the reverse wave at step B-1-s.  Whoever comes
// first cannot always finish the average, so the column is cut at slice M ~ B/2:
//   phase 1   forward wave: slices 1 .. M-1     out = avg_K(out, L_f)              (K = 0: plain store of L_f)
//             reverse wave: slices B-2 .. M     stores L_r — into `out` when K = 0 (the average is symmetric then),
//                                               into the scratch volume otherwise (out still holds out_{K-1})
//   __syncthreads (both waves of a column live in the same workgroup)
//   phase 2   forward wave: slices M .. B-2     out = avg_{K+1}(avg_K(out, L_f), L_r)   (K = 0: (L_f + L_r) >> 1)
//                           slice  B-1          out = avg_K(out, L_f)              (the reverse path never touches it)
//             reverse wave: slices M-1 .. 0     out = avg_{K+1}(out, L_r)          (slice 0 holds the 255 the forward
//                                                                                    path writes first, as in the reference)
// Per voxel the traffic is what the two sequential launches move (2 + 3 bytes for K = 0, 3 + 3 for K = 2): the pair
// kernel halves the launches and DOUBLES the waves in flight (columns are the only parallelism of the recurrence).
// Integer P1 only (the packed uint16 step with its fp32 fallback for P2 fractions close to 1); the sequential kernels remain
// for non-integer P1.
// =====================================================================================================================
enum { SGM_FIRST_FWD = 0, SGM_FIRST_REV = 1, SGM_SECOND_FWD = 2, SGM_SECOND_REV = 3 };

// n div (K+1) on packed uint16 pairs, n <= 1020
template <int KP1>
__device__ __forceinline__ unsigned pk_div(unsigned n)
{
    if(KP1 == 1)
        return n;
    if(KP1 == 2)
        return as_u32(as_pk(n) >> (unsigned short)1);
    if(KP1 == 4)
        return as_u32(as_pk(n) >> (unsigned short)2);
    // n div 3 == (n * 683) >> 11 == (n * 21856) >> 16 for n <= 1020 (exhaustively checked, tests/test_host_cpu.py): the two 32-bit
    // products by SDWA word selects, the two quotients (bits 16..25 of each product) gathered by one v_perm_b32 — 3 instructions per pair
    unsigned lo, hi;
    const unsigned mul = 21856u;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(lo) : "v"(n), "v"(mul));
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(hi) : "v"(n), "v"(mul));
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// (o * K + c) div (K + 1)
template <int K>
__device__ __forceinline__ unsigned pk_avg(unsigned o, unsigned c)
{
    if(K == 0)
        return c;
    return pk_div<K + 1>(as_u32(as_pk(o) * (unsigned short)K + as_pk(c)));
}

// packed uint16 step WITHOUT the output stage: updates P, returns min(L, 255) per plane pair in q
template <int NW>
__device__ __forceinline__ void sgm_lstep_u16(unsigned (&P)[2 * NW], const unsigned (&inw)[NW], unsigned iP2Pair, unsigned P1Pair,
                                              const unsigned (&keepM)[2 * NW], const unsigned (&forceV)[2 * NW], unsigned (&q)[2 * NW])
{
    constexpr int NR = 2 * NW;
    unsigned m = P[0];
#pragma unroll
    for(int r = 1; r < NR; ++r)
        m = pk_min(m, P[r]);
    // min of the two halves in ONE instruction (v_min_u32_sdwa WORD_0 / WORD_1), zero-extended; the wave-wide minimum is
    // replicated into both halves on the scalar unit
    const unsigned mlo = (unsigned)(unsigned short)m, mhi = m >> 16;
    const unsigned bestPair = wave_min_bits(min(mlo, mhi)) * 0x00010001u;
    const unsigned FPair = bestPair + iP2Pair;
    // z-1 / z+1 across lanes: wave_shr:1 / wave_shl:1 with bound_ctrl (out-of-wave lanes read 0).  What lane 0 gets for "plane -1"
    // and lane 63 for the plane past its last one only ever feeds plane 0 and the last plane of the wave, which are forced below
    // (255 on planes 0 / Z-1, BIG past Z): no "old" register to initialise, one v_mov_b32_dpp each
    unsigned Lp[NR + 1];
    Lp[0] = __builtin_amdgcn_alignbit(P[0], (unsigned)__builtin_amdgcn_mov_dpp((int)P[NR - 1], 0x138, 0xf, 0xf, true), 16);
#pragma unroll
    for(int r = 1; r < NR; ++r)
        Lp[r] = __builtin_amdgcn_alignbit(P[r], P[r - 1], 16);
    Lp[NR] = __builtin_amdgcn_alignbit((unsigned)__builtin_amdgcn_mov_dpp((int)P[0], 0x130, 0xf, 0xf, true), P[NR - 1], 16);
#pragma unroll
    for(int r = 0; r < NR; ++r)
    {
        const unsigned nb = pk_min(Lp[r], Lp[r + 1]);
        const unsigned mF = pk_min(pk_min(P[r], pk_add(nb, P1Pair)), FPair);
        const unsigned cur = __builtin_amdgcn_perm(0u, inw[r >> 1], (r & 1) ? 0x0c030c02u : 0x0c010c00u);
        unsigned L = pk_add(cur, pk_sub(mF, bestPair));
        L = (L & keepM[r]) | forceV[r];
        P[r] = L;
        q[r] = pk_min(L, 0x00ff00ffu);
    }
}

// fp32 restatement of one step WITHOUT the output stage (same expression order as sgm_step_f32).  lc = clamp(L, 0, 255) with its
// FRACTION: the reference averages the un-truncated clamped cost into the output volume (kernels.cuh:733-743), and on the steps that
// take this path (frac(P2) within 2^-13 of 1) that fraction can carry into the integer part of the average
template <int NW, bool FULL>
__device__ __forceinline__ void sgm_lstep_f32(unsigned (&prev)[4 * NW], const unsigned (&inw)[NW], float P2, float P1, const SgmLane<NW>& Ln,
                                              float (&lc)[4 * NW])
{
    constexpr int ZL = 4 * NW;
    unsigned m = prev[0];
#pragma unroll
    for(int i = 1; i < ZL; ++i)
        m = min(m, prev[i]);
    const float best = bitsf(wave_min_bits(m));
    const unsigned bestP2 = fbits(best + P2);
    const unsigned nbLo = min(dpp_u32<0x138>(0xffffffffu, prev[ZL - 1]), prev[1]);
    const unsigned nbHi = min(dpp_u32<0x130>(0xffffffffu, prev[0]), prev[ZL - 2]);
    unsigned nprev[ZL];
#pragma unroll
    for(int i = 0; i < ZL; ++i)
    {
        const int w = i >> 2, j = i & 3;
        const float cur = ubyte_f32(inw[w], j);
        const unsigned nb = (i == 0) ? nbLo : ((i == ZL - 1) ? nbHi : min(prev[i - 1], prev[i + 1]));
        const unsigned minCost = min(min(prev[i], fbits(bitsf(nb) + P1)), bestP2);
        float pathCost = (cur + bitsf(minCost)) - best;
        if(FULL)
        {
            if(i == 0)
                pathCost = (Ln.lane == 0) ? 255.0f : pathCost;
            if(i == ZL - 1)
                pathCost = (Ln.lane == 63) ? 255.0f : pathCost;
        }
        else
            pathCost = ((Ln.z0 + i == 0) || (Ln.z0 + i >= Ln.Z - 1)) ? 255.0f : pathCost;
        const float tr = truncf(pathCost);
        nprev[i] = (FULL || ((Ln.vmask[w] >> (8 * j)) & 1u)) ? fbits(tr) : fbits(SGM_BIG);
        lc[i] = __builtin_amdgcn_fmed3f(pathCost, 0.0f, 255.0f);
    }
#pragma unroll
    for(int i = 0; i < ZL; ++i)
        prev[i] = nprev[i];
}
// (uint8)((o * K + lc) / (K + 1)) of the reference (kernels.cuh:741-743) as an integer-valued float; o, K integers, lc in [0, 255] with a
// fraction.  Same expressions as sgm_step_f32 (o * K is exact; the n / 3 form is checked exhaustively)
template <int K>
__device__ __forceinline__ float avg_f32(float o, float lc)
{
    if(K == 0)
        return truncf(lc);
    const float n = fmaf(o, (float)K, lc);
    if(K == 1)
        return truncf(n * 0.5f);
    if(K == 3)
        return truncf(n * 0.25f);
    return truncf(truncf(n) * 0.33333334f);
}

#ifndef AVDM_SGM_PAIR_WPB
#define AVDM_SGM_PAIR_WPB 4 // columns per workgroup: 2 * WPB waves
#endif
#ifndef AVDM_SGM_AUX_LD
#define AVDM_SGM_AUX_LD 2 // cache policy of the ring loads / the stores: 0 = default, 2 = non-temporal (gfx94x/95x aux bit 1)
#endif
#ifndef AVDM_SGM_AUX_ST
#define AVDM_SGM_AUX_ST 0 // plain stores: 2-3 % faster than non-temporal ones on both kinds of box (profiles/r03_w_sgm_ring_variants.txt)
#endif
#ifndef AVDM_SGM_SLOTS_1LD
#define AVDM_SGM_SLOTS_1LD 4 // ring slots of the walks that load one dword per step (the others use 4)
#endif
#ifndef AVDM_SGM_PAIR_PF1
// steps per ring slot of the 256-plane instantiations: 4 slots x 4 steps = loads 12-16 steps ahead.  Twice that depth (8 steps per slot, the
// setting of rounds 1-3) costs nothing on the boxes where this kernel runs at 0.57-0.59 of the HBM peak, but 12-14 % on the other kind
// (0.46 -> 0.52: SQ_WAIT_ANY doubles there with the same instructions at the same clock — more rows in flight than that memory system
// serves well), and half of it (2 steps per slot) is no better: profiles/r03_w_sgm_ring_variants.txt, r03_t_sgm_clock_pmc.txt
#define AVDM_SGM_PAIR_PF1 4
#endif
#ifndef AVDM_SGM_PAIR_NSETS
#define AVDM_SGM_PAIR_NSETS 4
#endif
#ifndef AVDM_SGM_PROLOGUE_DRAIN
#define AVDM_SGM_PROLOGUE_DRAIN 0
#endif
#ifndef AVDM_SGM_USE_BUFFER
#define AVDM_SGM_USE_BUFFER 0 // 1: raw buffer instructions (descriptor + SGPR slice offset), 0: global_load / global_store
#endif
// A copy of a 32-bit lane offset that the optimizer cannot trace back to its definition.  col + sliceOff is wave-uniform (SGPRs); when the
// zero extension of the lane offset stays in the basic block of the access, instruction selection sees (uniform base + zext(i32)) and emits
// global_load_dword v, v_off, s[base:base+1]  — otherwise the extension is hoisted out of the loop and every access pays a 64-bit VALU add
// (v_lshl_add_u64: 3-4 of the ~40 VALU instructions of a step).  One copy (v_mov_b32) per basic block: per ring reload, per group of steps.
__device__ __forceinline__ unsigned opaque_lane_offset(unsigned laneOff)
{
    asm volatile("" : "+v"(laneOff));
    return laneOff;
}
// one dword of slice `sliceOff` (bytes from the column base) at lane offset `laneOff`
__device__ __forceinline__ unsigned ld_slice(__amdgpu_buffer_rsrc_t rsrc, const uint8_t* col, unsigned laneOff, unsigned sliceOff)
{
#if AVDM_SGM_USE_BUFFER
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)laneOff, (int)sliceOff, AVDM_SGM_AUX_LD);
#else
    const uint8_t* base = col + (size_t)sliceOff; // wave-uniform
    asm volatile("" : "+s"(base));                // ... and kept apart from the lane offset (no re-association into (col + lane) + slice)
    // (the asm hides that the pointer is a global one: say so again, or the access becomes a flat_load)
    typedef const __attribute__((address_space(1))) unsigned* gptr_t;
    gptr_t p = (gptr_t)(base + laneOff);
    return AVDM_SGM_AUX_LD ? __builtin_nontemporal_load(p) : *p;
#endif
}
__device__ __forceinline__ void st_slice(unsigned v, __amdgpu_buffer_rsrc_t rsrc, uint8_t* col, unsigned laneOff, unsigned sliceOff)
{
#if AVDM_SGM_USE_BUFFER
    __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)laneOff, (int)sliceOff, AVDM_SGM_AUX_ST);
#else
    uint8_t* base = col + (size_t)sliceOff;
    asm volatile("" : "+s"(base));
    typedef __attribute__((address_space(1))) unsigned* gptr_t;
    gptr_t p = (gptr_t)(base + laneOff);
    if(AVDM_SGM_AUX_ST)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
#endif
}
#define AVDM_SGM_STASH_SLOTS 4
// dynamic LDS of the pair kernel: the stash of the reverse wave's FRACTIONAL clamped costs on its (rare) fp32 steps of phase 1
static size_t pair_kernel_lds_bytes(int NW) { return (size_t)AVDM_SGM_PAIR_WPB * AVDM_SGM_STASH_SLOTS * (256 * NW * sizeof(float) + sizeof(int)); }

template <int NW, int K, bool FULL>
__global__ void __launch_bounds__(128 * AVDM_SGM_PAIR_WPB) sgm_pair_kernel(SgmPathBatch S)
{
    constexpr int ZL = 4 * NW;
    constexpr int NR = 2 * NW;
    constexpr int PF = NW == 1 ? AVDM_SGM_PAIR_PF1 : (NW == 2 ? 4 : 2); // steps per ring slot
    constexpr int NSETS = AVDM_SGM_PAIR_NSETS;                           // ring slots: the loads run (NSETS - 1) * PF .. NSETS * PF steps ahead
    // Stash (LDS): on a step whose P2 fraction is within 2^-13 of 1 the reference's running average sees the un-truncated clamped cost
    // (kernels.cuh:733-743) and its fraction can carry.  Three of the four roles have that cost in registers when they average; the
    // reverse wave of phase 1 only STORES a byte that the forward wave averages in phase 2 — so on such a step it also leaves the 256 * NW
    // floats here, keyed by slice, and the forward wave looks them up when its P2 map says the reverse path's step at that slice was one
    // of those (probability ~1e-4 per step; 4 slots per column: an overflow — 5 such steps in one half column, ~1e-9 — degrades to the
    // integer average of the stored bytes).
    extern __shared__ float pairLds[];
    float* const stashF = pairLds;
    int* const stashSlice = reinterpret_cast<int*>(pairLds + AVDM_SGM_PAIR_WPB * AVDM_SGM_STASH_SLOTS * 256 * NW);

    int ti = 0;
    while(ti < AVDM_SGM_MAX_TILES - 1 && (int)blockIdx.x >= S.t[ti].colEnd)
        ++ti;
    const SgmPathTile& T = S.t[ti];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= AVDM_SGM_PAIR_WPB ? 1 : 0;
    const int a = ((int)blockIdx.x - (ti > 0 ? S.t[ti - 1].colEnd : 0)) * AVDM_SGM_PAIR_WPB + (wv - rev * AVDM_SGM_PAIR_WPB);
    const bool active = a < T.A; // inactive waves only take part in the barrier
    const int B = T.B;
    const long long strideB = T.strideB;
    const int colInWg = wv - rev * AVDM_SGM_PAIR_WPB;
    float* const myStashF = stashF + colInWg * (AVDM_SGM_STASH_SLOTS * 256 * NW);
    int* const myStashSlice = stashSlice + colInWg * AVDM_SGM_STASH_SLOTS;
    int nStashed = 0; // reverse wave, phase 1
    if(rev && (threadIdx.x & 63) < AVDM_SGM_STASH_SLOTS)
        myStashSlice[threadIdx.x & 63] = -1;

    SgmLane<NW> Ln;
    Ln.lane = threadIdx.x & 63;
    Ln.Z = T.Z;
    Ln.z0 = Ln.lane * ZL;
#pragma unroll
    for(int w = 0; w < NW; ++w)
    {
        const int zw = Ln.z0 + 4 * w;
        const int nv = FULL ? 4 : min(max(Ln.Z - zw, 0), 4);
        Ln.wAny[w] = nv > 0;
        Ln.vmask[w] = nv >= 4 ? 0xffffffffu : ((1u << (8 * (nv & 3))) - 1u);
        Ln.offw[w] = (unsigned)(Ln.wAny[w] ? zw : ((Ln.Z - 1) & ~3));
    }
    const int aa = active ? a : 0;
    const uint8_t* __restrict__ inCol = T.in + (long long)aa * T.strideA;
    uint8_t* outCol = T.out + (long long)aa * T.strideA;
    uint8_t* tmpCol = T.tmp + (long long)aa * T.tStrideA;
    const long long tStrideB = T.tStrideB;
    // Raw buffer resources (base = my column, no stride, no range limit): an access is  base + lane offset (VGPR) + slice offset
    // (SGPR soffset), so walking the slices costs scalar adds only — no 64-bit VALU address arithmetic per load / store.
    // The host guarantees that a column spans less than 4 GiB (32-bit soffset).
    const __amdgpu_buffer_rsrc_t rsrcIn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(inCol), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcOut = __builtin_amdgcn_make_buffer_rsrc(outCol, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcTmp = __builtin_amdgcn_make_buffer_rsrc(tmpCol, 0, -1, 0x00020000);

    unsigned prevF[ZL];
    unsigned P[NR];
    unsigned keepM[NR], forceV[NR];
    if(active)
    {
#pragma unroll
        for(int w = 0; w < NW; ++w)
        {
            const unsigned v = *reinterpret_cast<const unsigned*>(inCol + Ln.offw[w]);
#pragma unroll
            for(int h = 0; h < 2; ++h)
            {
                unsigned pr = __builtin_amdgcn_perm(0u, v, h ? 0x0c030c02u : 0x0c010c00u);
                if(!FULL)
                {
                    if(!((Ln.vmask[w] >> (16 * h)) & 1u))
                        pr = (pr & 0xffff0000u) | SGM_BIG16;
                    if(!((Ln.vmask[w] >> (16 * h + 8)) & 1u))
                        pr = (pr & 0x0000ffffu) | (SGM_BIG16 << 16);
                }
                P[2 * w + h] = pr;
            }
            // out(slice 0) = 255: written by the forward wave only; the reverse wave reads it back in its last step (phase 2)
            if(!rev)
            {
                if(FULL)
                    *reinterpret_cast<unsigned*>(outCol + Ln.offw[w]) = 0xffffffffu;
                else if(Ln.wAny[w])
                {
                    const unsigned old = *reinterpret_cast<const unsigned*>(outCol + Ln.offw[w]);
                    *reinterpret_cast<unsigned*>(outCol + Ln.offw[w]) = old | Ln.vmask[w];
                }
            }
        }
#pragma unroll
        for(int r = 0; r < NR; ++r)
        {
            unsigned keep = 0, force = 0;
#pragma unroll
            for(int h = 0; h < 2; ++h)
            {
                const int z = Ln.z0 + 2 * r + h;
                const bool valid = FULL || z < Ln.Z;
                const bool border = (z == 0) || (z >= Ln.Z - 1);
                const unsigned k16 = (valid && !border) ? 0xffffu : 0u;
                const unsigned f16 = !valid ? SGM_BIG16 : (border ? 255u : 0u);
                keep |= k16 << (16 * h);
                force |= f16 << (16 * h);
            }
            keepM[r] = keep;
            forceV[r] = force;
        }
    }
    const float P1f = S.P1;
    const float* __restrict__ p2col = T.p2 + (long long)aa * B;
    const unsigned P1Pair = (unsigned)(int)P1f * 0x00010001u;
    const long long dirStride = rev ? -strideB : strideB;
    const long long tDirStride = rev ? -tStrideB : tStrideB;

    // walk the steps ib0 <= ib < ib1 of my path in the given role (the state P carries over between walks)
    auto walk = [&](auto roleTag, int ib0, int ib1) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(roleTag)::value;
        // which volumes this role reads besides the input, and where it writes
        constexpr bool STORE_TMP = (ROLE == SGM_FIRST_REV) && (K > 0);
        constexpr bool LOAD_OUT = (ROLE == SGM_SECOND_FWD) || (ROLE == SGM_SECOND_REV) || (ROLE == SGM_FIRST_FWD && (K > 0 || !FULL)) ||
                                  (ROLE == SGM_FIRST_REV && K == 0 && !FULL);
        constexpr bool LOAD_TMP = (ROLE == SGM_SECOND_FWD) && (K > 0);
        // ring slots: the waits the compiler derives are bounded by the number of younger LOADS (7 + (slots - 1) * PF * loads per
        // step) and by the 6-bit counter; with one load per step four slots give vmcnt(31), eight give the full vmcnt(63)
        constexpr int NS = (1 + (LOAD_OUT ? 1 : 0) + (LOAD_TMP ? 1 : 0)) * NW == 1 ? AVDM_SGM_SLOTS_1LD : NSETS;
        const int nSteps = ib1 - ib0;
        if(nSteps <= 0)
            return;
        const long long slice0 = rev ? (long long)(B - 1 - ib0) : (long long)ib0; // slice of step ib0
        // 32-bit slice offsets (soffset of the buffer instructions); the reverse path counts down (two's complement adds)
        unsigned inLoad = (unsigned)(slice0 * strideB);
        unsigned tmpLoad = (unsigned)(slice0 * tStrideB);
        unsigned outStore = (unsigned)(slice0 * (STORE_TMP ? tStrideB : strideB));
        const unsigned storeStride = (unsigned)(STORE_TMP ? tDirStride : dirStride);
        int nLoaded = 0;

        unsigned rin[NS][PF][NW], rout[NS][PF][NW], rtmp[NS][PF][NW];
        auto load_group = [&](unsigned (&ri)[PF][NW], unsigned (&ro)[PF][NW], unsigned (&rt)[PF][NW]) __attribute__((always_inline)) {
            unsigned lo[NW];
#pragma unroll
            for(int w = 0; w < NW; ++w)
                lo[w] = opaque_lane_offset(Ln.offw[w]);
#pragma unroll
            for(int t = 0; t < PF; ++t)
            {
#pragma unroll
                for(int w = 0; w < NW; ++w)
                {
                    ri[t][w] = ld_slice(rsrcIn, inCol, lo[w], inLoad);
                    if(LOAD_OUT)
                        ro[t][w] = ld_slice(rsrcOut, outCol, lo[w], inLoad);
                    if(LOAD_TMP)
                        rt[t][w] = ld_slice(rsrcTmp, tmpCol, lo[w], tmpLoad);
                }
                // past the end: keep re-reading the last slice of the walk (harmless).  Written as selects: a branch here splits
                // the loads over basic blocks and makes the waitcnt insertion fall back to vmcnt(0) at the joins
                // scalar min / multiply: a conditional increment (nLoaded += more ? 1 : 0) goes through v_cndmask + v_readfirstlane, and a plain
                // counter lets the optimizer thread the "past the end" case into branches between the loads
                const int nxt = min(nLoaded + 1, nSteps - 1);
                const unsigned adv = (unsigned)(nxt - nLoaded); // 1, or 0 past the end
                inLoad += adv * (unsigned)dirStride;
                tmpLoad += adv * (unsigned)tDirStride;
                nLoaded = nxt;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // adaptive P2 of 64 consecutive steps of this walk, one per lane
        int curBlk = 0; // which 64-step block of the map p2vec / the masks describe
        auto load_p2 = [&](int blk) __attribute__((always_inline)) -> float {
            const int ib = min(ib0 + blk * 64 + Ln.lane, ib1 - 1);
            return p2col[rev ? B - ib : ib];
        };
        float p2vec = 0.f;
        unsigned ip2vec = 0;
        unsigned long long riskyMask = 0, revRiskyMask = 0, ownRiskyMask = 0;
        auto is_risky = [](float v) __attribute__((always_inline)) -> bool {
            return !((v - floorf(v)) < (1.0f - 1.0f / 8192.0f)) || !(v >= 0.0f) || !(v < 8192.0f);
        };
        // the reverse path's step at my slice uses the NEXT entry of the map (its colour step is mine shifted by one): a second vector,
        // prefetched one block ahead like the first (a load consumed at once would drain the prefetch ring every 64 steps)
        auto load_p2_shifted = [&](int blk) __attribute__((always_inline)) -> float {
            const int ib = min(ib0 + blk * 64 + Ln.lane, ib1 - 1);
            return p2col[min(ib + 1, B - 1)];
        };
        auto set_p2_block = [&](float v, float vShifted) __attribute__((always_inline)) {
            p2vec = v;
            const float fl = floorf(v);
            ip2vec = (unsigned)(int)fl * 0x00010001u;
            ownRiskyMask = __ballot(is_risky(v));
            riskyMask = ownRiskyMask;
            if(ROLE == SGM_SECOND_FWD)
            {
                // when the reverse path's step at a slice took the fp32 step, phase 1 left fractional costs in the stash and this step
                // must average in fp32 too
                revRiskyMask = __ballot(is_risky(vShifted));
                riskyMask |= revRiskyMask;
            }
        };

        auto step = [&](auto fastTag, int i, const unsigned (&inw)[NW], const unsigned (&ow)[NW], const unsigned (&tw)[NW],
                        const unsigned (&stLo)[NW]) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fastTag)::value;
            const int idx = i & 63;
            unsigned newWords[NW];
            if(FAST)
            {
                unsigned q[NR];
                const unsigned iP2Pair = (unsigned)__builtin_amdgcn_readlane((int)ip2vec, idx);
                sgm_lstep_u16<NW>(P, inw, iP2Pair, P1Pair, keepM, forceV, q);
                // output stage (integers)
#pragma unroll
                for(int w = 0; w < NW; ++w)
                {
                    unsigned res[2];
#pragma unroll
                    for(int h = 0; h < 2; ++h)
                    {
                        const unsigned c = q[2 * w + h];
                        const unsigned o = LOAD_OUT ? __builtin_amdgcn_perm(0u, ow[w], h ? 0x0c030c02u : 0x0c010c00u) : 0u;
                        if(ROLE == SGM_FIRST_FWD)
                            res[h] = pk_avg<K>(o, c);
                        else if(ROLE == SGM_FIRST_REV)
                            res[h] = c;
                        else if(ROLE == SGM_SECOND_REV)
                            res[h] = pk_avg<K + 1>(o, c);
                        else if(K == 0)
                            res[h] = pk_avg<1>(o, c); // o = L of the reverse path
                        else
                        {
                            const unsigned t = __builtin_amdgcn_perm(0u, tw[w], h ? 0x0c030c02u : 0x0c010c00u);
                            res[h] = pk_avg<K + 1>(pk_avg<K>(o, c), t);
                        }
                    }
                    newWords[w] = __builtin_amdgcn_perm(res[1], res[0], 0x06040200u);
                }
            }
            else
            {
                // the literal fp32 step, and the running average on the un-truncated clamped costs like the reference (kernels.cuh:733-743)
                const bool ownRisky = (ownRiskyMask >> idx) & 1ull;
                const float P2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2vec), idx));
                float lc[ZL];
                if(ownRisky || ROLE != SGM_SECOND_FWD)
                {
#pragma unroll
                    for(int r = 0; r < NR; ++r)
                    {
                        const unsigned lo = P[r] & 0xffffu, hi = P[r] >> 16;
                        prevF[2 * r] = lo >= SGM_BIG16 ? fbits(SGM_BIG) : fbits((float)lo);
                        prevF[2 * r + 1] = hi >= SGM_BIG16 ? fbits(SGM_BIG) : fbits((float)hi);
                    }
                    sgm_lstep_f32<NW, FULL>(prevF, inw, P2, P1f, Ln, lc);
#pragma unroll
                    for(int r = 0; r < NR; ++r)
                    {
                        const float lo = bitsf(prevF[2 * r]), hi = bitsf(prevF[2 * r + 1]);
                        const unsigned ulo = lo > 60000.0f ? SGM_BIG16 : (unsigned)lo, uhi = hi > 60000.0f ? SGM_BIG16 : (unsigned)hi;
                        P[r] = ulo | (uhi << 16);
                    }
                }
                else
                {
                    // only the reverse path's step at this slice was an fp32 step: mine is the integer step, its costs have no fraction
                    unsigned q[NR];
                    const unsigned iP2Pair = (unsigned)__builtin_amdgcn_readlane((int)ip2vec, idx);
                    sgm_lstep_u16<NW>(P, inw, iP2Pair, P1Pair, keepM, forceV, q);
#pragma unroll
                    for(int r = 0; r < NR; ++r)
                    {
                        lc[2 * r] = (float)(q[r] & 0xffffu);
                        lc[2 * r + 1] = (float)(q[r] >> 16);
                    }
                }
                const int slice = rev ? (B - 1 - (ib0 + i)) : (ib0 + i);
                // FIRST_REV: leave the fractional costs for the forward wave (phase 2)
                if(ROLE == SGM_FIRST_REV)
                {
                    if(nStashed < AVDM_SGM_STASH_SLOTS)
                    {
                        float* dst = myStashF + nStashed * (256 * NW) + Ln.z0;
#pragma unroll
                        for(int z = 0; z < ZL; ++z)
                            dst[z] = lc[z];
                        if(Ln.lane == 0)
                            myStashSlice[nStashed] = slice;
                    }
                    ++nStashed;
                }
                // SECOND_FWD: the reverse path's costs of this slice — from the stash when its step was an fp32 step, else the stored bytes
                int slot = -1;
                if(ROLE == SGM_SECOND_FWD && ((revRiskyMask >> idx) & 1ull))
                {
#pragma unroll
                    for(int k = 0; k < AVDM_SGM_STASH_SLOTS; ++k)
                        slot = (myStashSlice[k] == slice) ? k : slot;
                }
#pragma unroll
                for(int w = 0; w < NW; ++w)
                {
                    unsigned neww = 0;
#pragma unroll
                    for(int j = 0; j < 4; ++j)
                    {
                        const int z = 4 * w + j;
                        const float o = LOAD_OUT ? ubyte_f32(ow[w], j) : 0.0f;
                        float q;
                        if(ROLE == SGM_FIRST_FWD)
                            q = avg_f32<K>(o, lc[z]);
                        else if(ROLE == SGM_FIRST_REV)
                            q = truncf(lc[z]);
                        else if(ROLE == SGM_SECOND_REV)
                            q = avg_f32<K + 1>(o, lc[z]);
                        else
                        {
                            // K == 0: `out` holds the reverse path's byte, tmp is unused; K > 0: `out` holds out_{K-1}, tmp the reverse byte
                            float pr = (K == 0) ? o : ubyte_f32(tw[w], j);
                            if(slot >= 0)
                                pr = myStashF[slot * (256 * NW) + Ln.z0 + z];
                            const float first = (K == 0) ? truncf(lc[z]) : avg_f32<K>(o, lc[z]);
                            q = avg_f32<K + 1>(first, pr);
                        }
                        neww = __builtin_amdgcn_cvt_pk_u8_f32(q, j, neww);
                    }
                    newWords[w] = neww;
                }
            }
            // the stores are COMMON code after the integer / fp32 alternatives (like the ring reloads): memory operations inside the
            // alternatives give the wait-counter insertion two histories to merge at the join, and it then waits for the older one
#pragma unroll
            for(int w = 0; w < NW; ++w)
            {
                if(FULL)
                    st_slice(newWords[w], STORE_TMP ? rsrcTmp : rsrcOut, STORE_TMP ? tmpCol : outCol, stLo[w], outStore);
                else if(Ln.wAny[w])
                {
                    if(STORE_TMP)
                        st_slice(newWords[w], rsrcTmp, tmpCol, stLo[w], outStore); // scratch padding is free
                    else
                        st_slice((newWords[w] & Ln.vmask[w]) | (ow[w] & ~Ln.vmask[w]), rsrcOut, outCol, stLo[w], outStore);
                }
            }
            outStore += storeStride;
            __builtin_amdgcn_sched_barrier(0); // keep the steps and the ring loads in source order: the waitcnt values follow it
        };

        const int nGroups = (nSteps + PF - 1) / PF;
        // One group = PF consecutive steps on ring slot s, then the reload of that slot.  The reload is COMMON code after the
        // fast / generic alternatives: the ring registers then have a single definition site per slot, so no copies of just-issued
        // loads appear at control-flow joins (such copies force the wait counter to zero and drain the prefetch ring).
        // MODE 0: all PF steps exist and none takes the fp32 alternative (the hot code: nothing but integer steps and the reload);
        // MODE 1: all PF steps exist, some may take the fp32 alternative; MODE 2: the last, possibly partial groups of a walk.
        // The main loop picks MODE 0 / MODE 1 per SPAN of NS groups, so the hot loop body is one contiguous run of integer steps: with the
        // choice made per group, the (almost never executed) fp32 alternatives sat between every two groups of the hot path and the kernel
        // ran 2.4 x slower than without them (profiles/README.md, r02_e).  In MODE 1 / 2 every path through a step still consumes its ring
        // registers: a path that skipped one would leave a load pending at the join, and the reload below would have to wait for it.
        auto group = [&](auto modeTag, int g, unsigned (&ri)[PF][NW], unsigned (&ro)[PF][NW], unsigned (&rt)[PF][NW]) __attribute__((always_inline)) {
            constexpr int MODE = decltype(modeTag)::value;
            if(MODE == 0)
            {
                unsigned stLo[NW]; // the PF stores of the group are in this one basic block
#pragma unroll
                for(int w = 0; w < NW; ++w)
                    stLo[w] = opaque_lane_offset(Ln.offw[w]);
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    step(std::true_type{}, g * PF + t, ri[t], ro[t], rt[t], stLo);
            }
            else
            {
                const unsigned risky = (unsigned)(riskyMask >> ((g * PF) & 63)) & ((1u << PF) - 1u);
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    if(MODE == 1 || g * PF + t < nSteps)
                    {
                        if((risky >> t) & 1u)
                            step(std::false_type{}, g * PF + t, ri[t], ro[t], rt[t], Ln.offw);
                        else
                            step(std::true_type{}, g * PF + t, ri[t], ro[t], rt[t], Ln.offw);
                    }
            }
            load_group(ri, ro, rt);
        };
#pragma unroll
        for(int s = 0; s < NS; ++s)
            load_group(rin[s], rout[s], rtmp[s]);
        set_p2_block(load_p2(0), ROLE == SGM_SECOND_FWD ? load_p2_shifted(0) : 0.0f);
        float p2next = load_p2(1);
        float p2nextShifted = ROLE == SGM_SECOND_FWD ? load_p2_shifted(1) : 0.0f;
#if AVDM_SGM_PROLOGUE_DRAIN
        // every prologue load has landed before the loop: the loop-header state of the wait counter is then the back edge's alone
        // (slot loaded 3 groups + 7 loads ago -> vmcnt(55)); without this the prologue path (slot loaded 31 ops ago) caps every
        // wait of the steady state at vmcnt(31), half the depth the ring was sized for
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only
#endif
        // main loop: whole spans of NS groups with NO per-group guards — every control-flow path that cannot happen at run
        // time but exists in the CFG (e.g. group 0 -> skip 1..3 -> group 0) makes the waitcnt insertion assume the ring slot was
        // reloaded one group ago and wait for it with vmcnt(7), which drains the ring
        const int nWholeGroups = nSteps / PF;
        int G = 0;
        for(; G + NS <= nWholeGroups; G += NS)
        {
            if(G > 0 && ((G * PF) & 63) == 0)
            {
                curBlk = (G * PF) / 64;
                set_p2_block(p2next, p2nextShifted);
                p2next = load_p2((G * PF) / 64 + 1);
                if(ROLE == SGM_SECOND_FWD)
                    p2nextShifted = load_p2_shifted((G * PF) / 64 + 1);
            }
            const unsigned long long spanBits = (NS * PF >= 64) ? ~0ull : ((1ull << (NS * PF)) - 1ull);
            if(((riskyMask >> ((G * PF) & 63)) & spanBits) == 0ull)
            {
#pragma unroll
                for(int s = 0; s < NS; ++s)
                    group(std::integral_constant<int, 0>{}, G + s, rin[s], rout[s], rtmp[s]);
            }
            else
            {
#pragma unroll
                for(int s = 0; s < NS; ++s)
                    group(std::integral_constant<int, 1>{}, G + s, rin[s], rout[s], rtmp[s]);
            }
        }
        if(G < nGroups) // last span: up to NS - 1 whole groups and a partial one (within one 64-step block of the P2 map: G * PF is a multiple of 32)
        {
            if(G > 0 && ((G * PF) & 63) == 0)
            {
                curBlk = (G * PF) / 64;
                set_p2_block(p2next, p2nextShifted);
            }
#pragma unroll
            for(int s = 0; s < NS; ++s)
                if(G + s < nGroups)
                    group(std::integral_constant<int, 2>{}, G + s, rin[s], rout[s], rtmp[s]);
        }
    };

    const int M = max(1, B / 2);
    if(active && B > 1)
    {
        if(!rev)
            walk(std::integral_constant<int, SGM_FIRST_FWD>{}, 1, M); // slices 1 .. M-1
        else
            walk(std::integral_constant<int, SGM_FIRST_REV>{}, 1, B - M); // slices B-2 .. M
    }
    __syncthreads(); // phase 1 stores of the whole workgroup are visible (same CU, workgroup-scope release / acquire)
    if(active && B > 1)
    {
        if(!rev)
        {
            walk(std::integral_constant<int, SGM_SECOND_FWD>{}, M, B - 1);    // slices M .. B-2
            walk(std::integral_constant<int, SGM_FIRST_FWD>{}, B - 1, B);     // slice B-1: the reverse path never writes it
        }
        else
            walk(std::integral_constant<int, SGM_SECOND_REV>{}, B - M, B);    // slices M-1 .. 0
    }
}

// Opt-in timing of the path kernels alone (avdm_debug_sgm_kernel_timing): a pair of HIP events per launch on the launch stream, read back
// (and recycled) by avdm_debug_sgm_kernel_timing_read.  Off by default: no events, no synchronisation.
// The events are handed to the launch itself (hipExtLaunchKernelGGL: "startEvent / stopEvent track the start / stop time of the kernel
// launch"), so their difference is the kernel's execution — what rocprofv3's kernel trace reports.  Events recorded with hipEventRecord
// before and after the launch also time the two command-processor hops around it: 8-9 us per launch here, 3.6 % of a 0.23 ms kernel
// (profiles/r03_r_*: 475 us by events against 459 us in the trace for the two launches of one volume).  AVDM_SGM_TIMER=record selects that
// older bracketing for comparison.
struct SgmKernelTimer
{
    std::mutex m;
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
    std::vector<int> pendingPath; // index of the first path a pending launch aggregates (0 / 2 for the pair kernel, 0 ... 3 otherwise)
    std::vector<long> pendingCall; // the avdm_volume_optimize* call a pending launch belongs to
    long callId = 0;               // incremented per call that launches path kernels
    double spanMs = 0.0;           // per call: start of its first path kernel -> end of its last one (the launches AND the gaps between them)
    long spans = 0;
    double ms = 0.0;
    long launches = 0;
    double msPath[4] = {0.0, 0.0, 0.0, 0.0};
    long launchesPath[4] = {0, 0, 0, 0};
};
static SgmKernelTimer g_sgmTimer;

static bool sgm_timer_brackets()
{
    static const bool v = [] {
        const char* e = getenv("AVDM_SGM_TIMER");
        return e && e[0] == 'r';
    }();
    return v;
}

struct SgmKernelTimerScope
{
    hipStream_t st;
    int path;
    bool brackets;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    explicit SgmKernelTimerScope(hipStream_t s, int path_ = 0) : st(s), path(path_ & 3), brackets(sgm_timer_brackets())
    {
        std::lock_guard<std::mutex> lock(g_sgmTimer.m);
        if(!g_sgmTimer.enabled)
            return;
        if(!g_sgmTimer.pool.empty())
        {
            ev = g_sgmTimer.pool.back();
            g_sgmTimer.pool.pop_back();
        }
        else if(hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess)
        {
            ev = {nullptr, nullptr};
            return;
        }
        if(brackets)
            (void)hipEventRecord(ev.first, st);
    }
    // the ONE launch of this scope
    template <typename Kernel, typename Arg>
    void launch(Kernel kernel, dim3 grid, dim3 block, size_t lds, const Arg& arg)
    {
        if(ev.first != nullptr && !brackets)
            hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, ev.first, ev.second, 0, arg);
        else
            hipLaunchKernelGGL(kernel, grid, block, lds, st, arg);
    }
    ~SgmKernelTimerScope()
    {
        if(ev.first == nullptr)
            return;
        if(brackets)
            (void)hipEventRecord(ev.second, st);
        std::lock_guard<std::mutex> lock(g_sgmTimer.m);
        g_sgmTimer.pending.push_back(ev);
        g_sgmTimer.pendingPath.push_back(path);
        g_sgmTimer.pendingCall.push_back(g_sgmTimer.callId);
    }
};

template <int NW>
static void launch_pair(const SgmPathBatch& S, int nWorkgroups, int K, bool full, hipStream_t st)
{
    dim3 grid(nWorkgroups), block(128 * AVDM_SGM_PAIR_WPB);
    const size_t lds = pair_kernel_lds_bytes(NW);
    // the whole-dword form (FULL: Z == 256 * NW exactly) only exists for NW = 1, the shape of the headline configuration: for more than 256
    // planes an exact multiple is a coincidence, the byte-masked form costs it a few percent, and every instantiation of this kernel takes
    // ~30 s to compile
    constexpr bool HAS_FULL = NW == 1;
    if(!HAS_FULL)
        full = false;
    static std::once_flag once[64]; // the attribute belongs to the function on ONE device: once per device, not once per process
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [&] {
        // the stash of the widest instantiations exceeds the 64 KB a kernel gets without asking
        if(HAS_FULL)
        {
            (void)hipFuncSetAttribute((const void*)sgm_pair_kernel<NW, 0, HAS_FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)sgm_pair_kernel<NW, 2, HAS_FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        (void)hipFuncSetAttribute((const void*)sgm_pair_kernel<NW, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)sgm_pair_kernel<NW, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    SgmKernelTimerScope timing(st, K);
    if(K == 0)
    {
        if(full)
            timing.launch(sgm_pair_kernel<NW, 0, HAS_FULL>, grid, block, lds, S);
        else
            timing.launch(sgm_pair_kernel<NW, 0, false>, grid, block, lds, S);
    }
    else
    {
        if(full)
            timing.launch(sgm_pair_kernel<NW, 2, HAS_FULL>, grid, block, lds, S);
        else
            timing.launch(sgm_pair_kernel<NW, 2, false>, grid, block, lds, S);
    }
}

template <int NW>
static void launch_path(const SgmPathBatch& S, int ncols, int K, bool full, bool int16, hipStream_t st)
{
    dim3 grid(ncols);
    SgmKernelTimerScope timing(st, K);
#define AVDM_SGM_LAUNCH2(KK, FF, II) timing.launch(sgm_path_kernel<NW, KK, FF, II>, grid, dim3(64 * AVDM_SGM_WPB), 0, S)
    // the fp32 kernel (non-integer P1) is only instantiated in its general form: FULL shapes run it with FULL = false
#define AVDM_SGM_LAUNCH(KK)                                                                                                                           \
    if(full && int16)                                                                                                                                 \
        AVDM_SGM_LAUNCH2(KK, true, true);                                                                                                             \
    else if(int16)                                                                                                                                    \
        AVDM_SGM_LAUNCH2(KK, false, true);                                                                                                            \
    else                                                                                                                                              \
        AVDM_SGM_LAUNCH2(KK, false, false)
    switch(K)
    {
        case 0: AVDM_SGM_LAUNCH(0); break;
        case 1: AVDM_SGM_LAUNCH(1); break;
        case 2: AVDM_SGM_LAUNCH(2); break;
        default: AVDM_SGM_LAUNCH(3); break;
    }
#undef AVDM_SGM_LAUNCH
#undef AVDM_SGM_LAUNCH2
}

static size_t p2_map_bytes(int dimX, int dimY) { return ((size_t)dimX * (size_t)dimY * sizeof(float) + 255) & ~(size_t)255; }
// scratch volume of the pair kernel (reverse-path costs of the second axis): tightly packed, z-fastest, 4-aligned planes
static size_t tmp_vol_bytes(int dimX, int dimY, int dimZ) { return ((size_t)dimX * (size_t)dimY * (size_t)((dimZ + 3) & ~3) + 255) & ~(size_t)255; }

// all tiles of one launch group share (NW, FULL); `idx` lists their positions in `tiles`
// phase: 0 = the adaptive-P2 maps, then the paths (avdm_volume_optimize[_tiles]); 1 = the maps only (avdm_volume_optimize_prepare: they depend
// on nothing but the R pyramid, so a caller computes them beside the similarity sweep); 2 = the paths only, on maps prepared earlier
static int optimize_group(const avdm_sgm_tile_t* tiles, const int* idx, int n, const size_t* p2off, void* scratch, const avdm_sgm_params_t* sp,
                          hipStream_t st, int phase = 0)
{
    const avdm_sgm_tile_t& t0 = tiles[idx[0]];
    const int NW = (t0.last_depth_index + 255) / 256;
    const bool full = (t0.last_depth_index == 256 * NW);
    const float P1 = (float)sp->p1;
    bool int16 = P1 >= 0.0f && P1 <= 8192.0f && P1 == floorf(P1);
    const char* e = getenv("AVDM_SGM_INT16");
    if(e && e[0] == '0')
        int16 = false;
    // forward + reverse path of an axis in one launch (two waves per column); AVDM_SGM_PAIR=0 selects the sequential kernels
    bool pair = int16;
    // a FIXED P2 (p2Weighting < 0) whose fraction is within 2^-13 of 1 makes EVERY step an fp32 step: the pair kernel's stash (sized
    // for the ~1e-4 of the steps an adaptive P2 sends there) cannot hold that; the sequential kernels average in fp32 in place
    if(sp->p2Weighting < 0)
    {
        const float v = fabsf((float)sp->p2Weighting);
        if(!((v - floorf(v)) < (1.0f - 1.0f / 8192.0f)) || !(v < 8192.0f))
            pair = false;
    }
    const char* ep = getenv("AVDM_SGM_PAIR");
    if(ep && ep[0] == '0')
        pair = false;

    SgmP2Batch Q;
    SgmPathBatch S;
    Q.step = (float)sp->stepXY;
    Q.P2w = (float)sp->p2Weighting;
    S.P1 = P1;
    // the filtering axes, in order
    bool axisIsX[2];
    int nAxes = 0;
    for(const char* ax = sp->filteringAxes; *ax; ++ax)
    {
        // the reference looks every character up in a {X, Y} table (std::map::at throws otherwise, deviceSimilarityVolume.cu:393-405)
        if(*ax != 'X' && *ax != 'Y')
            return set_error_msg(1, "avdm_volume_optimize: filteringAxes may only contain 'X' and 'Y'");
        if(nAxes >= 2)
            return set_error_msg(1, "avdm_volume_optimize: at most 2 filtering axes (the reference runs two paths per character for any length)");
        axisIsX[nAxes++] = (*ax == 'X');
    }
    // pass 1: the adaptive-P2 maps.  Default: both axes of every tile in one pass over the tile's pixels (sgm_p2_map2_kernel)
    static const bool legacyP2 = [] {
        const char* e = getenv("AVDM_SGM_P2_MAP");
        return e && e[0] == 'l';
    }();
    if(!legacyP2 && phase != 2)
    {
        SgmP2Batch2 Q2;
        Q2.step = (float)sp->stepXY;
        Q2.P2w = (float)sp->p2Weighting;
        Q2.fixed8 = 0;
        int maxX = 0, maxY = 0;
        for(int i = 0; i < n; ++i)
        {
            const avdm_sgm_tile_t& t = tiles[idx[i]];
            const int dimX = (int)(t.roi.x.end - t.roi.x.begin), dimY = (int)(t.roi.y.end - t.roi.y.begin);
            SgmP2Tile2& R = Q2.t[i];
            R.L = make_tex_lod(t.rc_pyr, sp->scale);
            R.rcW = (float)tex_dim_w(t.rc_pyr, sp->scale);
            R.rcH = (float)tex_dim_h(t.rc_pyr, sp->scale);
            R.dimX = dimX, R.dimY = dimY;
            for(int ai = 0; ai < 2; ++ai)
            {
                const bool scanX = ai < nAxes ? axisIsX[ai] : true;
                // deviceSimilarityVolumeKernels.cuh:688-689: beginX = (axisT.x == 0) ? roi.x.begin : roi.y.begin, applied to v.x (sic)
                const bool swap = sp->strictRoiQuirk && scanX;
                R.beginX[ai] = swap ? (int)t.roi.y.begin : (int)t.roi.x.begin;
                R.beginY[ai] = swap ? (int)t.roi.x.begin : (int)t.roi.y.begin;
                R.scanIsX[ai] = scanX ? 1 : 0;
                R.p2[ai] = ai < nAxes ? (float*)((char*)scratch + p2off[idx[i]] + (size_t)ai * p2_map_bytes(dimX, dimY)) : nullptr;
            }
            Q2.fixed8 = t.rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
            maxX = dimX > maxX ? dimX : maxX;
            maxY = dimY > maxY ? dimY : maxY;
        }
        for(int i = n; i < AVDM_SGM_MAX_TILES; ++i)
            Q2.t[i] = Q2.t[n - 1];
        hipLaunchKernelGGL(sgm_p2_map2_kernel, dim3(divUp(maxX, 16), divUp(maxY, 16), n), dim3(256), 0, st, Q2);
    }
    // (AVDM_SGM_P2_MAP=legacy) one map per (axis, tile) — ONE launch for all of them when they fit the kernel-argument table, otherwise
    // one launch per axis
    const bool oneP2Launch = nAxes * n <= AVDM_SGM_MAX_TILES;
    int maxA = 0, maxB = 0, nq = 0;
    for(int ai = 0; legacyP2 && phase != 2 && ai < nAxes; ++ai)
    {
        const bool scanX = axisIsX[ai];
        if(!oneP2Launch)
            maxA = maxB = nq = 0;
        for(int i = 0; i < n; ++i)
        {
            const avdm_sgm_tile_t& t = tiles[idx[i]];
            const int dimX = (int)(t.roi.x.end - t.roi.x.begin), dimY = (int)(t.roi.y.end - t.roi.y.begin);
            SgmP2Tile& R = Q.t[nq++];
            R.L = make_tex_lod(t.rc_pyr, sp->scale);
            R.rcW = (float)tex_dim_w(t.rc_pyr, sp->scale);
            R.rcH = (float)tex_dim_h(t.rc_pyr, sp->scale);
            // deviceSimilarityVolumeKernels.cuh:688-689: beginX = (axisT.x == 0) ? roi.x.begin : roi.y.begin, applied to v.x (sic)
            const bool swap = sp->strictRoiQuirk && scanX;
            R.beginX = swap ? (int)t.roi.y.begin : (int)t.roi.x.begin;
            R.beginY = swap ? (int)t.roi.x.begin : (int)t.roi.y.begin;
            R.A = scanX ? dimY : dimX;
            R.B = scanX ? dimX : dimY;
            R.scanIsX = scanX ? 1 : 0;
            R.p2 = (float*)((char*)scratch + p2off[idx[i]] + (size_t)ai * p2_map_bytes(dimX, dimY));
            Q.fixed8 = t.rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
            maxA = R.A > maxA ? R.A : maxA;
            maxB = R.B > maxB ? R.B : maxB;
        }
        if(!oneP2Launch || ai == nAxes - 1)
        {
            for(int i = nq; i < AVDM_SGM_MAX_TILES; ++i)
                Q.t[i] = Q.t[nq - 1];
            hipLaunchKernelGGL(sgm_p2_map_kernel, dim3(divUp(maxB, 16), divUp(maxA, 16), nq), dim3(256), 0, st, Q);
        }
    }
    // pass 2: the paths
    int npaths = 0;
    for(int ai = 0; phase != 1 && ai < nAxes; ++ai)
    {
        const bool scanX = axisIsX[ai];
        int cols = 0;
        for(int i = 0; i < n; ++i)
        {
            const avdm_sgm_tile_t& t = tiles[idx[i]];
            const int dimX = (int)(t.roi.x.end - t.roi.x.begin), dimY = (int)(t.roi.y.end - t.roi.y.begin);
            SgmPathTile& P = S.t[i];
            P.in = t.in_vol;
            P.out = t.out_vol;
            P.p2 = (const float*)((char*)scratch + p2off[idx[i]] + (size_t)ai * p2_map_bytes(dimX, dimY));
            P.Z = t.last_depth_index;
            P.A = scanX ? dimY : dimX;
            P.B = scanX ? dimX : dimY;
            P.strideA = scanX ? t.pitch_y : (long long)t.pitch_x;
            P.strideB = scanX ? (long long)t.pitch_x : t.pitch_y;
            const int wpb = pair ? AVDM_SGM_PAIR_WPB : AVDM_SGM_WPB;
            cols += (P.A + wpb - 1) / wpb; // workgroups
            P.colEnd = cols;
            const int Zp = (t.last_depth_index + 3) & ~3;
            P.tmp = (uint8_t*)scratch + p2off[idx[i]] + 2 * p2_map_bytes(dimX, dimY);
            const long long tpy = (long long)dimX * Zp;
            P.tStrideA = scanX ? tpy : (long long)Zp;
            P.tStrideB = scanX ? (long long)Zp : tpy;
        }
        for(int i = n; i < AVDM_SGM_MAX_TILES; ++i)
        {
            S.t[i] = S.t[n - 1];
            S.t[i].colEnd = 0x7fffffff;
        }
        S.t[n - 1].colEnd = (n == AVDM_SGM_MAX_TILES) ? cols : S.t[n - 1].colEnd;
        if(pair)
        {
            S.rev = 0;
            const int K = npaths;
            npaths += 2;
#ifdef AVDM_SGM_FAST_BUILD // experiments: NW = 1 only (the other instantiations take minutes to compile)
            if(NW != 1)
                return set_error_msg(1, "AVDM_SGM_FAST_BUILD: only <= 256 planes");
            launch_pair<1>(S, cols, K, full, st);
#else
            switch(NW)
            {
                case 1: launch_pair<1>(S, cols, K, full, st); break;
                case 2: launch_pair<2>(S, cols, K, full, st); break;
                case 3: launch_pair<3>(S, cols, K, full, st); break;
                case 4: launch_pair<4>(S, cols, K, full, st); break;
                default: launch_pair<6>(S, cols, K, full && NW == 6, st); break;
            }
#endif
            continue;
        }
        for(int rev = 0; rev < 2; ++rev)
        {
            S.rev = rev;
            const int K = npaths++;
#ifdef AVDM_SGM_FAST_BUILD
            if(NW != 1)
                return set_error_msg(1, "AVDM_SGM_FAST_BUILD: only <= 256 planes");
            launch_path<1>(S, cols, K, full, int16, st);
#else
            switch(NW)
            {
                case 1: launch_path<1>(S, cols, K, full, int16, st); break;
                case 2: launch_path<2>(S, cols, K, full, int16, st); break;
                case 3: launch_path<3>(S, cols, K, full, int16, st); break;
                case 4: launch_path<4>(S, cols, K, full, int16, st); break;
                default: launch_path<6>(S, cols, K, full && NW == 6, int16, st); break; // 1025..1536 planes (5 dwords/lane runs as 6)
            }
#endif
        }
    }
    return 0;
}

} // namespace avdm

using namespace avdm;

extern "C" {

/* measurement aids (not part of avdm.h; bench.py uses them for roofline.achieved): HIP events on the launch stream around every
 * path-aggregation kernel launch; _read waits for the recorded events and returns the summed kernel time and the number of launches */
int avdm_debug_sgm_kernel_timing(int enable)
{
    std::lock_guard<std::mutex> lock(g_sgmTimer.m);
    g_sgmTimer.enabled = enable != 0;
    return 0;
}

int avdm_debug_sgm_kernel_timing_read(double* total_ms, long* n_launches, int reset)
{
    std::lock_guard<std::mutex> lock(g_sgmTimer.m);
    for(size_t i = 0; i < g_sgmTimer.pending.size(); ++i)
    {
        auto& ev = g_sgmTimer.pending[i];
        float ms = 0.f;
        if(hipEventSynchronize(ev.second) != hipSuccess || hipEventElapsedTime(&ms, ev.first, ev.second) != hipSuccess)
            return set_error_msg(1, "avdm_debug_sgm_kernel_timing_read: reading a HIP event failed");
        // the span of a call: from the start of its first path kernel to the end of its last one, on the device's own clock — the launches
        // and the gaps between them, without the two command-processor hops a bracketing pair of hipEventRecord adds around the call (8-9 us)
        if(i + 1 == g_sgmTimer.pending.size() || g_sgmTimer.pendingCall[i + 1] != g_sgmTimer.pendingCall[i])
        {
            size_t f = i;
            while(f > 0 && g_sgmTimer.pendingCall[f - 1] == g_sgmTimer.pendingCall[i])
                --f;
            float span = 0.f;
            if(hipEventElapsedTime(&span, g_sgmTimer.pending[f].first, ev.second) == hipSuccess)
            {
                g_sgmTimer.spanMs += span;
                g_sgmTimer.spans += 1;
            }
        }
        g_sgmTimer.ms += ms;
        g_sgmTimer.launches += 1;
        g_sgmTimer.msPath[g_sgmTimer.pendingPath[i]] += ms;
        g_sgmTimer.launchesPath[g_sgmTimer.pendingPath[i]] += 1;
        g_sgmTimer.pool.push_back(ev);
    }
    g_sgmTimer.pending.clear();
    g_sgmTimer.pendingPath.clear();
    g_sgmTimer.pendingCall.clear();
    if(total_ms)
        *total_ms = g_sgmTimer.ms;
    if(n_launches)
        *n_launches = g_sgmTimer.launches;
    if(reset)
    {
        g_sgmTimer.ms = 0.0;
        g_sgmTimer.launches = 0;
    }
    return 0;
}

/* the same sums per first path of a launch (the pair kernel: [0] = first filtering axis, [2] = second); call it after _read(..., reset = 0), it
 * clears the per-path sums */
int avdm_debug_sgm_kernel_timing_read_paths(double ms[4], long n[4])
{
    std::lock_guard<std::mutex> lock(g_sgmTimer.m);
    for(int k = 0; k < 4; ++k)
    {
        ms[k] = g_sgmTimer.msPath[k];
        n[k] = g_sgmTimer.launchesPath[k];
        g_sgmTimer.msPath[k] = 0.0;
        g_sgmTimer.launchesPath[k] = 0;
    }
    return 0;
}

/* per avdm_volume_optimize* call: the time from the start of its first path kernel to the end of its last one (summed), and the number of calls;
 * call it after _read, it clears the sums */
int avdm_debug_sgm_kernel_timing_read_spans(double* span_ms, long* n_calls)
{
    std::lock_guard<std::mutex> lock(g_sgmTimer.m);
    if(span_ms)
        *span_ms = g_sgmTimer.spanMs;
    if(n_calls)
        *n_calls = g_sgmTimer.spans;
    g_sgmTimer.spanMs = 0.0;
    g_sgmTimer.spans = 0;
    return 0;
}

size_t avdm_volume_optimize_scratch_bytes(int dimX, int dimY, int dimZ)
{
    if(dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return 0;
    // one fp32 adaptive-P2 map per filtering axis (the path costs of the previous slice live in registers: no uint32 slice
    // buffers like Sgm.hpp:144-148 of the reference) + the scratch volume of the pair kernel
    return 2 * p2_map_bytes(dimX, dimY) + tmp_vol_bytes(dimX, dimY, dimZ);
}

static int optimize_tiles_phase(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* sp, void* stream, int phase);

int avdm_volume_optimize_tiles(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* sp, void* stream)
{
    return optimize_tiles_phase(n_tiles, tiles, scratch, sp, stream, 0);
}

int avdm_volume_optimize_prepare(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* sp, void* stream)
{
    return optimize_tiles_phase(n_tiles, tiles, scratch, sp, stream, 1);
}

int avdm_volume_optimize_tiles_prepared(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* sp, void* stream)
{
    return optimize_tiles_phase(n_tiles, tiles, scratch, sp, stream, 2);
}

static int optimize_tiles_phase(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* sp, void* stream, int phase)
{
    if(n_tiles <= 0)
        return 0;
    {
        std::lock_guard<std::mutex> lock(g_sgmTimer.m);
        g_sgmTimer.callId += 1;
    }
    if(scratch == nullptr || ((uintptr_t)scratch & 3))
        return set_error_msg(1, "avdm_volume_optimize: scratch (sum of avdm_volume_optimize_scratch_bytes() over the tiles, 4-byte aligned) is required");
    std::vector<size_t> p2off(n_tiles);
    std::vector<int> order;
    size_t off = 0;
    for(int i = 0; i < n_tiles; ++i)
    {
        const avdm_sgm_tile_t& t = tiles[i];
        const int dimX = (int)(t.roi.x.end - t.roi.x.begin), dimY = (int)(t.roi.y.end - t.roi.y.begin), Z = t.last_depth_index;
        p2off[i] = off;
        if(dimX <= 0 || dimY <= 0 || Z <= 0)
            continue; // nothing to do for this tile (cuda_volumeOptimize on an empty ROI)
        off += 2 * p2_map_bytes(dimX, dimY) + tmp_vol_bytes(dimX, dimY, Z);
        if(phase != 1)
        { // (the maps do not touch the volumes)
            if((t.pitch_x & 3) || (t.pitch_y & 3) || ((uintptr_t)t.out_vol & 3) || ((uintptr_t)t.in_vol & 3))
                return set_error_msg(1, "avdm_volume_optimize: volume base / pitches must be multiples of 4 bytes");
            if(((Z + 3) & ~3) > t.pitch_x)
                return set_error_msg(1, "avdm_volume_optimize: pitch_x must cover the 4-aligned depth count");
        }
        if(Z > 1536)
            return set_error_msg(1, "avdm_volume_optimize: more than 1536 depth planes are not supported");
        order.push_back(i);
    }
    // group by (NW, FULL): one kernel instantiation per launch
    std::vector<char> done(n_tiles, 0);
    for(size_t s = 0; s < order.size(); ++s)
    {
        if(done[order[s]])
            continue;
        const int Zs = tiles[order[s]].last_depth_index, NWs = (Zs + 255) / 256;
        const bool fulls = Zs == 256 * NWs;
        std::vector<int> grp;
        for(size_t j = s; j < order.size(); ++j)
        {
            const int Zj = tiles[order[j]].last_depth_index, NWj = (Zj + 255) / 256;
            if(!done[order[j]] && NWj == NWs && (Zj == 256 * NWj) == fulls)
            {
                grp.push_back(order[j]);
                done[order[j]] = 1;
            }
        }
        for(size_t g0 = 0; g0 < grp.size(); g0 += AVDM_SGM_MAX_TILES)
        {
            const int n = (int)std::min<size_t>(AVDM_SGM_MAX_TILES, grp.size() - g0);
            const int rc = optimize_group(tiles, grp.data() + g0, n, p2off.data(), scratch, sp, (hipStream_t)stream, phase);
            if(rc)
                return rc;
        }
    }
    AVDM_LAUNCH_CHECK("avdm_volume_optimize");
}

int avdm_volume_optimize(uint8_t* out_vol, const uint8_t* in_vol, long long pitch_y, int pitch_x, void* scratch, const avdm_pyramid_t* rc_pyr,
                         const avdm_sgm_params_t* sp, int last_depth_index, avdm_roi_t roi, void* stream)
{
    avdm_sgm_tile_t t;
    t.out_vol = out_vol;
    t.in_vol = in_vol;
    t.pitch_y = pitch_y;
    t.pitch_x = pitch_x;
    t.last_depth_index = last_depth_index;
    t.roi = roi;
    t.rc_pyr = rc_pyr;
    return avdm_volume_optimize_tiles(1, &t, scratch, sp, stream);
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
