// avdm_similarity.hip — plane-sweep weighted-NCC similarity volumes for gfx950.
//   avdm_volume_compute_similarity  <-> cuda_volumeComputeSimilarity  (planeSweeping/deviceSimilarityVolume.cu:155-206,
//                                       kernel planeSweeping/deviceSimilarityVolumeKernels.cuh:109-233)
//   avdm_volume_refine_similarity   <-> cuda_volumeRefineSimilarity   (deviceSimilarityVolume.cu:208-259, kernels.cuh:235-391)
//   NCC core                        <-> compNCCby3DptsYK              (cuda/device/Patch.cuh:466-572, SimStat.cuh, color.cuh:167-210)
//
// CDNA4 design (not the CUDA one — there is no texture path here, every bilinear tap is 4 explicit 8-byte texel reads):
//   * one lane per PIXEL, 8x8 pixels per wave64, 16x16 per workgroup; each lane walks a chunk of consecutive planes and writes
//     its results as one packed word into the z-fastest volume (4 x u8 = one dword RMW; 8 x fp16 = one 16-byte RMW).
//   * the 2 x 81 (SGM) / 2 x 49 (Refine) bilinear taps per voxel are served from LDS, not from the vector L1:
//       - the R footprint of the workgroup (16x16 stage pixels + patch halo) is staged ONCE per workgroup;
//       - for every plane the workgroup reduces the bounding box of its lanes' projected patch corners in the T image
//         (a planar patch projects to a convex quad, so the 4 corners bound all taps), stages that T window with coalesced row
//         loads, and the lanes gather from it; LDS row pitch = 8 (mod 16) texels keeps the 8x8-pixel gathers at the minimum
//         bank-conflict degree for both 1- and 2-texel pixel spacing.
//       - when a window does not fit (image border, depth discontinuity inside the workgroup, extreme view change) the
//         workgroup takes the generic path for that plane: identical arithmetic, taps from global memory with clamp addressing.
//   * patch samples are projected in homogeneous form:  P*(p + a*x + b*y) = h0 + a*(M*x) + b*(M*y)  — 3 FMA + 1 v_rcp per
//     camera and sample instead of a 3-D point + a 3x4 product; the two Yoon–Kweon exponentials are merged into one v_exp.
//   * camera parameters and the per-offset proximity table live in the kernarg segment (SGPR / scalar loads), no __constant__.
#include "avdm_device.h"

#include <limits.h>
#include <math.h>
#include <cmath>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <stdlib.h>
#include <type_traits>

// taps in flight per lane: each unrolled sample keeps 8 LDS reads (16 VGPRs) live; 3 keeps the kernels at 3 waves / SIMD
namespace avdm {

#ifndef AVDM_NCC_UNROLL
#define AVDM_NCC_UNROLL 3
#endif
constexpr int kNccUnroll = AVDM_NCC_UNROLL;
#ifndef AVDM_NCC_MULTI_UNROLL
#define AVDM_NCC_MULTI_UNROLL 3
#endif
constexpr int kNccMultiUnroll = AVDM_NCC_MULTI_UNROLL; // sample loop of the four-plane form (wsh != 3)
// The eight-plane pass (ncc_accumulate_lds_fixed8_multi<4>) — how its sample loop is scheduled, measured on the bench's SGM sweep against the
// four-plane pass at 247.1 ms (sessions r04_j, l, m, r05_a; profiles/r04_planes8_ab.txt): the T taps of pair j + 1 are requested before pair j
// is consumed, one sample per iteration of a rolled row loop (236.9 ms; 239.7 at unroll 3).  Tried and removed: the taps of two pairs
// requested together with a fence between the groups (246.0 / 244.0 ms at unroll 1 / 3; all four pairs in flight: 260.3 ms, the taps spill),
// rotating across the samples of a row too (242.4 ms), the next sample's R taps requested during the last pair (238.4 ms), the first pair's
// taps requested before the R side's arithmetic (236.2 ms: nothing).  The loop issues 34 VALU instructions per plane and sample instead of
// the four-plane pass's 43; over a launch that is -8.5 % instructions and -4.4 % time.
#ifndef AVDM_NCC_OCTO_UNROLL
#define AVDM_NCC_OCTO_UNROLL 1
#endif
constexpr int kNccOctoUnroll = AVDM_NCC_OCTO_UNROLL; // sample loop of the eight-plane form
constexpr bool kLdsSplitReads = true;
#ifndef AVDM_SGM_CHUNKS_PER_WG
#define AVDM_SGM_CHUNKS_PER_WG 4
#endif
constexpr unsigned kSgmChunksPerWg = AVDM_SGM_CHUNKS_PER_WG; // SGM similarity: chunks of 4 planes per workgroup
#ifndef AVDM_REFINE_CHUNKS_PER_WG
#define AVDM_REFINE_CHUNKS_PER_WG 4
#endif
static_assert(4u * AVDM_SGM_CHUNKS_PER_WG <= 32u && 8u * AVDM_REFINE_CHUNKS_PER_WG <= 32u, "knifeMask: one bit per plane of a workgroup in a 32-bit word");
constexpr unsigned kOutlierGrid = 2048; // workgroups of refine_outlier_kernel: its lanes stride over the units of the list
// -DAVDM_LEAN_STATS=1 (a variant build, scripts/build_variant.sh; never the shipped library): which pass the waves of the DEFAULT instantiations take —
// [0..6] SGM sweep {eight-plane passes, four-plane passes, one-plane passes from LDS, one-plane passes from global memory, workgroups without a
// chunk window, workgroups, workgroups that retried without outliers}, [8..14] Refine sweep {eight, four, one LDS, one global, workgroups with an
// anchored window, workgroups, workgroups without a chunk window}, [16..19] / [20..23] the outcome of the FIRST chunk-window search of a workgroup of
// the SGM / Refine sweep {window staged, R tile unusable or nobody valid, hull leaves the T image, hull exceeds the LDS budget}; read and cleared by
// avdm_debug_lean_stats
// 0 (variant build): the hull of a workgroup's T taps must lie inside the T image to become a window (rounds 1-6a), the A/B of the clamped windows
#ifndef AVDM_WINDOWS_OUTSIDE
#define AVDM_WINDOWS_OUTSIDE 1
#endif
#ifndef AVDM_LEAN_STATS
#define AVDM_LEAN_STATS 0
#endif
#if AVDM_LEAN_STATS
__device__ unsigned g_leanStats[32];
#define LEANSTAT(i)                                                                                                                                       \
    do                                                                                                                                                    \
    {                                                                                                                                                     \
        if((threadIdx.x & 63u) == 0u)                                                                                                                     \
            atomicAdd(&g_leanStats[i], 1u);                                                                                                               \
    } while(0)
#define LEANSTAT_WG(i)                                                                                                                                    \
    do                                                                                                                                                    \
    {                                                                                                                                                     \
        if(threadIdx.x == 0u)                                                                                                                             \
            atomicAdd(&g_leanStats[i], 1u);                                                                                                               \
    } while(0)
#else
#define LEANSTAT(i) ((void)0)
#define LEANSTAT_WG(i) ((void)0)
#endif
constexpr unsigned kRefineChunksPerWg = AVDM_REFINE_CHUNKS_PER_WG; // Refine: chunks of 8 planes per workgroup (they share one R tile, T window and pixel set-up)
#ifndef AVDM_SIM_WAVES_PER_SIMD
#define AVDM_SIM_WAVES_PER_SIMD 2 // occupancy the two kernels are compiled for: 3 -> 168 VGPRs, 2 -> 256 VGPRs
#endif
#ifndef AVDM_REFINE_OCTO_PARTIAL
#define AVDM_REFINE_OCTO_PARTIAL 1 // Refine: a chunk that only overlaps the plane range (the last 7 of the default 31 planes) through the eight-plane pass too,
#endif                             // its planes outside as invalid planes of the pass: 267.1 against 271.6 ms per depth map (session r05_a); 0 = two four-plane passes
#ifndef AVDM_OUTLIER_UNROLL
#define AVDM_OUTLIER_UNROLL 4 // refine_outlier_kernel: samples of a patch row whose global-memory taps are in flight together (session r06_c, Refine sweep over the 11 cameras: 1 -> 240.6 ms with the library sigmoid, 7 -> 237.2, 4 -> 236.6 ms with the fast one)
#endif
#ifndef AVDM_REFINE_ANCHORED_WINDOW
#define AVDM_REFINE_ANCHORED_WINDOW 1 // Refine with an outlier list: a workgroup whose lanes' hull is no window gets an anchored one (0: the round-4 tiers, for an A/B)
#endif

typedef float v2f_t __attribute__((ext_vector_type(2)));
// ---- per-deviation switches (compile time; every default is the product path) ---------------------------------------------------
// The default kernels deviate from the reference's arithmetic as written in four places (five until round 4: the R-side border test is the
// reference's own since then, see lit:: below); three can be REVERTED on the fast LDS path in a variant build (scripts/build_variant.sh <name>
// -DAVDM_DEV_...=1, selected with AVDM_LIB) so that the distance default <-> reference is attributed deviation by deviation
// (scripts/deviation_report.py).  The fourth — the R side shared by the planes of a pass — is the run-time switch
// AVDM_SIM_PLANE_PAIRS=0; the literal kernel (avdm_literal.hip) has the same switches in the other direction.
#ifndef AVDM_DEV_UNSHIFTED_SUMS
#define AVDM_DEV_UNSHIFTED_SUMS 0 // 1: the six NCC sums on the UNSHIFTED L values in the reference's order and form (SimStat.cuh:72-155), no FMA contraction
#endif
// ... per kernel (round 6, VERDICT r5 1a: "the flips are SGM's"): which of the two sweeps takes the unshifted sums when AVDM_DEV_UNSHIFTED_SUMS is set
#ifndef AVDM_DEV_UNSHIFTED_SGM
#define AVDM_DEV_UNSHIFTED_SGM 1
#endif
#ifndef AVDM_DEV_UNSHIFTED_REFINE
#define AVDM_DEV_UNSHIFTED_REFINE 1
#endif
template <bool TInvert>
constexpr bool kUnshifted = AVDM_DEV_UNSHIFTED_SUMS && (TInvert ? AVDM_DEV_UNSHIFTED_REFINE != 0 : AVDM_DEV_UNSHIFTED_SGM != 0);
#ifndef AVDM_DEV_TWO_EXP
#define AVDM_DEV_TWO_EXP 0 // 1: two Yoon-Kweon weights, two expf, multiplied (color.cuh:167-210)
#endif
#ifndef AVDM_DEV_IEEE_DIV
#define AVDM_DEV_IEEE_DIV 0 // 1: IEEE divisions where the projections use v_rcp_f32 (matrix.cuh:117-126 evaluated like the CPU pin evaluates it)
#endif
// ---- the knife-edge evaluation's A/B (compile time) ------------------------------------------------------------------------------------------
// profiles/r04_e_ab.txt, same box, 11 steps: the knife-edge evaluation costs 563.9 ms per depth map against 558.0 without it (one more live
// register in the set-up blocks of the SGM kernel).
#ifndef AVDM_KNIFE_LITERAL
#define AVDM_KNIFE_LITERAL 1 // 0: the R-side border test on the exact pixel on the knife-edge rows too (rounds 1-3; A/B of the lit:: evaluation's cost)
#endif
// The 1/256 weight quantisation floor(f * 256 + 0.5) as fma(f, 256 + 2^-14, 1.5 * 2^23) - 1.5 * 2^23: two PACKED operations for two weights
// instead of a packed FMA and two v_floor_f32 (round 4: with the R-side sums taken as fma(w, dLR, .) / fma(w, dLR^2, .), dLR^2 formed once per
// sample for all planes, 563.9 -> 551.1 ms per depth map; the round-3 forms are in the history).  The plain magic add rounds to nearest EVEN
// where floor(. + 0.5) rounds half UP, and exact halves are common — a texel coordinate near 2000 is a multiple of 2^-13, so f * 256 is a
// multiple of 1/32 and one weight in 32 is a tie (r04_f: the plain add moved the similarity volume of the far image corner from 94.4 % to
// 93.5 % identical voxels).  The 2^-14 in the multiplier lifts every tie k + 0.5 by (k + 0.5) 2^-22 inside the ONE rounding of the FMA: round
// half up again; a value is lifted across a tie wrongly only within 6e-5 below it, finer than the coordinate grid for x >= 4.
__device__ __forceinline__ v2f_t quant256(v2f_t f)
{
    const float M = 12582912.0f;
    v2f_t t = f * 256.00006103515625f + M; // 256 + 2^-14 (exact in fp32)
    asm volatile("" : "+v"(t)); // keep the two roundings apart (no re-association of (x + M) - M)
    return t - M;
}
__device__ __forceinline__ float proj_rcp(float x) { return AVDM_DEV_IEEE_DIV ? 1.0f / x : fast_rcp(x); }
// the reference's weight of one image: exp(-(dC / gammaC + dP / gammaP)) (CostYKfromLab)
__device__ __forceinline__ float yk_weight(float dC, float dP, float invGammaC) { return expf(-(dC * invGammaC + dP)); }
// sigmoid(0, 1, 0.7, -0.7, x) = 1 / (1 + exp(10 (x + 0.7) / 0.7)) of the Refine sweep (kernels.cuh:374) on the hardware's exp2 / rcp (1 ulp each:
// 2e-7 relative, the fp16 volume's quantum is 1e-3) — 4 instructions where the library expf and two IEEE divisions were ~35 per plane (round 6)
#ifndef AVDM_FAST_SIGMOID
#define AVDM_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float refine_sigmoid(float x)
{
#if AVDM_FAST_SIGMOID
    return fast_rcp(1.0f + __builtin_amdgcn_exp2f(fmaf(x, 20.609929155556620f, 14.426950408889634f))); // 10 / 0.7 * log2(e), 10 * log2(e)
#else
    return sigmoid(0.0f, 1.0f, 0.7f, -0.7f, x);
#endif
}
// SimStat::update(gx, gy, w) and computeWSim as written (no contraction: the pinned CPU build of the reference has none)
struct SimStatLit
{
    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
    __device__ __forceinline__ void update(float gx, float gy, float w)
    {
#pragma clang fp contract(off)
        wsum += w;
        xsum += w * gx;
        ysum += w * gy;
        xxsum += w * gx * gx;
        yysum += w * gy * gy;
        xysum += w * gx * gy;
    }
    __device__ __forceinline__ float raw_sim() const
    {
#pragma clang fp contract(off)
        const float varXW = (xxsum - xsum * xsum / wsum) / wsum;
        const float varYW = (yysum - ysum * ysum / wsum) / wsum;
        const float varXYW = (xysum - xsum * ysum / wsum) / wsum;
        return varXYW / sqrtf(varXW * varYW);
    }
};

// ---- knife-edge rows: the reference's R-side border test AS WRITTEN --------------------------------------------------------------------
// The reference tests the patch centre, projected back into R, against the wsh + 2 margin (Patch.cuh:486-496).  The centre lies on the ray
// of the lane's own pixel, so the re-projection is the pixel itself up to fp32 rounding, and the kernels test the exact pixel — EXCEPT where
// the pixel lies exactly on the margin (x == wsh + 2 or x == W - 1 - (wsh + 2), same in y: a handful of rows / columns per image): there the
// reference's outcome is decided by the last bit of its re-projection, per voxel, SGM then spreads it along the row, and on small images
// those rows carried most of the untrimmed distance between the default kernels and the reference's arithmetic (profiles/r04_deviation_table.json:
// 0.98 of it on cfg1).  On those lanes — and only there — the test is evaluated with the reference's own operations in its own order
// (volume_computePatch / get3DPointForPixelAndFrontoParellePlaneRC / move3DPointByRcPixSize / project3DPoint: kernels.cuh:17-35, Patch.cuh:157-170,
// matrix.cuh:66-126), IEEE division and square root, no FMA contraction: the same bits as avdm_literal.hip and as the reference's code compiled
// for the CPU (oracle/_ref), so the validity of every voxel of a knife-edge row equals the pinned reference's.
// The functions themselves are plain C++ in avdm_knife.h: tests/test_oracle.py compiles the SAME text for the host (g++ -ffp-contract=off) and
// holds it, voxel for voxel, to the oracle's literal evaluation — which tests/test_oracle_ref.py pins to the reference's code.
#pragma clang fp contract(off)
#define AVDM_KNIFE_FN __device__ __forceinline__
#include "avdm_knife.h"
#undef AVDM_KNIFE_FN
namespace lit {
__device__ __forceinline__ bool sgm_r_inside(const avdm_camera_t& rc, float x, float y, float depth, float dd, float W1, float H1)
{
    return knife::sgm_r_inside(rc.P, rc.iP, rc.C, rc.ZVect, x, y, depth, dd, W1, H1);
}
__device__ __forceinline__ bool refine_r_inside(const avdm_camera_t& rc, float x, float y, float depth, float pixSize, int rel, float dd, float W1, float H1)
{
    return knife::refine_r_inside(rc.P, rc.iP, rc.C, x, y, depth, pixSize, rel, dd, W1, H1);
}
} // namespace lit
#pragma clang fp contract(fast)

struct PatchTable
{
    float c[81]; // 2 * sqrt(xp^2 + yp^2) * invGammaP * log2(e), row-major over (yp, xp), for wsh <= 4
};

struct NccArgs
{
    TexLevel rcL, tcL;            // integral-level fast path
    float rcSx, rcOx, rcSy, rcOy; // nominal-level pixel -> texel space of the actual level: x = px*S + O
    float tcSx, tcOx, tcSy, tcOy;
    float rcW1, rcH1, tcW1, tcH1; // float(levelDim - 1) of the nominal dims (border test)
    float negInvGammaC_log2e;
    float invGammaC, invGammaP;
    float mipmapLevel;
    int wsh;
    int rcap, tcap;   // LDS capacities in texels (R tile, T window)
    int rpitch;       // row pitch of the R tile in texel positions: the same for every workgroup of a launch (the kernels specialised on it fold
                      // the second-row tap of R into the offset field of the LDS read)
    int forceGeneric; // debugging / A-B switch: never use the LDS path
    int noPacked;     // debugging / A-B switch: LDS path with the plain fp32 tap arithmetic
    int chunkWindow;  // packed path: ONE T window for the planes of a chunk (0: one window per plane, the A/B reference)
    int planePairs;   // packed path with a chunk window: two adjacent planes per pass over the patch (0: one plane per pass, the A/B reference)
    unsigned* stats;  // optional device counters per plane-workgroup: {LDS path, R tile unusable, T taps leave the image, T window too large}
};

// ---------------------------------------------------------------------------------------------
// tap sources: global memory (clamp addressing) or an LDS window (all taps guaranteed inside)
// ---------------------------------------------------------------------------------------------
struct GlobalTap
{
    TexLevel L;
    template <bool FIXED8>
    __device__ __forceinline__ float4 fetch(float x, float y) const
    {
        return tex_bilinear_px<FIXED8>(L, x, y);
    }
};

struct LdsTap
{
    const uint2* t;
    int pitch, x0, y0;
    template <bool FIXED8>
    __device__ __forceinline__ float4 fetch(float x, float y) const
    {
        const float fx = floorf(x), fy = floorf(y);
        float a = x - fx, b = y - fy;
        if(FIXED8)
        {
            a = quant8(a);
            b = quant8(b);
        }
        const int i = (int)fx - x0, j = (int)fy - y0;
        const int o00 = __mul24(j, pitch) + i, o01 = o00 + pitch;
        int o10 = o00 + 1, o11 = o01 + 1;
        if(kLdsSplitReads)
        { // keep the compiler from fusing horizontally adjacent taps into ds_read2_b64 (half the LDS rate of two ds_read_b64)
            asm volatile("" : "+v"(o10));
            asm volatile("" : "+v"(o11));
        }
        return bilinear_blend(unpack_h4(t[o00]), unpack_h4(t[o10]), unpack_h4(t[o01]), unpack_h4(t[o11]), a, b);
    }
};

// LDS row pitch (texels) for a window of w texels: smallest value = 8 (mod 16) that is >= w
__host__ __device__ __forceinline__ int lds_pitch_for(int w) { return (((w + 7) >> 4) << 4) + 8; }
// (De-interleaved T windows — column c of a 12-byte-record window at record (c >> 1) + (c & 1) * pitch / 2, which removes the two-way bank
// conflicts of the stepXY = 2 sweep — were built and measured in round 4 (AVDM_SIM_DEINT: volumes bit-identical, the sweep 2.3 % SLOWER: the
// + 4.6 % VALU instructions cost more than the conflict cycles they remove) and removed in round 5.)

// cooperative copy of the window [x0, x0+w) x [y0, y0+h) of level L into LDS (row pitch `pitch` texels); the window is inside the image
__device__ __forceinline__ void stage_window(uint2* dst, int pitch, const TexLevel& L, int x0, int y0, int w, int h)
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

// "Paired" window layout for the FIXED8 LDS path: one 16-byte record per texel position holding, per colour channel, the fp16 pair
// {texel c, texel c + 1} of the row (L, a, b and — for the centre fetch of the chunk-window path — alpha) — exactly the operand
// v_dot2_f32_f16 multiplies with the weight pair {256 - A, A}.  The pairing
// (three v_perm_b32) is then done once per staged texel instead of once per tap (12 of the ~92 VALU instructions of a sample), and a
// row tap is ONE ds_read_b128 instead of two ds_read_b64.  Costs twice the LDS per texel: used where the windows are small enough.
#define AVDM_PERM_LO 0x05040100u // {lo16(src1), lo16(src0)}
#define AVDM_PERM_HI 0x07060302u // {hi16(src1), hi16(src0)}
// The window may reach OUTSIDE the image (round 6): a position outside holds the nearest texel of the image — clamp addressing, what the texture
// unit (tex_bilinear_px: texel_clamped per tap) gives a tap there — so that the patches of a workgroup at the image border take their taps from
// LDS like everybody else.  Until then a hull that left the T image was no window: 2-3.5 % of the SGM sweep's workgroups ran one plane per pass,
// a third of those passes from global memory — 1.2 M one-plane wave passes per depth map beside 3.4 M eight-plane ones (session r06_o).
__device__ __forceinline__ void stage_window_paired(uint4* dst, int pitch, const TexLevel& L, int x0, int y0, int w, int h)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int r = wave; r < h; r += 4)
    {
        const uint2* src = L.base + (long long)min(max(y0 + r, 0), L.H - 1) * L.pitch8;
        uint4* d = dst + r * pitch;
        // 63 records per pass: every lane loads one texel, the right neighbour comes from the next lane (wave_shl:1); lane 63 only lends its texel
        for(int c0 = 0; c0 < w; c0 += 63)
        {
            const int c = c0 + lane;
            const uint2 t0 = src[min(max(x0 + min(c, w - 1), 0), L.W - 1)]; // the last column is never the left tap of a lerp: its record pairs it with itself
            uint2 t1;
            t1.x = (unsigned)__builtin_amdgcn_update_dpp((int)t0.x, (int)t0.x, 0x130, 0xf, 0xf, false);
            t1.y = (unsigned)__builtin_amdgcn_update_dpp((int)t0.y, (int)t0.y, 0x130, 0xf, 0xf, false);
            uint4 rec;
            rec.x = __builtin_amdgcn_perm(t1.x, t0.x, AVDM_PERM_LO);
            rec.y = __builtin_amdgcn_perm(t1.x, t0.x, AVDM_PERM_HI);
            rec.z = __builtin_amdgcn_perm(t1.y, t0.y, AVDM_PERM_LO);
            rec.w = __builtin_amdgcn_perm(t1.y, t0.y, AVDM_PERM_HI); // {alpha(c) | alpha(c + 1)}: only the centre fetch reads it
            if(lane < 63 && c < w)
                d[c] = rec;
        }
    }
}

// 12-byte records {L(c) | L(c + 1)}, {a(c) | a(c + 1)}, {b(c) | b(c + 1)}: the paired record without its alpha word.  The three dwords of a
// row tap ARE the three v_dot2_f32_f16 operands (no v_perm_b32 per tap, like the 16-byte records) at 3/4 of their LDS footprint — what the
// SGM kernel's windows (stepXY 2: twice the texels per pixel) fit since the kernels run two workgroups per compute unit (80 KB each).
// A record is 4-byte aligned: a tap is a ds_read2_b32 + a ds_read_b32.
struct __attribute__((packed, aligned(4))) Rec12
{
    unsigned L, a, b;
};
typedef __attribute__((address_space(3))) const Rec12* lds_rec12_ptr;
__device__ __forceinline__ uint4 lds_record12(unsigned byteAddr)
{
    const lds_rec12_ptr p = (lds_rec12_ptr)(size_t)byteAddr;
    return make_uint4(p->L, p->a, p->b, 0u);
}
__device__ __forceinline__ void stage_window_rec12(Rec12* dst, int pitch, const TexLevel& L, int x0, int y0, int w, int h)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int r = wave; r < h; r += 4)
    {
        const uint2* src = L.base + (long long)min(max(y0 + r, 0), L.H - 1) * L.pitch8; // (a window may reach outside the image: stage_window_paired)
        Rec12* d = dst + r * pitch;
        for(int c0 = 0; c0 < w; c0 += 63)
        {
            const int c = c0 + lane;
            const uint2 t0 = src[min(max(x0 + min(c, w - 1), 0), L.W - 1)]; // the last column is never the left tap of a lerp: its record pairs it with itself
            uint2 t1;
            t1.x = (unsigned)__builtin_amdgcn_update_dpp((int)t0.x, (int)t0.x, 0x130, 0xf, 0xf, false);
            t1.y = (unsigned)__builtin_amdgcn_update_dpp((int)t0.y, (int)t0.y, 0x130, 0xf, 0xf, false);
            Rec12 rec;
            rec.L = __builtin_amdgcn_perm(t1.x, t0.x, AVDM_PERM_LO);
            rec.a = __builtin_amdgcn_perm(t1.x, t0.x, AVDM_PERM_HI);
            rec.b = __builtin_amdgcn_perm(t1.y, t0.y, AVDM_PERM_LO);
            if(lane < 63 && c < w)
                d[c] = rec;
        }
    }
}

// "Half-paired" 8-byte records for the packed FIXED8 path when the 16-byte records do not fit: {L(c) | L(c + 1)}, {a(c) | b(c)}.  The L
// pair of a row tap is ready as stored; the a and b pairs still take one v_perm_b32 each from the second dwords of columns c and
// c + 1 (8 instead of 12 permutes per sample), and the right-hand column is a ds_read_b32.
#define AVDM_PERM_AB 0x05040302u // {hi16(src1), lo16(src0)}: (a | b) from {L | a}, {b | pad}
__device__ __forceinline__ void stage_window_halfpaired(uint2* dst, int pitch, const TexLevel& L, int x0, int y0, int w, int h)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for(int r = wave; r < h; r += 4)
    {
        const uint2* src = L.base + (long long)(y0 + r) * L.pitch8 + x0;
        uint2* d = dst + r * pitch;
        for(int c0 = 0; c0 < w; c0 += 63)
        {
            const int c = c0 + lane;
            const uint2 t0 = src[min(c, w - 1)];
            const unsigned t1x = (unsigned)__builtin_amdgcn_update_dpp((int)t0.x, (int)t0.x, 0x130, 0xf, 0xf, false); // next lane's {L | a}
            uint2 rec;
            rec.x = __builtin_amdgcn_perm(t1x, t0.x, AVDM_PERM_LO);
            rec.y = __builtin_amdgcn_perm(t0.y, t0.x, AVDM_PERM_AB);
            if(lane < 63 && c < w)
                d[c] = rec;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-voxel patch: homogeneous image coordinates of the centre and of the two scaled patch axes in R and T
// ---------------------------------------------------------------------------------------------
struct PatchProj
{
    f3 hr0, ht0;           // P * p
    f3 rax, ray, tax, tay; // M * (patch.x * d), M * (patch.y * d)
};

__device__ __forceinline__ void patch_axes(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 p, f3& ax, f3& ay)
{
    // computeRotCSEpip (Patch.cuh:111-135); only x and y are used downstream
    const f3 v1 = normalize(ld3(rc.C) - p);
    const f3 v2 = normalize(ld3(tc.C) - p);
    ay = normalize(cross(v1, v2));
    const f3 n = normalize((v1 + v2) * 0.5f);
    ax = normalize(cross(ay, n));
}

__device__ __forceinline__ PatchProj make_patch_proj(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 pp, f3 px, f3 py, float pd)
{
    PatchProj Q;
    Q.hr0 = M3x4mulV3(rc.P, pp);
    Q.ht0 = M3x4mulV3(tc.P, pp);
    const f3 ax = px * pd, ay = py * pd; // patch.x * patch.d, patch.y * patch.d
    Q.rax = M3x3mulV3(rc.P, ax);
    Q.ray = M3x3mulV3(rc.P, ay);
    Q.tax = M3x3mulV3(tc.P, ax);
    Q.tay = M3x3mulV3(tc.P, ay);
    return Q;
}

// c ? a : b, field by field (a conditional struct assignment is a memcpy under a branch, which kept the patches in scratch memory)
__device__ __forceinline__ f3 sel3(bool c, f3 a, f3 b) { return f3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
__device__ __forceinline__ float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
__device__ __forceinline__ PatchProj selQ(bool c, const PatchProj& a, const PatchProj& b)
{
    PatchProj Q;
    Q.hr0 = sel3(c, a.hr0, b.hr0);
    Q.ht0 = sel3(c, a.ht0, b.ht0);
    Q.rax = sel3(c, a.rax, b.rax);
    Q.ray = sel3(c, a.ray, b.ray);
    Q.tax = sel3(c, a.tax, b.tax);
    Q.tay = sel3(c, a.tay, b.tay);
    return Q;
}

// the same with the centre's homogeneous coordinates given: a point on a fixed ray, p = C + v * t, projects to  P * (C, 1) + t * (M * v)  —
// two vectors per (lane, camera) instead of a 3 x 4 product per plane
__device__ __forceinline__ PatchProj make_patch_proj_on_ray(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 hr0, f3 ht0, f3 px, f3 py, float pd)
{
    PatchProj Q;
    Q.hr0 = hr0;
    Q.ht0 = ht0;
    const f3 ax = px * pd, ay = py * pd;
    Q.rax = M3x3mulV3(rc.P, ax);
    Q.ray = M3x3mulV3(rc.P, ay);
    Q.tax = M3x3mulV3(tc.P, ax);
    Q.tay = M3x3mulV3(tc.P, ay);
    return Q;
}
// Per-lane constants of a pixel ray C + v * t for the plane loop: the pixel size is proportional to t (computePixSize is the distance of p to
// the ray of the next pixel: t * |v' x v|), and the homogeneous coordinates in both cameras are affine in t.
struct RayConsts
{
    float pixK;  // pixel size per unit distance
    float hrW;   // M_R * v = hrW * (x, y, 1): my own pixel's ray projects onto my own pixel
    f3 htB;      // M_T * v
    f3 hrA, htA; // P_R * (C, 1), P_T * (C, 1): wave-uniform (kept in SGPRs; the first is zero up to the rounding of P's last column)
};
__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ RayConsts make_ray_consts(const avdm_camera_t& rc, const avdm_camera_t& tc, f3 C, f3 v, float x, float y)
{
    RayConsts K;
    const f3 vNext = normalize(M3x3mulV2(rc.iP, x + 1.0f, y));
    K.pixK = size(cross(vNext, v));
    K.hrW = M3x3mulV3(rc.P, v).z;
    K.htB = M3x3mulV3(tc.P, v);
    const f3 a = M3x4mulV3(rc.P, C), b = M3x4mulV3(tc.P, C);
    K.hrA = f3{uniform_f32(a.x), uniform_f32(a.y), uniform_f32(a.z)};
    K.htA = f3{uniform_f32(b.x), uniform_f32(b.y), uniform_f32(b.z)};
    return K;
}
__device__ __forceinline__ f3 fma3(float t, f3 b, f3 a) { return f3{fmaf(t, b.x, a.x), fmaf(t, b.y, a.y), fmaf(t, b.z, a.z)}; }

// texel-space tap positions of patch sample (fx, fy) in R and T — the ONE expression used by the sample loop and by the
// window bounding boxes (so the boxes bound exactly what the loop will fetch)
__device__ __forceinline__ void sample_pos(const PatchProj& Q, const NccArgs& A, f3 hrRow, f3 htRow, float fx, float& rX, float& rY, float& tX,
                                           float& tY)
{
    const float hrz = fmaf(fx, Q.rax.z, hrRow.z), htz = fmaf(fx, Q.tax.z, htRow.z);
    const float ir = proj_rcp(hrz), it = proj_rcp(htz);
    const float rx = fmaf(fx, Q.rax.x, hrRow.x) * ir, ry = fmaf(fx, Q.rax.y, hrRow.y) * ir;
    const float tx = fmaf(fx, Q.tax.x, htRow.x) * it, ty = fmaf(fx, Q.tax.y, htRow.y) * it;
    rX = fmaf(rx, A.rcSx, A.rcOx);
    rY = fmaf(ry, A.rcSy, A.rcOy);
    tX = fmaf(tx, A.tcSx, A.tcOx);
    tY = fmaf(ty, A.tcSy, A.tcOy);
}
__device__ __forceinline__ void row_of(const PatchProj& Q, float fy, f3& hrRow, f3& htRow)
{
    hrRow = f3{fmaf(fy, Q.ray.x, Q.hr0.x), fmaf(fy, Q.ray.y, Q.hr0.y), fmaf(fy, Q.ray.z, Q.hr0.z)};
    htRow = f3{fmaf(fy, Q.tay.x, Q.ht0.x), fmaf(fy, Q.tay.y, Q.ht0.y), fmaf(fy, Q.tay.z, Q.ht0.z)};
}

// weighted NCC over the (2*wsh+1)^2 patch; TInvert: sigmoid-filtered positive similarity (Refine) else raw NCC in [-1, 1]
// UNR: samples of a row in flight (1 inside the sweep kernels, see below; the outlier kernel — every tap a global-memory round trip, one lane per
// plane, registers to spare — takes a whole 7-tap row)
template <bool FIXED8, int WSH, bool TInvert, class RTap, class TTap, int UNR = 1>
__device__ __forceinline__ float ncc_accumulate(const PatchProj& Q, const NccArgs& A, const PatchTable& tab, const RTap& rt, const TTap& tt,
                                                float4 rcCenter, float4 tcCenter)
{
    const int wsh = WSH > 0 ? WSH : A.wsh;
    float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
    [[maybe_unused]] SimStatLit lit;
    const int n = 2 * wsh + 1;

#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
    {
        f3 hrRow, htRow;
        row_of(Q, (float)yp, hrRow, htRow);
        const float* trow = tab.c + (yp + wsh) * n + wsh;
// (the generic paths run on ~5 % of the plane-workgroups: no unrolling here — unrolled, their taps raise the register demand of the whole
        // kernel and the compiler spills values that are live across the packed path's loop)
#pragma unroll UNR
        for(int xp = -wsh; xp <= wsh; ++xp)
        {
            float rX, rY, tX, tY;
            sample_pos(Q, A, hrRow, htRow, (float)xp, rX, rY, tX, tY);
            const float4 rcC = rt.template fetch<FIXED8>(rX, rY);
            const float4 tcC = tt.template fetch<FIXED8>(tX, tY);

            // w = exp(-(dC_r/gC + dP/gP)) * exp(-(dC_t/gC + dP/gP)) = exp2((dC_r + dC_t) * (-log2e/gC) - 2*dP*log2e/gP)
            const float drx = rcCenter.x - rcC.x, dry = rcCenter.y - rcC.y, drz = rcCenter.z - rcC.z;
            const float dtx = tcCenter.x - tcC.x, dty = tcCenter.y - tcC.y, dtz = tcCenter.z - tcC.z;
            const float dcr = __builtin_amdgcn_sqrtf(fmaf(drx, drx, fmaf(dry, dry, drz * drz)));
            const float dct = __builtin_amdgcn_sqrtf(fmaf(dtx, dtx, fmaf(dty, dty, dtz * dtz)));
#if AVDM_DEV_TWO_EXP
            const float dPl = sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP;
            const float w = yk_weight(dcr, dPl, A.invGammaC) * yk_weight(dct, dPl, A.invGammaC);
#else
            const float w = __builtin_amdgcn_exp2f(fmaf(dcr + dct, A.negInvGammaC_log2e, -trow[xp]));
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
            lit.update(rcC.x, tcC.x, w);
    }
#endif

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
    float rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
    rawSim = lit.raw_sim();
    }
#endif
    const float sim = isfinite(rawSim) ? -rawSim : 1.0f;
    if(TInvert)
        return refine_sigmoid(sim);
    return sim;
}

// ---------------------------------------------------------------------------------------------
// FIXED8 fast path from LDS: the same weighted NCC with the R and T sides carried as the two halves of packed fp32 registers
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per VALU slot — the plain fp32 VALU issues one wave instruction per
// 4 cycles per SIMD, so packing is what cuts the instruction count), and the fp16 texels consumed without unpacking:
//   * FIXED8 weights are k/256, k = 0..256; they are kept as integers (k, 256 - k), exactly representable in fp16, so every
//     interpolated colour carries a factor 2^16, which is folded into the centre colours and into the gammaC factor (powers of
//     two commute with fp32 rounding; the NCC itself is scale-free);
//   * the horizontal half of each tap is one v_dot2_f32_f16 per channel on fp16 texel pairs that were paired when the window was staged;
//   * window-relative texel indices are formed in fp32 (exact integers) with one packed FMA for both images.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f floor2(v2f v) { return v2f{floorf(v.x), floorf(v.y)}; }

typedef __attribute__((address_space(3))) const unsigned long long* lds_texel_ptr; // one ds_read_b64
struct LdsWindows
{
    unsigned rPitchB, tPitchB; // row pitches in bytes
    float rPitchBF, tPitchBF;
    float rOffB, tOffB;        // LDS byte address of texel (0, 0) of each window: base + 8 * (-(y0 * pitch + x0) [+ rcap]); |.| < 2^24
};
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4u32* lds_record_ptr; // one ds_read_b128
__device__ __forceinline__ uint4 lds_record(unsigned byteAddr)
{
    const v4u32 v = *(lds_record_ptr)(size_t)byteAddr;
    return make_uint4(v.x, v.y, v.z, v.w);
}
typedef __attribute__((address_space(3))) const unsigned* lds_u32_ptr; // one ds_read_b32
__device__ __forceinline__ unsigned lds_u32(unsigned byteAddr) { return *(lds_u32_ptr)(size_t)byteAddr; }
__device__ __forceinline__ uint2 lds_texel(unsigned byteAddr)
{
    const unsigned long long v = *(lds_texel_ptr)(size_t)byteAddr;
    uint2 r;
    r.x = (unsigned)v;
    r.y = (unsigned)(v >> 32);
    return r;
}

// Horizontal half of a bilinear tap: v_dot2_f32_f16 multiplies the fp16 pair {texel c, texel c + 1} of a channel with the fp16 weight
// pair {256 - A, A} (exact: k <= 256) and adds in fp32 — products of two fp16 values are exact in fp32, so the row lerp has a single
// rounding.  The pairs come paired from LDS (stage_window_paired / stage_window_halfpaired).
// The four horizontal lerps of a sample (R / T image x top / bottom texel row, three channels each) are ONE asm block of twelve VOP3P
// v_dot2_f32_f16 with the inline constant 0 as accumulator.  The builtin only ever selects the VOP2 form v_dot2c_f32_f16, which
// accumulates into its destination and therefore costs a v_mov_b32 0 per product.  The compiler does not see the DOT inside an asm
// statement and would not pad its result hazard (3 wait states before another VALU reads a DOT result on gfx94x/95x — what made an
// earlier asm attempt return wrong similarities): the block ends with s_nop 2, and the twelve products are independent of each other.
struct Lab3
{
    float L, a, b;
};
struct Lab3x4
{
    Lab3 rt, tt, rb, tb;
};
__device__ __forceinline__ v2h pk_half_weights(float w0, float w1) { return __builtin_bit_cast(v2h, __builtin_amdgcn_cvt_pkrtz(w0, w1)); }

// the twelve products of a sample from paired records (see stage_window_paired): no v_perm_b32 per tap
__device__ __forceinline__ Lab3x4 hlerp3x4_paired(uint4 r0, uint4 r1, uint4 t0, uint4 t1, v2h wr, v2h wt)
{
    Lab3x4 o;
    const unsigned uwr = __builtin_bit_cast(unsigned, wr), uwt = __builtin_bit_cast(unsigned, wt);
    asm("v_dot2_f32_f16 %0, %12, %24, 0\n\t"
        "v_dot2_f32_f16 %1, %13, %24, 0\n\t"
        "v_dot2_f32_f16 %2, %14, %24, 0\n\t"
        "v_dot2_f32_f16 %3, %15, %25, 0\n\t"
        "v_dot2_f32_f16 %4, %16, %25, 0\n\t"
        "v_dot2_f32_f16 %5, %17, %25, 0\n\t"
        "v_dot2_f32_f16 %6, %18, %24, 0\n\t"
        "v_dot2_f32_f16 %7, %19, %24, 0\n\t"
        "v_dot2_f32_f16 %8, %20, %24, 0\n\t"
        "v_dot2_f32_f16 %9, %21, %25, 0\n\t"
        "v_dot2_f32_f16 %10, %22, %25, 0\n\t"
        "v_dot2_f32_f16 %11, %23, %25, 0\n\t"
        "s_nop 2"
        : "=&v"(o.rt.L), "=&v"(o.rt.a), "=&v"(o.rt.b), "=&v"(o.tt.L), "=&v"(o.tt.a), "=&v"(o.tt.b), "=&v"(o.rb.L), "=&v"(o.rb.a), "=&v"(o.rb.b),
          "=&v"(o.tb.L), "=&v"(o.tb.a), "=&v"(o.tb.b)
        : "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(t0.x), "v"(t0.y), "v"(t0.z), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(t1.x), "v"(t1.y), "v"(t1.z), "v"(uwr),
          "v"(uwt));
    return o;
}

// half-paired records: r0 / r1 = top / bottom record of column c, r0n / r1n = second dword ({a | b}) of column c + 1
__device__ __forceinline__ Lab3x4 hlerp3x4_halfpaired(uint2 r0, unsigned r0n, uint2 r1, unsigned r1n, uint2 t0, unsigned t0n, uint2 t1, unsigned t1n, v2h wr,
                                                      v2h wt)
{
    Lab3x4 o;
    const unsigned uwr = __builtin_bit_cast(unsigned, wr), uwt = __builtin_bit_cast(unsigned, wt);
    const unsigned p1 = __builtin_amdgcn_perm(r0n, r0.y, AVDM_PERM_LO), p2 = __builtin_amdgcn_perm(r0n, r0.y, AVDM_PERM_HI);
    const unsigned p4 = __builtin_amdgcn_perm(t0n, t0.y, AVDM_PERM_LO), p5 = __builtin_amdgcn_perm(t0n, t0.y, AVDM_PERM_HI);
    const unsigned p7 = __builtin_amdgcn_perm(r1n, r1.y, AVDM_PERM_LO), p8 = __builtin_amdgcn_perm(r1n, r1.y, AVDM_PERM_HI);
    const unsigned p10 = __builtin_amdgcn_perm(t1n, t1.y, AVDM_PERM_LO), p11 = __builtin_amdgcn_perm(t1n, t1.y, AVDM_PERM_HI);
    asm("v_dot2_f32_f16 %0, %12, %24, 0\n\t"
        "v_dot2_f32_f16 %1, %13, %24, 0\n\t"
        "v_dot2_f32_f16 %2, %14, %24, 0\n\t"
        "v_dot2_f32_f16 %3, %15, %25, 0\n\t"
        "v_dot2_f32_f16 %4, %16, %25, 0\n\t"
        "v_dot2_f32_f16 %5, %17, %25, 0\n\t"
        "v_dot2_f32_f16 %6, %18, %24, 0\n\t"
        "v_dot2_f32_f16 %7, %19, %24, 0\n\t"
        "v_dot2_f32_f16 %8, %20, %24, 0\n\t"
        "v_dot2_f32_f16 %9, %21, %25, 0\n\t"
        "v_dot2_f32_f16 %10, %22, %25, 0\n\t"
        "v_dot2_f32_f16 %11, %23, %25, 0\n\t"
        "s_nop 2"
        : "=&v"(o.rt.L), "=&v"(o.rt.a), "=&v"(o.rt.b), "=&v"(o.tt.L), "=&v"(o.tt.a), "=&v"(o.tt.b), "=&v"(o.rb.L), "=&v"(o.rb.a), "=&v"(o.rb.b),
          "=&v"(o.tb.L), "=&v"(o.tb.a), "=&v"(o.tb.b)
        : "v"(r0.x), "v"(p1), "v"(p2), "v"(t0.x), "v"(p4), "v"(p5), "v"(r1.x), "v"(p7), "v"(p8), "v"(t1.x), "v"(p10), "v"(p11), "v"(uwr), "v"(uwt));
    return o;
}

// R side of a multi-plane pass (ncc_accumulate_lds_fixed8_quad / _multi): the six products of one image, top and bottom texel row
struct Lab3x2
{
    Lab3 t, b;
};
__device__ __forceinline__ Lab3x2 hlerp3x2_paired(uint4 r0, uint4 r1, v2h wr)
{
    Lab3x2 o;
    const unsigned uwr = __builtin_bit_cast(unsigned, wr);
    asm("v_dot2_f32_f16 %0, %6, %12, 0\n\t"
        "v_dot2_f32_f16 %1, %7, %12, 0\n\t"
        "v_dot2_f32_f16 %2, %8, %12, 0\n\t"
        "v_dot2_f32_f16 %3, %9, %12, 0\n\t"
        "v_dot2_f32_f16 %4, %10, %12, 0\n\t"
        "v_dot2_f32_f16 %5, %11, %12, 0\n\t"
        "s_nop 2"
        : "=&v"(o.t.L), "=&v"(o.t.a), "=&v"(o.t.b), "=&v"(o.b.L), "=&v"(o.b.a), "=&v"(o.b.b)
        : "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(uwr));
    return o;
}
__device__ __forceinline__ Lab3x2 hlerp3x2_halfpaired(uint2 r0, unsigned r0n, uint2 r1, unsigned r1n, v2h wr)
{
    Lab3x2 o;
    const unsigned uwr = __builtin_bit_cast(unsigned, wr);
    const unsigned p1 = __builtin_amdgcn_perm(r0n, r0.y, AVDM_PERM_LO), p2 = __builtin_amdgcn_perm(r0n, r0.y, AVDM_PERM_HI);
    const unsigned p4 = __builtin_amdgcn_perm(r1n, r1.y, AVDM_PERM_LO), p5 = __builtin_amdgcn_perm(r1n, r1.y, AVDM_PERM_HI);
    asm("v_dot2_f32_f16 %0, %6, %12, 0\n\t"
        "v_dot2_f32_f16 %1, %7, %12, 0\n\t"
        "v_dot2_f32_f16 %2, %8, %12, 0\n\t"
        "v_dot2_f32_f16 %3, %9, %12, 0\n\t"
        "v_dot2_f32_f16 %4, %10, %12, 0\n\t"
        "v_dot2_f32_f16 %5, %11, %12, 0\n\t"
        "s_nop 2"
        : "=&v"(o.t.L), "=&v"(o.t.a), "=&v"(o.t.b), "=&v"(o.b.L), "=&v"(o.b.a), "=&v"(o.b.b)
        : "v"(r0.x), "v"(p1), "v"(p2), "v"(r1.x), "v"(p4), "v"(p5), "v"(uwr));
    return o;
}

template <int WSH, bool TInvert, bool PAIRED, int RP = 0, bool REC12 = false>
__device__ __forceinline__ float ncc_accumulate_lds_fixed8(const PatchProj& Q, const NccArgs& A, const PatchTable& tab, const LdsWindows& Wn,
                                                           float4 rcCenter, float4 tcCenter)
{
    const int wsh = WSH > 0 ? WSH : A.wsh;
    const int n = 2 * wsh + 1;
    const v2f ax = {Q.rax.x, Q.tax.x}, ay = {Q.rax.y, Q.tax.y}, az = {Q.rax.z, Q.tax.z};
    const v2f bx = {Q.ray.x, Q.tay.x}, by = {Q.ray.y, Q.tay.y}, bz = {Q.ray.z, Q.tay.z};
    const v2f h0x = {Q.hr0.x, Q.ht0.x}, h0y = {Q.hr0.y, Q.ht0.y}, h0z = {Q.hr0.z, Q.ht0.z};
    const v2f Sx = {A.rcSx, A.tcSx}, Ox = {A.rcOx, A.tcOx}, Sy = {A.rcSy, A.tcSy}, Oy = {A.rcOy, A.tcOy};
    const v2f axS = ax * Sx, ayS = ay * Sy;
    const v2f pitch2 = {Wn.rPitchBF, Wn.tPitchBF}, off2 = {Wn.rOffB, Wn.tOffB};
    const float S16 = 65536.0f;
    const v2f cL = v2f{rcCenter.x, tcCenter.x} * S16, ca = v2f{rcCenter.y, tcCenter.y} * S16, cb = v2f{rcCenter.z, tcCenter.z} * S16;
    const float kC = A.negInvGammaC_log2e * (1.0f / 65536.0f);

    v2f sum1 = {0.f, 0.f}; // {xsum, ysum}
    v2f sum2 = {0.f, 0.f}; // {xxsum, yysum}
    float xysum = 0.f, wsum = 0.f;
    [[maybe_unused]] SimStatLit lit;

#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
    {
        const float fy = (float)yp;
        // == row_of(), with the texel-space scale folded into the numerators: X = ((fx*ax + rowx) * Sx) / hz + Ox
        const v2f rowx = (fy * bx + h0x) * Sx, rowy = (fy * by + h0y) * Sy, rowz = fy * bz + h0z;
        const float* trow = tab.c + (yp + wsh) * n + wsh;
        auto sample = [&](int xp) __attribute__((always_inline)) {
                const float fx = (float)xp;
                // == sample_pos(): homogeneous coordinates, one v_rcp per image, texel-space transform
                // (replacing the two v_rcp_f32 by a packed third-order series in the patch's depth extent was measured, r02_e: no change —
                // the transcendental pipe runs beside the packed FMAs, it is not what the loop waits for)
                const v2f hz = fx * az + rowz;
                const v2f inv = {proj_rcp(hz.x), proj_rcp(hz.y)};
                const v2f X = (fx * axS + rowx) * inv + Ox;
                const v2f Y = (fx * ayS + rowy) * inv + Oy;
                const v2f fX = floor2(X), fY = floor2(Y);
                // quant8(): weights in units of 1/256
                const v2f wa = floor2((X - fX) * 256.0f + 0.5f), wb = floor2((Y - fY) * 256.0f + 0.5f);
                const v2f na = 256.0f - wa, nnb = wb - 256.0f; // nnb = -(256 - B)
                // LDS byte address of the top-left tap, formed in fp32 (exact integers), then one conversion per image
                v2f oidx = fY * pitch2 + (fX * (PAIRED ? 16.0f : (REC12 ? 12.0f : 8.0f)) + off2);
                const unsigned oR = (unsigned)(int)oidx.x, oT = (unsigned)(int)oidx.y;
                const v2h wr = pk_half_weights(na.x, wa.x), wt = pk_half_weights(na.y, wa.y);
                Lab3x4 h;
                if(PAIRED)
                {
                    // RP > 0: the R pitch is a compile-time constant and the bottom tap is the top tap's address + an immediate offset
                    __builtin_assume(oR < 65536u);
                    const unsigned oRb = RP > 0 ? oR + (unsigned)(RP * 16) : oR + Wn.rPitchB;
                    const uint4 r0 = lds_record(oR), r1 = lds_record(oRb), t0 = lds_record(oT), t1 = lds_record(oT + Wn.tPitchB);
                    h = hlerp3x4_paired(r0, r1, t0, t1, wr, wt);
                }
                else if(REC12)
                {
                    // 12-byte records: the three dwords of a tap are the dot2 operands as stored
                    __builtin_assume(oR < 65536u);
                    const unsigned oRb = RP > 0 ? oR + (unsigned)(RP * 12) : oR + Wn.rPitchB;
                    const uint4 r0 = lds_record12(oR), r1 = lds_record12(oRb), t0 = lds_record12(oT), t1 = lds_record12(oT + Wn.tPitchB);
                    h = hlerp3x4_paired(r0, r1, t0, t1, wr, wt);
                }
                else
                {
                    // half-paired records: column c whole (ds_read_b64), of column c + 1 only the {a | b} dword (ds_read_b32 at +12)
                    __builtin_assume(oR < 65536u);
                    const unsigned oRb = RP > 0 ? oR + (unsigned)(RP * 8) : oR + Wn.rPitchB, oTb = oT + Wn.tPitchB;
                    const uint2 r0 = lds_texel(oR), r1 = lds_texel(oRb), t0 = lds_texel(oT), t1 = lds_texel(oTb);
                    const unsigned r0n = lds_u32(oR + 12u), r1n = lds_u32(oRb + 12u), t0n = lds_u32(oT + 12u), t1n = lds_u32(oTb + 12u);
                    h = hlerp3x4_halfpaired(r0, r0n, r1, r1n, t0, t0n, t1, t1n, wr, wt);
                }
                const Lab3 &rt = h.rt, &tt = h.tt, &rb = h.rb, &tb = h.tb;
                // centre - bilinear value (x 2^16), the vertical lerp folded into the difference: two packed FMAs per channel
                const v2f dL = (v2f{rt.L, tt.L} * nnb + cL) - v2f{rb.L, tb.L} * wb;
                const v2f da = (v2f{rt.a, tt.a} * nnb + ca) - v2f{rb.a, tb.a} * wb;
                const v2f db = (v2f{rt.b, tt.b} * nnb + cb) - v2f{rb.b, tb.b} * wb;
                const v2f sq = dL * dL + (da * da + db * db);
                const float dcs = __builtin_amdgcn_sqrtf(sq.x) + __builtin_amdgcn_sqrtf(sq.y);
#if AVDM_DEV_TWO_EXP
                const float dPl = sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP;
                const float w = yk_weight(sqrtf(sq.x) * (1.0f / 65536.0f), dPl, A.invGammaC) * yk_weight(sqrtf(sq.y) * (1.0f / 65536.0f), dPl, A.invGammaC);
#else
                const float w = __builtin_amdgcn_exp2f(fmaf(dcs, kC, -trow[xp]));
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
                {
                    // the bilinear L values themselves (x 2^16: exact horizontal products, one rounding per vertical step), then the reference's sums
                    const v2f V = v2f{rb.L, tb.L} * wb - v2f{rt.L, tt.L} * nnb;
                    lit.update(V.x * (1.0f / 65536.0f), V.y * (1.0f / 65536.0f), w);
                }
    }
#endif

                const v2f wg = dL * w;
                wsum += w;
                sum1 += wg;
                sum2 = wg * dL + sum2;
                xysum = fmaf(wg.x, dL.y, xysum);
        };
        if(WSH == 3)
        {
            // 7 taps per row do not divide by the 3 samples in flight the register budget allows (8 LDS reads = 16 VGPRs each): any
            // `#pragma unroll k` with a remainder makes the compiler unroll the whole row and spill 20-40 VGPRs to scratch inside the
            // hottest loop of the program.  3 + 3 + 1 with scheduling fences keeps three samples in flight and nothing in scratch.
#pragma unroll
            for(int xp = -3; xp < 0; ++xp)
                sample(xp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int xp = 0; xp < 3; ++xp)
                sample(xp);
            __builtin_amdgcn_sched_barrier(0);
            sample(3);
            __builtin_amdgcn_sched_barrier(0);
        }
        else
        {
#pragma unroll kNccUnroll
            for(int xp = -wsh; xp <= wsh; ++xp)
                sample(xp);
        }
    }

    const float iw = fast_rcp(wsum);
    const float varXW = (sum2.x - sum1.x * sum1.x * iw) * iw;
    const float varYW = (sum2.y - sum1.y * sum1.y * iw) * iw;
    const float varXYW = (xysum - sum1.x * sum1.y * iw) * iw;
    float rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
    rawSim = lit.raw_sim();
    }
#endif
    const float sim = isfinite(rawSim) ? -rawSim : 1.0f;
    if(TInvert)
        return refine_sigmoid(sim);
    return sim;
}

// FOUR adjacent planes of a pixel (one chunk of the SGM kernel) in one pass over the patch (the plane-pair pass of round 2 taken one step
// further; that pass itself was removed in round 5).  The R side of a sample — position, weights, taps, colour distance — is evaluated ONCE, from the patch of a reference plane, for
// two packed T pairs {plane 0, plane 1}, {plane 2, plane 3}: per plane-sample (37 + 2 x 81) / 4 = 50 VALU instructions instead of 59.
// The R taps of a plane up to two depth steps from the reference plane move by <= 2e-4 ... 2e-3 texel (the tilt of the patch's x axis; zero
// at the principal point) — the class of the fp32 rounding of the pixel coordinates themselves.
// The planes' geometry comes in the form the pixel RAY gives it (RayConsts): the patch centre of plane k is C + v t_k, so its homogeneous T
// coordinates are htA + t_k htB; the patch's y axis is the normal of the epipolar plane of the ray — the same for every plane — and the pixel
// size is pixK t_k, so M_T (y d) = t_k Bt with ONE vector Bt per lane; only M_T (x d) is per plane.  Per pair of planes the loop keeps
// {t}, {M x} and the centre colours (7 packed registers) instead of 12, which is what lets two pairs + three samples of taps fit 256 VGPRs.
struct QuadPlane
{
    float t;  // distance along the pixel ray
    f3 tax;   // M_T * (patch.x * pixSize)
    float4 c; // T centre colour
};
template <int WSH, bool TInvert, bool PAIRED, int RP, bool REC12 = false>
__device__ __forceinline__ void ncc_accumulate_lds_fixed8_quad(f3 rax, f3 ray, f3 hr0, const QuadPlane& q0, const QuadPlane& q1, const QuadPlane& q2,
                                                               const QuadPlane& q3, f3 Bt, f3 htB, f3 htA, const NccArgs& A, const PatchTable& tab,
                                                               const LdsWindows& Wn, float4 rcCenter, float& sim0, float& sim1, float& sim2, float& sim3)
{
    constexpr int NPAIR = 2;
    const int wsh = WSH > 0 ? WSH : A.wsh;
    const int n = 2 * wsh + 1;
    const float S16 = 65536.0f;
    v2f tt[NPAIR], axS[NPAIR], ayS[NPAIR], az[NPAIR], cL[NPAIR], ca[NPAIR], cb[NPAIR];
    auto set_pair = [&](int j, const QuadPlane& a, const QuadPlane& b) __attribute__((always_inline)) {
        tt[j] = v2f{a.t, b.t};
        axS[j] = v2f{a.tax.x, b.tax.x} * A.tcSx;
        ayS[j] = v2f{a.tax.y, b.tax.y} * A.tcSy;
        az[j] = v2f{a.tax.z, b.tax.z};
        cL[j] = v2f{a.c.x, b.c.x} * S16;
        ca[j] = v2f{a.c.y, b.c.y} * S16;
        cb[j] = v2f{a.c.z, b.c.z} * S16;
    };
    set_pair(0, q0, q1);
    set_pair(1, q2, q3);
    // R side, {x, y} of the one image
    const v2f raS = {rax.x * A.rcSx, rax.y * A.rcSy};
    const v2f rS = {A.rcSx, A.rcSy}, rO = {A.rcOx, A.rcOy};
    const v2f rcLa = v2f{rcCenter.x, rcCenter.y} * S16;
    const float rcb = rcCenter.z * S16;
    const float kC = A.negInvGammaC_log2e * (1.0f / 65536.0f);
    constexpr float recB = PAIRED ? 16.0f : (REC12 ? 12.0f : 8.0f);
    constexpr bool WIDE = PAIRED || REC12; // a tap's record IS its three dot2 operands

    v2f wsum[NPAIR], s1R[NPAIR], s1T[NPAIR], s2R[NPAIR], s2T[NPAIR], sxy[NPAIR];
#pragma unroll
    for(int j = 0; j < NPAIR; ++j)
        wsum[j] = s1R[j] = s1T[j] = s2R[j] = s2T[j] = sxy[j] = v2f{0.f, 0.f};
#if AVDM_DEV_UNSHIFTED_SUMS
    SimStatLit lit[2 * NPAIR];
#endif

    struct RTaps
    {
        typename std::conditional<PAIRED || REC12, uint4, uint2>::type r0, r1;
        unsigned r0n, r1n;
        float rNy, rWy;
        v2h wr;
    };
    struct TTaps
    {
        typename std::conditional<PAIRED || REC12, uint4, uint2>::type a0, a1, b0, b1;
        unsigned a0n, a1n, b0n, b1n;
        v2f nnb, wb;
        v2h wtA, wtB;
    };

#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
    {
        const float fy = (float)yp;
        // row of plane k: htA + t_k * (htB + fy * Bt), the texel-space scale folded into x and y
        const f3 sB = f3{fmaf(fy, Bt.x, htB.x), fmaf(fy, Bt.y, htB.y), fmaf(fy, Bt.z, htB.z)};
        v2f rowx[NPAIR], rowy[NPAIR], rowz[NPAIR];
#pragma unroll
        for(int j = 0; j < NPAIR; ++j)
        {
            rowx[j] = (tt[j] * sB.x + htA.x) * A.tcSx;
            rowy[j] = (tt[j] * sB.y + htA.y) * A.tcSy;
            rowz[j] = tt[j] * sB.z + htA.z;
        }
        const v2f rrow = (fy * v2f{ray.x, ray.y} + v2f{hr0.x, hr0.y}) * rS;
        const float rrowz = fmaf(fy, ray.z, hr0.z);
        const float* trow = tab.c + (yp + wsh) * n + wsh;
        auto fetch_r = [&](int xp) __attribute__((always_inline)) -> RTaps {
            RTaps t;
            const float fx = (float)xp;
            const float rinv = proj_rcp(fmaf(fx, rax.z, rrowz));
            const v2f rXY = (fx * raS + rrow) * rinv + rO;
            const v2f rF = floor2(rXY);
            const v2f rW = quant256(rXY - rF);                           // {A, B} of quant8(), in units of 1/256
            const v2f rN = rW * v2f{-1.0f, 1.0f} + v2f{256.0f, -256.0f}; // {256 - A, -(256 - B)}
            const unsigned oR = (unsigned)(int)fmaf(rF.y, Wn.rPitchBF, fmaf(rF.x, recB, Wn.rOffB));
            __builtin_assume(oR < 65536u);
            t.wr = pk_half_weights(rN.x, rW.x);
            t.rNy = rN.y;
            t.rWy = rW.y;
            const unsigned oRb = RP > 0 ? oR + (unsigned)(RP * (PAIRED ? 16 : (REC12 ? 12 : 8))) : oR + Wn.rPitchB;
            if constexpr(PAIRED)
            {
                t.r0 = lds_record(oR), t.r1 = lds_record(oRb);
                t.r0n = t.r1n = 0u;
            }
            else if constexpr(REC12)
            {
                t.r0 = lds_record12(oR), t.r1 = lds_record12(oRb);
                t.r0n = t.r1n = 0u;
            }
            else
            {
                t.r0 = lds_texel(oR), t.r1 = lds_texel(oRb);
                t.r0n = lds_u32(oR + 12u), t.r1n = lds_u32(oRb + 12u);
            }
            return t;
        };
        auto fetch_t = [&](int xp, int j) __attribute__((always_inline)) -> TTaps {
            TTaps t;
            const float fx = (float)xp;
            const v2f hz = fx * az[j] + rowz[j];
            const v2f inv = {proj_rcp(hz.x), proj_rcp(hz.y)};
            const v2f X = (fx * axS[j] + rowx[j]) * inv + A.tcOx;
            const v2f Y = (fx * ayS[j] + rowy[j]) * inv + A.tcOy;
            const v2f fX = floor2(X), fY = floor2(Y);
            const v2f wa = quant256(X - fX);
            t.wb = quant256(Y - fY);
            const v2f na = 256.0f - wa;
            t.nnb = t.wb - 256.0f;
            const v2f oidx = fY * Wn.tPitchBF + (fX * recB + Wn.tOffB);
            const unsigned oA = (unsigned)(int)oidx.x, oB = (unsigned)(int)oidx.y;
            t.wtA = pk_half_weights(na.x, wa.x);
            t.wtB = pk_half_weights(na.y, wa.y);
            if constexpr(PAIRED)
            {
                t.a0 = lds_record(oA), t.a1 = lds_record(oA + Wn.tPitchB), t.b0 = lds_record(oB), t.b1 = lds_record(oB + Wn.tPitchB);
                t.a0n = t.a1n = t.b0n = t.b1n = 0u;
            }
            else if constexpr(REC12)
            {
                t.a0 = lds_record12(oA), t.a1 = lds_record12(oA + Wn.tPitchB), t.b0 = lds_record12(oB), t.b1 = lds_record12(oB + Wn.tPitchB);
                t.a0n = t.a1n = t.b0n = t.b1n = 0u;
            }
            else
            {
                t.a0 = lds_texel(oA), t.a1 = lds_texel(oA + Wn.tPitchB), t.b0 = lds_texel(oB), t.b1 = lds_texel(oB + Wn.tPitchB);
                t.a0n = lds_u32(oA + 12u), t.a1n = lds_u32(oA + Wn.tPitchB + 12u), t.b0n = lds_u32(oB + 12u), t.b1n = lds_u32(oB + Wn.tPitchB + 12u);
            }
            return t;
        };
        auto sample = [&](int xp) __attribute__((always_inline)) {
            const RTaps r = fetch_r(xp);
            TTaps t[NPAIR];
#pragma unroll
            for(int j = 0; j < NPAIR; ++j)
                t[j] = fetch_t(xp, j);
            Lab3x2 hr;
            if constexpr(WIDE)
                hr = hlerp3x2_paired(r.r0, r.r1, r.wr);
            else
                hr = hlerp3x2_halfpaired(r.r0, r.r0n, r.r1, r.r1n, r.wr);
            const v2f dRLa = (v2f{hr.t.L, hr.t.a} * r.rNy + rcLa) - v2f{hr.b.L, hr.b.a} * r.rWy;
            const float dRb = fmaf(hr.t.b, r.rNy, rcb) - hr.b.b * r.rWy;
            const v2f qR = dRLa * dRLa;
            const float base = fmaf(__builtin_amdgcn_sqrtf(fmaf(dRb, dRb, qR.x + qR.y)), kC, -trow[xp]);
            const float dLR = dRLa.x;
            const float dLR2 = dLR * dLR;
#if AVDM_DEV_TWO_EXP
            const float dPl = sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP;
            const float wRl = yk_weight(sqrtf(fmaf(dRb, dRb, qR.x + qR.y)) * (1.0f / 65536.0f), dPl, A.invGammaC);
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
            const float VR = (hr.b.L * r.rWy - hr.t.L * r.rNy) * (1.0f / 65536.0f); // the bilinear L of R itself
#endif
#pragma unroll
            for(int j = 0; j < NPAIR; ++j)
            {
                Lab3x4 h; // rt / rb = plane 2j top / bottom row, tt / tb = plane 2j + 1
                if constexpr(WIDE)
                    h = hlerp3x4_paired(t[j].a0, t[j].a1, t[j].b0, t[j].b1, t[j].wtA, t[j].wtB);
                else
                    h = hlerp3x4_halfpaired(t[j].a0, t[j].a0n, t[j].a1, t[j].a1n, t[j].b0, t[j].b0n, t[j].b1, t[j].b1n, t[j].wtA, t[j].wtB);
                const v2f dL = (v2f{h.rt.L, h.tt.L} * t[j].nnb + cL[j]) - v2f{h.rb.L, h.tb.L} * t[j].wb;
                const v2f da = (v2f{h.rt.a, h.tt.a} * t[j].nnb + ca[j]) - v2f{h.rb.a, h.tb.a} * t[j].wb;
                const v2f db = (v2f{h.rt.b, h.tt.b} * t[j].nnb + cb[j]) - v2f{h.rb.b, h.tb.b} * t[j].wb;
                const v2f sq = dL * dL + (da * da + db * db);
                const v2f e = v2f{__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)} * kC + base;
#if AVDM_DEV_TWO_EXP
                const v2f w = {wRl * yk_weight(sqrtf(sq.x) * (1.0f / 65536.0f), dPl, A.invGammaC), wRl * yk_weight(sqrtf(sq.y) * (1.0f / 65536.0f), dPl, A.invGammaC)};
#else
                const v2f w = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
                {
                    const v2f VT = (v2f{h.rb.L, h.tb.L} * t[j].wb - v2f{h.rt.L, h.tt.L} * t[j].nnb) * (1.0f / 65536.0f);
                    lit[2 * j].update(VR, VT.x, w.x);
                    lit[2 * j + 1].update(VR, VT.y, w.y);
                }
    }
#endif
                const v2f wgT = w * dL;
                wsum[j] += w;
                s1R[j] = w * dLR + s1R[j];
                s1T[j] += wgT;
                s2R[j] = w * dLR2 + s2R[j];
                s2T[j] = wgT * dL + s2T[j];
                sxy[j] = wgT * dLR + sxy[j];
            }
        };
        if(WSH == 3)
        {
            // 7 taps per row as 2 + 2 + 2 + 1 with fences (see ncc_accumulate_lds_fixed8: an unroll factor with a remainder unrolls the whole
            // row; one sample at a time and 3 + 3 + 1 were measured slower in round 3)
#pragma unroll
            for(int g = 0; g < 3; ++g)
            {
                sample(-3 + 2 * g);
                sample(-2 + 2 * g);
                __builtin_amdgcn_sched_barrier(0);
            }
            sample(3);
            __builtin_amdgcn_sched_barrier(0);
        }
        else
        {
#pragma unroll kNccMultiUnroll
            for(int xp = -wsh; xp <= wsh; ++xp)
                sample(xp);
        }
    }

    auto finish = [&](float ws, float x1, float y1, float xx, float yy, float xy) __attribute__((always_inline)) -> float {
        const float iw = fast_rcp(ws);
        const float varXW = (xx - x1 * x1 * iw) * iw;
        const float varYW = (yy - y1 * y1 * iw) * iw;
        const float varXYW = (xy - x1 * y1 * iw) * iw;
        const float rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
        const float s = isfinite(rawSim) ? -rawSim : 1.0f;
        return TInvert ? refine_sigmoid(s) : s;
    };
    sim0 = finish(wsum[0].x, s1R[0].x, s1T[0].x, s2R[0].x, s2T[0].x, sxy[0].x);
    sim1 = finish(wsum[0].y, s1R[0].y, s1T[0].y, s2R[0].y, s2T[0].y, sxy[0].y);
    sim2 = finish(wsum[1].x, s1R[1].x, s1T[1].x, s2R[1].x, s2T[1].x, sxy[1].x);
    sim3 = finish(wsum[1].y, s1R[1].y, s1T[1].y, s2R[1].y, s2T[1].y, sxy[1].y);
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
    auto finish_lit = [&](const SimStatLit& st) __attribute__((always_inline)) -> float {
        const float rawSim = st.raw_sim();
        const float s = isfinite(rawSim) ? -rawSim : 1.0f;
        return TInvert ? refine_sigmoid(s) : s;
    };
    sim0 = finish_lit(lit[0]);
    sim1 = finish_lit(lit[1]);
    sim2 = finish_lit(lit[2]);
    sim3 = finish_lit(lit[3]);
    }
#endif
}

// ncc_accumulate_lds_fixed8_quad generalised to NPAIR pairs of adjacent planes per pass (a copy: the four-plane pass above is the measured default
// and stays byte for byte what it is).  NPAIR = 4 = EIGHT planes per pass: the R side of a sample — 37 of the 518 VALU instructions of twelve
// plane-samples are per SAMPLE — amortised over twice the planes.  Behind AVDM_SIM_PLANES8=1: -4 % on the SGM sweep (see AVDM_NCC_MULTI_PIPE
// above), volumes equal to the default's to the storage quantum (tests/test_gpu_parity.py::test_sgm_similarity_experiments_equal_the_default);
// not the default until the parity tables have been re-measured with it (the R side now comes from a plane up to four depth steps away).
template <int WSH, bool TInvert, bool PAIRED, int RP, bool REC12, int NPAIR>
__device__ __forceinline__ void ncc_accumulate_lds_fixed8_multi(f3 rax, f3 ray, f3 hr0, const QuadPlane (&q)[2 * NPAIR], f3 Bt, f3 htB, f3 htA, const NccArgs& A,
                                                                const PatchTable& tab, const LdsWindows& Wn, float4 rcCenter, float (&sim)[2 * NPAIR])
{
    const int wsh = WSH > 0 ? WSH : A.wsh;
    const int n = 2 * wsh + 1;
    const float S16 = 65536.0f;
    v2f tt[NPAIR], axS[NPAIR], ayS[NPAIR], az[NPAIR], cL[NPAIR], ca[NPAIR], cb[NPAIR];
    auto set_pair = [&](int j, const QuadPlane& a, const QuadPlane& b) __attribute__((always_inline)) {
        tt[j] = v2f{a.t, b.t};
        axS[j] = v2f{a.tax.x, b.tax.x} * A.tcSx;
        ayS[j] = v2f{a.tax.y, b.tax.y} * A.tcSy;
        az[j] = v2f{a.tax.z, b.tax.z};
        cL[j] = v2f{a.c.x, b.c.x} * S16;
        ca[j] = v2f{a.c.y, b.c.y} * S16;
        cb[j] = v2f{a.c.z, b.c.z} * S16;
    };
#pragma unroll
    for(int j = 0; j < NPAIR; ++j)
        set_pair(j, q[2 * j], q[2 * j + 1]);
    // R side, {x, y} of the one image
    const v2f raS = {rax.x * A.rcSx, rax.y * A.rcSy};
    const v2f rS = {A.rcSx, A.rcSy}, rO = {A.rcOx, A.rcOy};
    const v2f rcLa = v2f{rcCenter.x, rcCenter.y} * S16;
    const float rcb = rcCenter.z * S16;
    const float kC = A.negInvGammaC_log2e * (1.0f / 65536.0f);
    constexpr float recB = PAIRED ? 16.0f : (REC12 ? 12.0f : 8.0f);
    constexpr bool WIDE = PAIRED || REC12; // a tap's record IS its three dot2 operands

    v2f wsum[NPAIR], s1R[NPAIR], s1T[NPAIR], s2R[NPAIR], s2T[NPAIR], sxy[NPAIR];
#pragma unroll
    for(int j = 0; j < NPAIR; ++j)
        wsum[j] = s1R[j] = s1T[j] = s2R[j] = s2T[j] = sxy[j] = v2f{0.f, 0.f};
#if AVDM_DEV_UNSHIFTED_SUMS
    SimStatLit lit[2 * NPAIR];
#endif

    struct RTaps
    {
        typename std::conditional<PAIRED || REC12, uint4, uint2>::type r0, r1;
        unsigned r0n, r1n;
        float rNy, rWy;
        v2h wr;
    };
    struct TTaps
    {
        typename std::conditional<PAIRED || REC12, uint4, uint2>::type a0, a1, b0, b1;
        unsigned a0n, a1n, b0n, b1n;
        v2f nnb, wb;
        v2h wtA, wtB;
    };

#pragma unroll 1
    for(int yp = -wsh; yp <= wsh; ++yp)
    {
        const float fy = (float)yp;
        // row of plane k: htA + t_k * (htB + fy * Bt), the texel-space scale folded into x and y
        const f3 sB = f3{fmaf(fy, Bt.x, htB.x), fmaf(fy, Bt.y, htB.y), fmaf(fy, Bt.z, htB.z)};
        v2f rowx[NPAIR], rowy[NPAIR], rowz[NPAIR];
#pragma unroll
        for(int j = 0; j < NPAIR; ++j)
        {
            rowx[j] = (tt[j] * sB.x + htA.x) * A.tcSx;
            rowy[j] = (tt[j] * sB.y + htA.y) * A.tcSy;
            rowz[j] = tt[j] * sB.z + htA.z;
        }
        const v2f rrow = (fy * v2f{ray.x, ray.y} + v2f{hr0.x, hr0.y}) * rS;
        const float rrowz = fmaf(fy, ray.z, hr0.z);
        const float* trow = tab.c + (yp + wsh) * n + wsh;
        auto fetch_r = [&](int xp) __attribute__((always_inline)) -> RTaps {
            RTaps t;
            const float fx = (float)xp;
            const float rinv = proj_rcp(fmaf(fx, rax.z, rrowz));
            const v2f rXY = (fx * raS + rrow) * rinv + rO;
            const v2f rF = floor2(rXY);
            const v2f rW = quant256(rXY - rF);                           // {A, B} of quant8(), in units of 1/256
            const v2f rN = rW * v2f{-1.0f, 1.0f} + v2f{256.0f, -256.0f}; // {256 - A, -(256 - B)}
            const unsigned oR = (unsigned)(int)fmaf(rF.y, Wn.rPitchBF, fmaf(rF.x, recB, Wn.rOffB));
            __builtin_assume(oR < 65536u);
            t.wr = pk_half_weights(rN.x, rW.x);
            t.rNy = rN.y;
            t.rWy = rW.y;
            const unsigned oRb = RP > 0 ? oR + (unsigned)(RP * (PAIRED ? 16 : (REC12 ? 12 : 8))) : oR + Wn.rPitchB;
            if constexpr(PAIRED)
            {
                t.r0 = lds_record(oR), t.r1 = lds_record(oRb);
                t.r0n = t.r1n = 0u;
            }
            else if constexpr(REC12)
            {
                t.r0 = lds_record12(oR), t.r1 = lds_record12(oRb);
                t.r0n = t.r1n = 0u;
            }
            else
            {
                t.r0 = lds_texel(oR), t.r1 = lds_texel(oRb);
                t.r0n = lds_u32(oR + 12u), t.r1n = lds_u32(oRb + 12u);
            }
            return t;
        };
        auto fetch_t = [&](int xp, int j) __attribute__((always_inline)) -> TTaps {
            TTaps t;
            const float fx = (float)xp;
            const v2f hz = fx * az[j] + rowz[j];
            const v2f inv = {proj_rcp(hz.x), proj_rcp(hz.y)};
            const v2f X = (fx * axS[j] + rowx[j]) * inv + A.tcOx;
            const v2f Y = (fx * ayS[j] + rowy[j]) * inv + A.tcOy;
            const v2f fX = floor2(X), fY = floor2(Y);
            const v2f wa = quant256(X - fX);
            t.wb = quant256(Y - fY);
            const v2f na = 256.0f - wa;
            t.nnb = t.wb - 256.0f;
            const v2f oidx = fY * Wn.tPitchBF + (fX * recB + Wn.tOffB);
            const unsigned oA = (unsigned)(int)oidx.x, oB = (unsigned)(int)oidx.y;
            t.wtA = pk_half_weights(na.x, wa.x);
            t.wtB = pk_half_weights(na.y, wa.y);
            if constexpr(PAIRED)
            {
                t.a0 = lds_record(oA), t.a1 = lds_record(oA + Wn.tPitchB), t.b0 = lds_record(oB), t.b1 = lds_record(oB + Wn.tPitchB);
                t.a0n = t.a1n = t.b0n = t.b1n = 0u;
            }
            else if constexpr(REC12)
            {
                t.a0 = lds_record12(oA), t.a1 = lds_record12(oA + Wn.tPitchB), t.b0 = lds_record12(oB), t.b1 = lds_record12(oB + Wn.tPitchB);
                t.a0n = t.a1n = t.b0n = t.b1n = 0u;
            }
            else
            {
                t.a0 = lds_texel(oA), t.a1 = lds_texel(oA + Wn.tPitchB), t.b0 = lds_texel(oB), t.b1 = lds_texel(oB + Wn.tPitchB);
                t.a0n = lds_u32(oA + 12u), t.a1n = lds_u32(oA + Wn.tPitchB + 12u), t.b0n = lds_u32(oB + 12u), t.b1n = lds_u32(oB + Wn.tPitchB + 12u);
            }
            return t;
        };
        auto sample = [&](int xp) __attribute__((always_inline)) {
            const RTaps r = fetch_r(xp);
            Lab3x2 hr;
            if constexpr(WIDE)
                hr = hlerp3x2_paired(r.r0, r.r1, r.wr);
            else
                hr = hlerp3x2_halfpaired(r.r0, r.r0n, r.r1, r.r1n, r.wr);
            const v2f dRLa = (v2f{hr.t.L, hr.t.a} * r.rNy + rcLa) - v2f{hr.b.L, hr.b.a} * r.rWy;
            const float dRb = fmaf(hr.t.b, r.rNy, rcb) - hr.b.b * r.rWy;
            const v2f qR = dRLa * dRLa;
            const float base = fmaf(__builtin_amdgcn_sqrtf(fmaf(dRb, dRb, qR.x + qR.y)), kC, -trow[xp]);
            const float dLR = dRLa.x;
            const float dLR2 = dLR * dLR;
#if AVDM_DEV_TWO_EXP
            const float dPl = sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP;
            const float wRl = yk_weight(sqrtf(fmaf(dRb, dRb, qR.x + qR.y)) * (1.0f / 65536.0f), dPl, A.invGammaC);
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
            const float VR = (hr.b.L * r.rWy - hr.t.L * r.rNy) * (1.0f / 65536.0f); // the bilinear L of R itself
#endif
            // rotating: the taps of pair j + 1 are requested before pair j is consumed (all four pairs' taps in flight would be 4 x 18 registers)
            TTaps t[NPAIR];
#pragma unroll
            for(int j = 0; j < NPAIR; ++j)
            {
                if(j == 0)
                    t[0] = fetch_t(xp, 0);
                if(j + 1 < NPAIR)
                    t[j + 1] = fetch_t(xp, j + 1);
                Lab3x4 h; // rt / rb = plane 2j top / bottom row, tt / tb = plane 2j + 1
                if constexpr(WIDE)
                    h = hlerp3x4_paired(t[j].a0, t[j].a1, t[j].b0, t[j].b1, t[j].wtA, t[j].wtB);
                else
                    h = hlerp3x4_halfpaired(t[j].a0, t[j].a0n, t[j].a1, t[j].a1n, t[j].b0, t[j].b0n, t[j].b1, t[j].b1n, t[j].wtA, t[j].wtB);
                const v2f dL = (v2f{h.rt.L, h.tt.L} * t[j].nnb + cL[j]) - v2f{h.rb.L, h.tb.L} * t[j].wb;
                const v2f da = (v2f{h.rt.a, h.tt.a} * t[j].nnb + ca[j]) - v2f{h.rb.a, h.tb.a} * t[j].wb;
                const v2f db = (v2f{h.rt.b, h.tt.b} * t[j].nnb + cb[j]) - v2f{h.rb.b, h.tb.b} * t[j].wb;
                const v2f sq = dL * dL + (da * da + db * db);
                const v2f e = v2f{__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)} * kC + base;
#if AVDM_DEV_TWO_EXP
                const v2f w = {wRl * yk_weight(sqrtf(sq.x) * (1.0f / 65536.0f), dPl, A.invGammaC), wRl * yk_weight(sqrtf(sq.y) * (1.0f / 65536.0f), dPl, A.invGammaC)};
#else
                const v2f w = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
#endif
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
                {
                    const v2f VT = (v2f{h.rb.L, h.tb.L} * t[j].wb - v2f{h.rt.L, h.tt.L} * t[j].nnb) * (1.0f / 65536.0f);
                    lit[2 * j].update(VR, VT.x, w.x);
                    lit[2 * j + 1].update(VR, VT.y, w.y);
                }
    }
#endif
                const v2f wgT = w * dL;
                wsum[j] += w;
                s1R[j] = w * dLR + s1R[j];
                s1T[j] += wgT;
                s2R[j] = w * dLR2 + s2R[j];
                s2T[j] = wgT * dL + s2T[j];
                sxy[j] = wgT * dLR + sxy[j];
                if(j + 1 < NPAIR)
                    __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(NPAIR >= 3, "the eight-plane pass (the four-plane pass is ncc_accumulate_lds_fixed8_quad)");
        // a rolled loop of one sample per iteration for 7-tap rows too (the unrolled forms of the four-plane pass spill at eight planes)
        {
#pragma unroll(kNccOctoUnroll)
            for(int xp = -wsh; xp <= wsh; ++xp)
                sample(xp);
        }
    }

    auto finish = [&](float ws, float x1, float y1, float xx, float yy, float xy) __attribute__((always_inline)) -> float {
        const float iw = fast_rcp(ws);
        const float varXW = (xx - x1 * x1 * iw) * iw;
        const float varYW = (yy - y1 * y1 * iw) * iw;
        const float varXYW = (xy - x1 * y1 * iw) * iw;
        const float rawSim = varXYW * __builtin_amdgcn_rsqf(varXW * varYW);
        const float s = isfinite(rawSim) ? -rawSim : 1.0f;
        return TInvert ? refine_sigmoid(s) : s;
    };
#pragma unroll
    for(int j = 0; j < NPAIR; ++j)
    {
        sim[2 * j] = finish(wsum[j].x, s1R[j].x, s1T[j].x, s2R[j].x, s2T[j].x, sxy[j].x);
        sim[2 * j + 1] = finish(wsum[j].y, s1R[j].y, s1T[j].y, s2R[j].y, s2T[j].y, sxy[j].y);
    }
#if AVDM_DEV_UNSHIFTED_SUMS
    if constexpr(kUnshifted<TInvert>)
    {
    auto finish_lit = [&](const SimStatLit& st) __attribute__((always_inline)) -> float {
        const float rawSim = st.raw_sim();
        const float s = isfinite(rawSim) ? -rawSim : 1.0f;
        return TInvert ? refine_sigmoid(s) : s;
    };
#pragma unroll
    for(int j = 0; j < 2 * NPAIR; ++j)
        sim[j] = finish_lit(lit[j]);
    }
#endif
}

// recB = bytes per window record: 8 (one texel) or 16 (paired layout); rcap stays in 8-byte units
__device__ __forceinline__ LdsWindows make_windows(const uint2* smem, int rcap, int rPitch, int rx0, int ry0, int tPitch, int tx0, int ty0, int recB)
{
    LdsWindows W;
    const int base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem; // LDS byte address of the dynamic segment
    W.rPitchB = (unsigned)(rPitch * recB);
    W.tPitchB = (unsigned)(tPitch * recB);
    W.rPitchBF = (float)(rPitch * recB);
    W.tPitchBF = (float)(tPitch * recB);
    W.rOffB = (float)(base - recB * (ry0 * rPitch + rx0));
    W.tOffB = (float)(base + 8 * rcap - recB * (ty0 * tPitch + tx0));
    return W;
}

// tex_bilinear_px<true>() from a staged window of PAIRED records (stage_window_paired): the same fp16 texels, the same quantised weights,
// the same blend — four texels out of two 16-byte records instead of four global loads behind 64-bit address arithmetic.  (x, y) are
// texel-space coordinates whose taps lie inside the window (never its last column: that record pairs the texel with itself).
__device__ __forceinline__ float4 lds_center_paired(const uint2* win, int pitch, int x0, int y0, float x, float y)
{
    const float fx = floorf(x), fy = floorf(y);
    const float a = quant8(x - fx), b = quant8(y - fy);
    const int i = (int)fx - x0, j = (int)fy - y0;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)win;
    const unsigned addr = base + (unsigned)(j * pitch + i) * 16u;
    const uint4 r0 = lds_record(addr), r1 = lds_record(addr + (unsigned)pitch * 16u);
    auto lo = [](unsigned v) __attribute__((always_inline)) { return __half2float(__ushort_as_half((unsigned short)(v & 0xffffu))); };
    auto hi = [](unsigned v) __attribute__((always_inline)) { return __half2float(__ushort_as_half((unsigned short)(v >> 16))); };
    const float4 t00 = make_float4(lo(r0.x), lo(r0.y), lo(r0.z), lo(r0.w)), t10 = make_float4(hi(r0.x), hi(r0.y), hi(r0.z), hi(r0.w));
    const float4 t01 = make_float4(lo(r1.x), lo(r1.y), lo(r1.z), lo(r1.w)), t11 = make_float4(hi(r1.x), hi(r1.y), hi(r1.z), hi(r1.w));
    return bilinear_blend(t00, t10, t01, t11, a, b);
}

__device__ __forceinline__ void pixel_of_lane(int& tx, int& ty)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    tx = (w & 1) * 8 + (lane & 7);
    ty = (w >> 1) * 8 + (lane >> 3);
}

// ---------------------------------------------------------------------------------------------
// workgroup state shared by the two kernels
// ---------------------------------------------------------------------------------------------
#define AVDM_MAX_CHUNK 8
#define AVDM_CHUNK_BOX AVDM_MAX_CHUNK // entry of the window shared by all planes of a chunk
struct BlockShared
{
    int box[AVDM_MAX_CHUNK + 1][4]; // per plane of the chunk (+ the chunk itself): min x, min y, max x, max y (floor of the texel-space tap positions in T)
    int bad[AVDM_MAX_CHUNK + 1];    // per plane: some lane's R taps leave the staged R tile
};

struct RTile
{
    int x0, y0, w, h, pitch;
    bool ok;
};

// R footprint of the workgroup: stage pixels [bx, bx+15] x [by, by+15] of the ROI, patch halo wsh + 2 (the border-test margin)
// LDS capacity a window of n records takes, in the 8-byte units of NccArgs::rcap / tcap
__device__ __forceinline__ int lds_units(int n, bool paired, bool rec12) { return paired ? 2 * n : (rec12 ? (3 * n + 1) / 2 : n); }

__device__ __forceinline__ RTile stage_r_tile(uint2* sR, const NccArgs& A, int wsh, int stepXY, avdm_roi_t roi, bool paired, bool halfPaired, int bw = 16,
                                              bool lean = false, bool rec12 = false)
{
    RTile T;
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    const int bx = blockIdx.x * bw, by = blockIdx.y * bw;
    const float pxMin = (float)((int)roi.x.begin + bx) * (float)stepXY;
    const float pxMax = (float)((int)roi.x.begin + min(bx + bw - 1, roiW - 1)) * (float)stepXY;
    const float pyMin = (float)((int)roi.y.begin + by) * (float)stepXY;
    const float pyMax = (float)((int)roi.y.begin + min(by + bw - 1, roiH - 1)) * (float)stepXY;
    const float m = (float)wsh + 2.0f;
    int x0 = (int)floorf(fmaf(pxMin - m, A.rcSx, A.rcOx)) - 1;
    int x1 = (int)floorf(fmaf(pxMax + m, A.rcSx, A.rcOx)) + 2;
    int y0 = (int)floorf(fmaf(pyMin - m, A.rcSy, A.rcOy)) - 1;
    int y1 = (int)floorf(fmaf(pyMax + m, A.rcSy, A.rcOy)) + 2;
    x0 = max(x0, 0);
    y0 = max(y0, 0);
    x1 = min(x1, A.rcL.W - 1);
    y1 = min(y1, A.rcL.H - 1);
    T.x0 = x0;
    T.y0 = y0;
    T.w = x1 - x0 + 1;
    T.h = y1 - y0 + 1;
    T.pitch = A.rpitch; // the full-width pitch also where the image border clips the tile
    T.ok = (lean || !A.forceGeneric) && T.w > 1 && T.h > 1 && T.w <= T.pitch && lds_units(T.pitch * T.h, paired, rec12) <= A.rcap;
    if(T.ok)
    {
        if(paired)
            stage_window_paired((uint4*)sR, T.pitch, A.rcL, T.x0, T.y0, T.w, T.h);
        else if(rec12)
            stage_window_rec12((Rec12*)sR, T.pitch, A.rcL, T.x0, T.y0, T.w, T.h);
        else if(halfPaired)
            stage_window_halfpaired(sR, T.pitch, A.rcL, T.x0, T.y0, T.w, T.h);
        else
            stage_window(sR, T.pitch, A.rcL, T.x0, T.y0, T.w, T.h);
    }
    return T;
}

// per-lane part of the window search: min/max of the 4 projected patch corners in T, and whether the R taps stay in the R tile
__device__ __forceinline__ void corner_boxes(const PatchProj& Q, const NccArgs& A, int wsh, const RTile& R, float& tminx, float& tminy, float& tmaxx,
                                             float& tmaxy, bool& rInside)
{
    tminx = tminy = INFINITY;
    tmaxx = tmaxy = -INFINITY;
    float rminx = INFINITY, rminy = INFINITY, rmaxx = -INFINITY, rmaxy = -INFINITY;
#pragma unroll
    for(int cy = -1; cy <= 1; cy += 2)
    {
        f3 hrRow, htRow;
        row_of(Q, (float)(cy * wsh), hrRow, htRow);
#pragma unroll
        for(int cx = -1; cx <= 1; cx += 2)
        {
            float rX, rY, tX, tY;
            sample_pos(Q, A, hrRow, htRow, (float)(cx * wsh), rX, rY, tX, tY);
            rminx = fminf(rminx, rX);
            rmaxx = fmaxf(rmaxx, rX);
            rminy = fminf(rminy, rY);
            rmaxy = fmaxf(rmaxy, rY);
            tminx = fminf(tminx, tX);
            tmaxx = fmaxf(tmaxx, tX);
            tminy = fminf(tminy, tY);
            tmaxy = fmaxf(tmaxy, tY);
        }
    }
    // the taps themselves (texels floor(x), floor(x) + 1) must be inside the image and inside the staged tile; one more texel
    // of slack on each side for the interior taps (they lie in the corners' hull up to rounding), clipped at the image edge
    const float W1 = (float)(A.rcL.W - 1), H1 = (float)(A.rcL.H - 1);
    const float x0 = floorf(rminx), x1 = floorf(rmaxx) + 1.0f, y0 = floorf(rminy), y1 = floorf(rmaxy) + 1.0f;
    rInside = x0 >= 0.0f && y0 >= 0.0f && x1 <= W1 && y1 <= H1 && (fmaxf(x0 - 1.0f, 0.0f) >= (float)R.x0) &&
              (fminf(x1 + 1.0f, W1) <= (float)(R.x0 + R.w - 1)) && (fmaxf(y0 - 1.0f, 0.0f) >= (float)R.y0) &&
              (fminf(y1 + 1.0f, H1) <= (float)(R.y0 + R.h - 1));
}

__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, dpp_f32<0x111>(v, v));
    v = fmaxf(v, dpp_f32<0x112>(v, v));
    v = fmaxf(v, dpp_f32<0x114>(v, v));
    v = fmaxf(v, dpp_f32<0x118>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_f32<0x143, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += dpp_f32<0x111>(0.0f, v);
    v += dpp_f32<0x112>(0.0f, v);
    v += dpp_f32<0x114>(0.0f, v);
    v += dpp_f32<0x118>(0.0f, v);
    v += dpp_f32<0x142, 0xa>(0.0f, v);
    v += dpp_f32<0x143, 0xc>(0.0f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Outliers do not stretch the workgroup's T window.  A pixel whose depth is far off its neighbours' (an SGM outlier under the Refine sweep, a
// depth edge) projects its patch tens of texels away from theirs in a wide-baseline T view; one such lane used to make the hull of the 256
// lanes exceed the LDS budget and send the WHOLE workgroup to the global-memory taps (~4 x slower; `profiles/r03_i_bench_stats.json`: ~11 %
// of the Refine workgroups of the bench's outer cameras).  A lane whose hull centre lies farther than `radius` texels from the mean centre
// of its wave's participating lanes stays out of the hull: only ITS wave takes the global-memory taps for the planes it is valid on.
__device__ __forceinline__ bool wave_inlier(bool part, float cx, float cy, float radius)
{
    const float n = wave_sum_f32(part ? 1.0f : 0.0f);
    const float mx = wave_sum_f32(part ? cx : 0.0f) / fmaxf(n, 1.0f), my = wave_sum_f32(part ? cy : 0.0f) / fmaxf(n, 1.0f);
    return part && fabsf(cx - mx) <= radius && fabsf(cy - my) <= radius;
}

// workgroup reduction of the lanes' T boxes into sh.box[k] (first half; the caller synchronises afterwards)
__device__ __forceinline__ void publish_box(BlockShared& sh, int k, bool valid, float tminx, float tminy, float tmaxx, float tmaxy, bool rInside)
{
    const float mnx = wave_min_f32(valid ? tminx : INFINITY);
    const float mny = wave_min_f32(valid ? tminy : INFINITY);
    const float mxx = wave_max_f32(valid ? tmaxx : -INFINITY);
    const float mxy = wave_max_f32(valid ? tmaxy : -INFINITY);
    const bool anyBad = __any(valid && !(rInside && isfinite(tminx) && isfinite(tminy) && isfinite(tmaxx) && isfinite(tmaxy)));
    if((threadIdx.x & 63) == 0)
    {
        if(anyBad)
            atomicOr(&sh.bad[k], 1);
        if(mnx <= mxx && mny <= mxy && fabsf(mnx) < 1.0e8f && fabsf(mny) < 1.0e8f && fabsf(mxx) < 1.0e8f && fabsf(mxy) < 1.0e8f)
        {
            atomicMin(&sh.box[k][0], (int)floorf(mnx));
            atomicMin(&sh.box[k][1], (int)floorf(mny));
            atomicMax(&sh.box[k][2], (int)floorf(mxx));
            atomicMax(&sh.box[k][3], (int)floorf(mxy));
        }
    }
}

struct TWindow
{
    int x0, y0, w, h, pitch;
    bool ok;
    bool tooLarge; // not ok because the window exceeds the LDS budget (the only failure a smaller hull can cure)
};

// second half (after the barrier): decide — uniformly for the workgroup — whether plane k runs from LDS, and stage the T window
__device__ __forceinline__ TWindow stage_t_window(uint2* sT, const BlockShared& sh, int k, const NccArgs& A, bool rTileOk, bool paired, bool halfPaired,
                                                  bool lean = false, bool rec12 = false, int leanStatBase = -1)
{
    TWindow Wd;
    const int mnx = sh.box[k][0], mny = sh.box[k][1], mxx = sh.box[k][2], mxy = sh.box[k][3];
    Wd.ok = false;
    Wd.tooLarge = false;
    Wd.x0 = Wd.y0 = Wd.w = Wd.h = Wd.pitch = 0;
    int reason = 1; // 0 = LDS path, 1 = R tile unusable / nothing valid, 2 = T taps leave the image, 3 = T window exceeds the LDS budget
    if(rTileOk && !sh.bad[k] && mnx != INT_MAX && mxx != INT_MIN)
    {
        // the taps (texels floor(x), floor(x) + 1) of the packed layouts may lie up to kOutside texels outside the T image (their windows hold
        // the clamped texels there: stage_window_paired); the other layouts need them inside.  The window adds one texel of slack on each side
        // (see corner_boxes), clipped at the image edge where the layout cannot leave it
        const bool mayLeave = AVDM_WINDOWS_OUTSIDE && (paired || rec12);
        const int kOutside = mayLeave ? 24 : 0;
        const bool inImage = mnx >= -kOutside && mny >= -kOutside && mxx + 1 <= A.tcL.W - 1 + kOutside && mxy + 1 <= A.tcL.H - 1 + kOutside;
        Wd.x0 = mayLeave ? mnx - 1 : max(mnx - 1, 0);
        Wd.y0 = mayLeave ? mny - 1 : max(mny - 1, 0);
        const int x1 = mayLeave ? mxx + 2 : min(mxx + 2, A.tcL.W - 1), y1 = mayLeave ? mxy + 2 : min(mxy + 2, A.tcL.H - 1);
        Wd.w = x1 - Wd.x0 + 1;
        Wd.h = y1 - Wd.y0 + 1;
        Wd.pitch = lds_pitch_for(Wd.w);
        const bool fits = Wd.w <= 4096 && Wd.h <= 4096 && lds_units(Wd.pitch * Wd.h, paired, rec12) <= A.tcap;
        Wd.ok = inImage && fits;
        reason = Wd.ok ? 0 : (inImage ? 3 : 2);
        Wd.tooLarge = inImage && !fits;
    }
    if(Wd.ok)
    {
        if(paired)
            stage_window_paired((uint4*)sT, Wd.pitch, A.tcL, Wd.x0, Wd.y0, Wd.w, Wd.h);
        else if(rec12)
            stage_window_rec12((Rec12*)sT, Wd.pitch, A.tcL, Wd.x0, Wd.y0, Wd.w, Wd.h);
        else if(halfPaired)
            stage_window_halfpaired(sT, Wd.pitch, A.tcL, Wd.x0, Wd.y0, Wd.w, Wd.h);
        else
            stage_window(sT, Wd.pitch, A.tcL, Wd.x0, Wd.y0, Wd.w, Wd.h);
    }
    if(!lean && A.stats != nullptr && threadIdx.x == 0)
        atomicAdd(A.stats + reason, 1u);
#if AVDM_LEAN_STATS
    if(leanStatBase >= 0 && threadIdx.x == 0)
        atomicAdd(&g_leanStats[leanStatBase + reason], 1u);
#endif
    return Wd;
}

__device__ __forceinline__ void init_shared(BlockShared& sh)
{
    if(threadIdx.x < (AVDM_MAX_CHUNK + 1) * 4)
        sh.box[threadIdx.x >> 2][threadIdx.x & 3] = (threadIdx.x & 2) ? INT_MIN : INT_MAX;
    if(threadIdx.x < AVDM_MAX_CHUNK + 1)
        sh.bad[threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// SGM similarity: best / second-best uint8 volumes, 4 planes per lane per launch-z
// ---------------------------------------------------------------------------------------------
// PLANES = planes per pass over the patch on the packed chunk-window path: 1, 4 (ncc_accumulate_lds_fixed8_quad: the whole chunk in one
// pass) or 8 (..._multi<4>: two chunks of the workgroup, the default since round 5); chunks a pass cannot take — a plane range that ends
// inside it, a wave with a lane outside the window — run four planes or one plane per pass
// (Rounds 3-4 also carried a two-launch form of the default instantiations — a fast kernel and a fix-up kernel over the same grid,
// AVDM_SIM_SPLIT — measured 3 % slower than the combined kernel (profiles/r03_o_split_ab.txt) and removed in round 5.)
template <bool FIXED8, int WSH, bool PAIRED, int RP = 0, int PLANES = 1, bool REC12 = false>
__global__ void __launch_bounds__(256, AVDM_SIM_WAVES_PER_SIMD)
  similarity_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, long long pitch_y, int pitch_x, const float* __restrict__ depths,
                    avdm_camera_t rc, avdm_camera_t tc, NccArgs A, PatchTable tab, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    extern __shared__ __attribute__((aligned(16))) uint2 smem[];
    uint2* sR = smem;
    uint2* sT = smem + A.rcap;
    __shared__ BlockShared sh;
    const int wsh = WSH > 0 ? WSH : A.wsh;

    int tx, ty;
    pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    const bool inRoi = vx < roi.x.end - roi.x.begin && vy < roi.y.end - roi.y.begin;
    // kSgmChunksPerWg chunks of 4 planes per workgroup (one R tile, one T window, one pixel set-up)
    const unsigned z0 = ((zBegin >> 2) + blockIdx.z * kSgmChunksPerWg) << 2;
    constexpr unsigned kPlanesPerWg = 4u * kSgmChunksPerWg;

    // LEAN = the default instantiations (RP > 0): the A/B switches of NccArgs are compile-time constants at their defaults there (the host
    // only launches them in that state), which removes the plain-arithmetic LDS path and the counters from the hot kernels
    constexpr bool LEAN = RP > 0;
    const bool noPacked = LEAN ? false : (A.noPacked != 0);
    const bool useChunkWindow = LEAN ? true : (A.chunkWindow != 0);
    const bool usePlanePairs = LEAN ? true : (A.planePairs != 0);
    // paired LDS records only feed the packed FIXED8 path (uniform)
    static_assert(!(REC12 && PAIRED), "12-byte records or 16-byte records");
    const bool paired = PAIRED && FIXED8 && !noPacked;
    const bool rec12 = REC12 && FIXED8 && !noPacked;                  // 12-byte records (see stage_window_rec12)
    const bool halfPaired = !PAIRED && !REC12 && FIXED8 && !noPacked; // the packed path without room for wider records
    init_shared(sh);
    const RTile R = stage_r_tile(sR, A, wsh, stepXY, roi, paired, halfPaired, 16, LEAN, rec12);
    __syncthreads();

    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;

    // pixel ray (shared by the planes of the chunk): get3DPointForPixelAndFrontoParellePlaneRC restated
    const f3 C = ld3(rc.C), Z = ld3(rc.ZVect);
    const f3 v = normalize(M3x3mulV2(rc.iP, x, y));
    const float dnC = dot(Z, C), dnv = dot(Z, v);

    // R side of the validity test (Patch.cuh:486-496, 523-526): the patch centre lies on the ray of pixel (x, y), so its R
    // projection IS (x, y); using the exact pixel makes the border test deterministic on the knife-edge rows where
    // x == wsh + 2 (DESIGN.md "knife-edge rows")
    const float dd = (float)wsh + 2.0f;
    bool rValid = inRoi && !((x < dd) || (x > A.rcW1 - dd) || (y < dd) || (y > A.rcH1 - dd));
    // knife-edge rows / columns (my pixel lies EXACTLY wsh + 2 from an image border): the reference's own border test, per plane (lit::)
    // — evaluated once for the planes of this workgroup, into one bit per plane, ahead of the plane loops (a wave without such a lane skips it)
    unsigned knifeMask = 0xffffffffu;
    {
        const bool knife = rValid && (x == dd || x == A.rcW1 - dd || y == dd || y == A.rcH1 - dd);
        if(AVDM_KNIFE_LITERAL && __any(knife)) // wave-uniform
        {
            if(knife)
            {
                knifeMask = 0u;
#pragma unroll 1
                for(unsigned k = 0; k < 4u * kSgmChunksPerWg; ++k)
                {
                    const unsigned vz = z0 + k;
                    if(vz >= zBegin && vz < zEnd && lit::sgm_r_inside(rc, x, y, depths[vz], dd, A.rcW1, A.rcH1))
                        knifeMask |= 1u << k;
                }
            }
        }
    }
    auto r_border_ok = [&](unsigned vz) __attribute__((always_inline)) -> bool {
        const unsigned k = vz - z0;
        return k >= 32u || ((knifeMask >> k) & 1u) != 0u;
    };
    float4 rcCenter = make_float4(0.f, 0.f, 0.f, 0.f);
    if(rValid)
    {
        rcCenter = tex_bilinear_px<FIXED8>(A.rcL, fmaf(x, A.rcSx, A.rcOx), fmaf(y, A.rcSy, A.rcOy));
        rValid = !(rcCenter.w < (255.f * 0.9f));
    }

    uint8_t* const pb0 = best + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
    uint8_t* const ps0 = second + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;

    const RayConsts RK = make_ray_consts(rc, tc, C, v, x, y);
    // geometry of my patch on plane vz (see the Refine kernel): false when the patch centre fails the border test in T
    auto plane_geometry = [&](unsigned vz, PatchProj& Q, float& tpx, float& tpy) __attribute__((always_inline)) -> bool {
        const float depthPlane = depths[vz];
        const f3 planep = C + Z * depthPlane;
        const float kk = (dot(planep, Z) - dnC) / dnv;
        const f3 p = C + v * kk;
        // on my pixel's ray: pixel size, view direction and centre projections from the per-lane constants (make_ray_consts)
        const float pd = RK.pixK * kk;
        f3 ax, ay;
        {
            const f3 v1 = f3{-v.x, -v.y, -v.z}; // normalize(C - p)
            const f3 v2 = normalize(ld3(tc.C) - p);
            ay = normalize(cross(v1, v2));
            const f3 n = normalize((v1 + v2) * 0.5f);
            ax = normalize(cross(ay, n));
        }
        const float tw = kk * RK.hrW;
        Q = make_patch_proj_on_ray(rc, tc, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z}, fma3(kk, RK.htB, RK.htA), ax, ay, pd);
        const float it0 = proj_rcp(Q.ht0.z);
        tpx = Q.ht0.x * it0;
        tpy = Q.ht0.y * it0;
        return r_border_ok(vz) && !((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd));
    };

    // ONE T window for the 4 planes of the chunk (see the Refine kernel; here for both record layouts of the packed path, the centre
    // colour still comes from global memory — the half-paired records carry no alpha — but no longer decides who contributes to the box)
    bool chunkWin = false;
    TWindow Wc;
    Wc.ok = false;
    Wc.tooLarge = false;
    Wc.x0 = Wc.y0 = Wc.w = Wc.h = Wc.pitch = 0;
    float extX = 0.f, extY = 0.f;
    bool lanePart = false;
    if(FIXED8 && (paired || halfPaired || rec12) && useChunkWindow)
    {
        const unsigned ka = z0 > zBegin ? z0 : zBegin, kbEnd = (z0 + kPlanesPerWg < zEnd) ? z0 + kPlanesPerWg : zEnd;
        if(ka < kbEnd) // uniform
        {
            float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
            bool part = false, rIn = true;
#pragma unroll 1
            for(int e = 0; e < 2; ++e)
            {
                const unsigned vz = e == 0 ? ka : kbEnd - 1u;
                if(e == 1 && vz == ka)
                    break;
                PatchProj Q;
                float tpx, tpy;
                if(rValid && plane_geometry(vz, Q, tpx, tpy))
                {
                    float cx0, cy0, cx1, cy1;
                    bool ri;
                    corner_boxes(Q, A, wsh, R, cx0, cy0, cx1, cy1, ri);
                    bx0 = fminf(bx0, cx0);
                    by0 = fminf(by0, cy0);
                    bx1 = fmaxf(bx1, cx1);
                    by1 = fmaxf(by1, cy1);
                    extX = fmaxf(extX, 0.5f * (cx1 - cx0));
                    extY = fmaxf(extY, 0.5f * (cy1 - cy0));
                    rIn = rIn && ri;
                    part = true;
                }
            }
            publish_box(sh, AVDM_CHUNK_BOX, part, bx0 - 1.0f, by0 - 1.0f, bx1 + 1.0f, by1 + 1.0f, rIn);
            __syncthreads();
            Wc = stage_t_window(sT, sh, AVDM_CHUNK_BOX, A, R.ok, paired, halfPaired, LEAN, rec12, LEAN ? 16 : -1);
            __syncthreads();
            if(Wc.tooLarge) // uniform
            {
                if(LEAN)
                    LEANSTAT_WG(6);
                // the hull of ALL lanes does not fit: once more without the outliers (wave_inlier) — their waves then take the global-memory
                // taps on the planes they are valid on, the rest of the workgroup keeps the LDS path
                if(threadIdx.x < 4)
                    sh.box[AVDM_CHUNK_BOX][threadIdx.x] = (threadIdx.x & 2) ? INT_MIN : INT_MAX;
                __syncthreads();
                part = wave_inlier(part, 0.5f * (bx0 + bx1), 0.5f * (by0 + by1), 16.0f * (float)stepXY + 16.0f);
                publish_box(sh, AVDM_CHUNK_BOX, part, bx0 - 1.0f, by0 - 1.0f, bx1 + 1.0f, by1 + 1.0f, rIn);
                __syncthreads();
                Wc = stage_t_window(sT, sh, AVDM_CHUNK_BOX, A, R.ok, paired, halfPaired, LEAN, rec12);
                __syncthreads();
            }
            chunkWin = Wc.ok;
            lanePart = part;
        }
    }
    if(LEAN)
    {
        LEANSTAT_WG(5);
        if(!chunkWin)
            LEANSTAT_WG(4);
    }

#pragma unroll 1
    for(unsigned c = 0; c < kSgmChunksPerWg; ++c)
    {
    const unsigned zc = z0 + 4u * c;
    if(zc >= zEnd) // uniform
        break;
    uint8_t* const pb = pb0 + 4u * c;
    uint8_t* const ps = ps0 + 4u * c;
    unsigned wb = 0, ws = 0;
    if(inRoi)
    {
        wb = *reinterpret_cast<const unsigned*>(pb);
        ws = *reinterpret_cast<const unsigned*>(ps);
    }
    if(!chunkWin && c > 0)
    { // the per-plane fall-back reuses the box entries of the previous chunk
        __syncthreads();
        init_shared(sh);
        __syncthreads();
    }
    // best / second-best update of plane k of the chunk (kernels.cuh:180-200)
    auto commit = [&](int k, float fsim) __attribute__((always_inline)) {
        const unsigned sh8 = 8u * k;
        const unsigned b1 = (wb >> sh8) & 0xffu, b2 = (ws >> sh8) & 0xffu;
        if(fsim < (float)b1)
        {
            ws = (ws & ~(0xffu << sh8)) | (b1 << sh8);
            wb = (wb & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
        }
        else if(fsim < (float)b2)
            ws = (ws & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
    };
    auto to_fsim = [](float s) __attribute__((always_inline)) -> float {
        s = (s + 1.0f) * 0.5f;
        s = fminf(1.0f, fmaxf(0.0f, s));
        return s * 254.0f;
    };
    bool quadDone = false;
    if constexpr(PLANES >= 4 && FIXED8)
    {
        // the four planes of the chunk in one pass over the patch (ncc_accumulate_lds_fixed8_multi): uniform conditions
        if(chunkWin && usePlanePairs && !noPacked)
        {
            // the planes' geometry in the form the pixel ray gives it (ncc_accumulate_lds_fixed8_quad): ay — the normal of the epipolar plane
            // of my ray, cross(v1, C_T - p) for ANY p on the ray — once, M_T (ax d) per plane
            const f3 v1 = f3{-v.x, -v.y, -v.z};
            const f3 ay = normalize(cross(v1, ld3(tc.C) - C));
            const f3 Bt = M3x3mulV3(tc.P, ay) * RK.pixK, Br = M3x3mulV3(rc.P, ay) * RK.pixK;
            auto plane_q = [&](unsigned vz, QuadPlane& q, f3& raxOut, bool& valid, bool& laneLds) __attribute__((always_inline)) {
                const bool inRange = vz >= zBegin && vz < zEnd; // uniform; a plane outside the T camera's range is an invalid plane of the pass
                valid = rValid && inRange;
                q.c = make_float4(0.f, 0.f, 0.f, 0.f);
                const f3 planep = C + Z * depths[inRange ? vz : zBegin];
                const float kk = (dot(planep, Z) - dnC) / dnv;
                q.t = kk;
                const f3 p = C + v * kk;
                const f3 v2 = normalize(ld3(tc.C) - p);
                const f3 nn = normalize((v1 + v2) * 0.5f);
                const f3 axd = normalize(cross(ay, nn)) * (RK.pixK * kk);
                q.tax = M3x3mulV3(tc.P, axd);
                raxOut = axd; // (projected into R only for the ONE plane that provides the R side of the pass: M3x3mulV3(rc.P, .) at the call)
                const f3 ht0 = fma3(kk, RK.htB, RK.htA);
                const float it0 = proj_rcp(ht0.z);
                const float tpx = ht0.x * it0, tpy = ht0.y * it0;
                valid = valid && r_border_ok(vz) && !((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd));
                laneLds = true;
                if(valid)
                {
                    const float cxT = fmaf(tpx, A.tcSx, A.tcOx), cyT = fmaf(tpy, A.tcSy, A.tcOy);
                    q.c = tex_bilinear_px<FIXED8>(A.tcL, cxT, cyT);
                    valid = !(q.c.w < (255.f * 0.4f));
                    laneLds = lanePart && (cxT - extX - 1.0f >= (float)Wc.x0) && (cxT + extX + 2.0f <= (float)(Wc.x0 + Wc.w - 1)) &&
                              (cyT - extY - 1.0f >= (float)Wc.y0) && (cyT + extY + 2.0f <= (float)(Wc.y0 + Wc.h - 1));
                }
                laneLds = __ballot(valid && !laneLds) == 0ull; // wave-uniform choice of the tap source
            };
            // EIGHT planes per pass (AVDM_SIM_PLANES8=1, experimental): this chunk and the next one together when both lie in the T camera's
            // range; the R side comes from plane 3 of the eight (or its stand-in): at most four depth steps from every plane of the pass.
            // Falls through to the four-plane pass below whenever it cannot take the two chunks (uniform conditions).
            if constexpr(PLANES == 8)
            {
                if((c & 1u) == 0u && c + 1u < kSgmChunksPerWg && zc >= zBegin && zc + 8u <= zEnd) // uniform: eight planes, all in range
                {
                    QuadPlane q[8];
                    f3 ra[8];
                    bool vv[8], ll[8];
#pragma unroll
                    for(int k = 0; k < 8; ++k)
                        plane_q(zc + (unsigned)k, q[k], ra[k], vv[k], ll[k]);
                    bool allLds = true, anyValid = false;
#pragma unroll
                    for(int k = 0; k < 8; ++k)
                    {
                        allLds = allLds && ll[k];
                        anyValid = anyValid || vv[k];
                    }
                    if(allLds) // wave-uniform
                    {
                        uint8_t* const pbB = pb + 4;
                        uint8_t* const psB = ps + 4;
                        unsigned wbB = 0, wsB = 0;
                        if(inRoi)
                        {
                            wbB = *reinterpret_cast<const unsigned*>(pbB);
                            wsB = *reinterpret_cast<const unsigned*>(psB);
                        }
                        float sim[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        LEANSTAT(0);
                        if(anyValid)
                        {
                            auto selP8 = [](bool cnd, const QuadPlane& a, const QuadPlane& b) __attribute__((always_inline)) -> QuadPlane {
                                QuadPlane r;
                                r.t = cnd ? a.t : b.t;
                                r.tax = sel3(cnd, a.tax, b.tax);
                                r.c = sel4(cnd, a.c, b.c);
                                return r;
                            };
                            // the reference plane of the R side: the first valid one in the order 3, 4, 2, 5, 1, 6, 0, 7
                            QuadPlane qf = q[7];
                            f3 raf = ra[7];
                            constexpr int order[7] = {0, 6, 1, 5, 2, 4, 3};
#pragma unroll
                            for(int i = 0; i < 7; ++i)
                            {
                                qf = selP8(vv[order[i]], q[order[i]], qf);
                                raf = sel3(vv[order[i]], ra[order[i]], raf);
                            }
                            QuadPlane qq[8];
#pragma unroll
                            for(int k = 0; k < 8; ++k)
                                qq[k] = selP8(vv[k], q[k], qf);
                            const float tw = qf.t * RK.hrW;
                            ncc_accumulate_lds_fixed8_multi<WSH, false, PAIRED, RP, REC12, 4>(
                              M3x3mulV3(rc.P, raf), Br * qf.t, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z}, qq, Bt, RK.htB, RK.htA, A, tab,
                              make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wc.pitch, Wc.x0, Wc.y0, PAIRED ? 16 : (REC12 ? 12 : 8)), rcCenter, sim);
                        }
                        auto commit2 = [&](unsigned& wbx, unsigned& wsx, int k, float fsim) __attribute__((always_inline)) {
                            const unsigned sh8 = 8u * k;
                            const unsigned b1 = (wbx >> sh8) & 0xffu, b2 = (wsx >> sh8) & 0xffu;
                            if(fsim < (float)b1)
                            {
                                wsx = (wsx & ~(0xffu << sh8)) | (b1 << sh8);
                                wbx = (wbx & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
                            }
                            else if(fsim < (float)b2)
                                wsx = (wsx & ~(0xffu << sh8)) | ((unsigned)fsim << sh8);
                        };
#pragma unroll
                        for(int k = 0; k < 4; ++k)
                        {
                            commit2(wb, ws, k, vv[k] ? to_fsim(sim[k]) : 255.0f);
                            commit2(wbB, wsB, k, vv[k + 4] ? to_fsim(sim[k + 4]) : 255.0f);
                        }
                        if(inRoi)
                        {
                            *reinterpret_cast<unsigned*>(pb) = wb;
                            *reinterpret_cast<unsigned*>(ps) = ws;
                            *reinterpret_cast<unsigned*>(pbB) = wbB;
                            *reinterpret_cast<unsigned*>(psB) = wsB;
                        }
                        ++c; // the next chunk is done too
                        continue;
                    }
                }
            }
            QuadPlane q0, q1, q2, q3;
            f3 ra0, ra1, ra2, ra3;
            bool v0, v1b, v2b, v3, l0, l1, l2, l3;
            plane_q(zc, q0, ra0, v0, l0);
            plane_q(zc + 1u, q1, ra1, v1b, l1);
            plane_q(zc + 2u, q2, ra2, v2b, l2);
            plane_q(zc + 3u, q3, ra3, v3, l3);
            if(l0 && l1 && l2 && l3) // wave-uniform
            {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                LEANSTAT(1);
                if(v0 || v1b || v2b || v3)
                {
                    // a lane's invalid planes run a copy of one of its valid planes (their results are not committed); the R side comes
                    // from plane 1 of the chunk (or its stand-in): at most two depth steps from every plane of the pass
                    auto selP = [](bool c, const QuadPlane& a, const QuadPlane& b) __attribute__((always_inline)) -> QuadPlane {
                        QuadPlane r;
                        r.t = c ? a.t : b.t;
                        r.tax = sel3(c, a.tax, b.tax);
                        r.c = sel4(c, a.c, b.c);
                        return r;
                    };
                    const QuadPlane qf = selP(v1b, q1, selP(v2b, q2, selP(v0, q0, q3)));
                    const f3 raf = sel3(v1b, ra1, sel3(v2b, ra2, sel3(v0, ra0, ra3)));
                    const float tw = qf.t * RK.hrW;
                    ncc_accumulate_lds_fixed8_quad<WSH, false, PAIRED, RP, REC12>(M3x3mulV3(rc.P, raf), Br * qf.t, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z},
                                                                    selP(v0, q0, qf), selP(v1b, q1, qf), selP(v2b, q2, qf), selP(v3, q3, qf), Bt, RK.htB, RK.htA, A, tab,
                                                                    make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wc.pitch, Wc.x0, Wc.y0, PAIRED ? 16 : (REC12 ? 12 : 8)), rcCenter,
                                                                    s0, s1, s2, s3);
                }
                // (a plane outside the range is not committed at all: its bytes belong to other T cameras)
                if(zc >= zBegin && zc < zEnd)
                    commit(0, v0 ? to_fsim(s0) : 255.0f);
                if(zc + 1u >= zBegin && zc + 1u < zEnd)
                    commit(1, v1b ? to_fsim(s1) : 255.0f);
                if(zc + 2u >= zBegin && zc + 2u < zEnd)
                    commit(2, v2b ? to_fsim(s2) : 255.0f);
                if(zc + 3u >= zBegin && zc + 3u < zEnd)
                    commit(3, v3 ? to_fsim(s3) : 255.0f);
                quadDone = true;
            }
        }
    }
#pragma unroll 1
    for(int k0 = 0; k0 < 4 && !quadDone; k0 += 2)
    {
#pragma unroll 1
    for(int k = k0; k < k0 + 2; ++k)
    {
        const unsigned vz = zc + k;
        if(vz < zBegin || vz >= zEnd) // uniform
            continue;

        bool valid = rValid;
        PatchProj Q;
        float4 tcCenter = make_float4(0.f, 0.f, 0.f, 0.f);
        float tpx = 0.f, tpy = 0.f;
        if(valid)
            valid = plane_geometry(vz, Q, tpx, tpy);
        if(valid)
        {
            tcCenter = tex_bilinear_px<FIXED8>(A.tcL, fmaf(tpx, A.tcSx, A.tcOx), fmaf(tpy, A.tcSy, A.tcOy));
            valid = !(tcCenter.w < (255.f * 0.4f));
        }
        TWindow Wd = Wc;
        bool laneLds = true; // my taps of this plane lie inside the staged window
        if(chunkWin)
        {
            if(valid)
            {
                const float cxT = fmaf(tpx, A.tcSx, A.tcOx), cyT = fmaf(tpy, A.tcSy, A.tcOy);
                // (a lane that was valid on neither extreme plane has no extent on record: it never reads the window)
                laneLds = lanePart && (cxT - extX - 1.0f >= (float)Wd.x0) && (cxT + extX + 2.0f <= (float)(Wd.x0 + Wd.w - 1)) &&
                          (cyT - extY - 1.0f >= (float)Wd.y0) && (cyT + extY + 2.0f <= (float)(Wd.y0 + Wd.h - 1));
            }
            laneLds = __ballot(valid && !laneLds) == 0ull; // wave-uniform choice of the tap source (see the Refine kernel)
        }
        else
        {
            float bx0 = 0.f, by0 = 0.f, bx1 = 0.f, by1 = 0.f;
            bool rInside = false;
            if(valid)
                corner_boxes(Q, A, wsh, R, bx0, by0, bx1, by1, rInside);
            publish_box(sh, k, valid, bx0, by0, bx1, by1, rInside);
            __syncthreads();
            Wd = stage_t_window(sT, sh, k, A, R.ok, paired, halfPaired, LEAN, rec12);
            __syncthreads();
        }

        float fsim = 255.0f;
        if(LEAN && __any(valid))
        {
            if(Wd.ok && laneLds)
                LEANSTAT(2);
            else
                LEANSTAT(3);
        }
        if(valid)
        {
            float s;
            if(Wd.ok && laneLds && FIXED8 && !noPacked)
                s = ncc_accumulate_lds_fixed8<WSH, false, PAIRED, RP, REC12>(Q, A, tab, make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wd.pitch, Wd.x0, Wd.y0, PAIRED ? 16 : (REC12 ? 12 : 8)), rcCenter,
                                                          tcCenter);
            else if(Wd.ok && laneLds)
                s = ncc_accumulate<FIXED8, WSH, false>(Q, A, tab, LdsTap{sR, R.pitch, R.x0, R.y0}, LdsTap{sT, Wd.pitch, Wd.x0, Wd.y0}, rcCenter, tcCenter);
            else
                s = ncc_accumulate<FIXED8, WSH, false>(Q, A, tab, GlobalTap{A.rcL}, GlobalTap{A.tcL}, rcCenter, tcCenter);
            fsim = to_fsim(s);
        }
        commit(k, fsim);
    }
    }
    if(inRoi)
    {
        *reinterpret_cast<unsigned*>(pb) = wb;
        *reinterpret_cast<unsigned*>(ps) = ws;
    }
    }
}

// ---------------------------------------------------------------------------------------------
// Refine similarity: fp16 volume += sigmoid-filtered NCC, 8 planes per lane per launch-z
// ---------------------------------------------------------------------------------------------
template <bool FIXED8, int WSH, bool PAIRED, int RP = 0, int PLANES = 1>
__global__ void __launch_bounds__(256, AVDM_SIM_WAVES_PER_SIMD)
  refine_similarity_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize,
                           int map_pitch, const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, NccArgs A,
                           PatchTable tab, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi, unsigned* __restrict__ outliers, unsigned listCap)
{
    // The OUTLIER LIST (listCap > 0; round 5).  A pixel whose SGM depth is wrong projects its patch tens of texels away from its
    // neighbours' in a wide-baseline T camera: it cannot take its taps from the workgroup's T window, and until round 4 its whole WAVE then ran
    // the chunk one plane per pass with every tap from global memory (~4 x the instructions per plane-sample, for 64 lanes, because of one) —
    // 30 % more sweep time on the outer cameras of the bench.  Now such a LANE is taken out of the pass (all of its planes invalid: the lane is
    // masked off inside the pass, it reads nothing) and appended to a list of (pixel, first plane, number of planes) units in `outliers` —
    // {count, pad, entries ...} — that refine_outlier_kernel works off after this kernel with one lane per unit; the other 63 lanes keep the
    // eight-plane pass.  A full list (count >= listCap) leaves the wave on the old path.
    const bool listing = PLANES >= 4 && listCap != 0u;
    extern __shared__ __attribute__((aligned(16))) uint2 smem[];
    uint2* sR = smem;
    uint2* sT = smem + A.rcap;
    __shared__ BlockShared sh;
    const int wsh = WSH > 0 ? WSH : A.wsh;

    int tx, ty;
    pixel_of_lane(tx, ty);
    const unsigned vx = blockIdx.x * 16 + tx, vy = blockIdx.y * 16 + ty;
    const bool inRoi = vx < roi.x.end - roi.x.begin && vy < roi.y.end - roi.y.begin;
    // kRefineChunksPerWg chunks of 8 planes per workgroup: one R tile, one T window and one pixel set-up for 16 planes
    const unsigned z0 = ((zBegin >> 3) + blockIdx.z * kRefineChunksPerWg) << 3;
    constexpr unsigned kPlanesPerWg = 8u * kRefineChunksPerWg;

    // LEAN = the default instantiations (RP > 0): the A/B switches of NccArgs are compile-time constants at their defaults there (the host
    // only launches them in that state), which removes the plain-arithmetic LDS path and the counters from the hot kernels
    constexpr bool LEAN = RP > 0;
    const bool noPacked = LEAN ? false : (A.noPacked != 0);
    const bool useChunkWindow = LEAN ? true : (A.chunkWindow != 0);
    const bool usePlanePairs = LEAN ? true : (A.planePairs != 0);
    // paired LDS records only feed the packed FIXED8 path (uniform)
    const bool paired = PAIRED && FIXED8 && !noPacked;
    const bool halfPaired = !PAIRED && FIXED8 && !noPacked; // the packed path without room for 16-byte records
    init_shared(sh);
    const RTile R = stage_r_tile(sR, A, wsh, stepXY, roi, paired, halfPaired, 16, LEAN);
    __syncthreads();

    float2 dps = make_float2(-1.f, 0.f);
    if(inRoi)
        dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    const bool pixActive = inRoi && dps.x > 0.0f; // kernels.cuh:266-270: pixels without an SGM depth keep their volume entries

    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const f3 C = ld3(rc.C);
    const f3 rpv = normalize(M3x3mulV2(rc.iP, x, y));
    const f3 pMid = C + rpv * dps.x;
    // move3DPointByRcPixSize direction: normalize(p - C) recomputed from the mid point like the reference (kernels.cuh:17-24)
    const f3 dir = normalize(pMid - C);

    const float dd = (float)wsh + 2.0f;
    bool rValid = pixActive && !((x < dd) || (x > A.rcW1 - dd) || (y < dd) || (y > A.rcH1 - dd));
    // knife-edge rows / columns (my pixel lies EXACTLY wsh + 2 from an image border): the reference's own border test, per plane (lit::)
    // — evaluated once for the planes of this workgroup, into one bit per plane, ahead of the plane loops (a wave without such a lane skips it)
    unsigned knifeMask = 0xffffffffu;
    {
        const bool knife = rValid && (x == dd || x == A.rcW1 - dd || y == dd || y == A.rcH1 - dd);
        if(AVDM_KNIFE_LITERAL && __any(knife)) // wave-uniform
        {
            if(knife)
            {
                knifeMask = 0u;
#pragma unroll 1
                for(unsigned k = 0; k < 8u * kRefineChunksPerWg; ++k)
                {
                    const unsigned vz = z0 + k;
                    if(vz >= zBegin && vz < zEnd && lit::refine_r_inside(rc, x, y, dps.x, dps.y, (int)vz - ((volDimZ - 1) / 2), dd, A.rcW1, A.rcH1))
                        knifeMask |= 1u << k;
                }
            }
        }
    }
    auto r_border_ok = [&](unsigned vz) __attribute__((always_inline)) -> bool {
        const unsigned k = vz - z0;
        return k >= 32u || ((knifeMask >> k) & 1u) != 0u;
    };
    float4 rcCenter = make_float4(0.f, 0.f, 0.f, 0.f);
    if(rValid)
    {
        rcCenter = tex_bilinear_px<FIXED8>(A.rcL, fmaf(x, A.rcSx, A.rcOx), fmaf(y, A.rcSy, A.rcOy));
        rValid = !(rcCenter.w < (255.f * 0.9f));
    }

    __half* const pv0 = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2 + z0;

    const RayConsts RK = make_ray_consts(rc, tc, C, dir, x, y);
    // geometry of my patch on plane vz: the 3-D point, the patch axes and their projections; false when the patch centre fails the border
    // test in T (Patch.cuh:486-496).  (tpx, tpy) = the centre's pixel in T.
    auto plane_geometry = [&](unsigned vz, PatchProj& Q, float& tpx, float& tpy) __attribute__((always_inline)) -> bool {
        const int rel = (int)vz - ((volDimZ - 1) / 2);
        // p = pMid + dir * (rel * pixSize) (move3DPointByRcPixSize) = C + dir * t: on my pixel's ray
        const float t = fmaf((float)rel, dps.y, dps.x);
        const f3 p = C + dir * t;
        const float pd = RK.pixK * t;
        f3 ax, ay;
        {
            const f3 v1 = f3{-dir.x, -dir.y, -dir.z}; // normalize(C - p)
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
        const float tw = t * RK.hrW;
        Q = make_patch_proj_on_ray(rc, tc, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z}, fma3(t, RK.htB, RK.htA), ax, ay, pd);
        const float it0 = proj_rcp(Q.ht0.z);
        tpx = Q.ht0.x * it0;
        tpy = Q.ht0.y * it0;
        return r_border_ok(vz) && !((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd));
    };

    // ---- ONE T window for all planes of the chunk (paired records) -----------------------------------------------------------------
    // The planes of a Refine chunk are one pixel size apart along the ray: the projected patch moves by about a texel per plane, so the
    // union of the windows of 8 planes is barely larger than one of them — and building a window is the expensive part of a plane outside
    // the sample loop (four corner projections and four wave reductions for the box, a cooperative copy, two barriers, and the centre colour
    // fetched from global memory before the box is even known).  Each lane projects its patch corners on the FIRST and the LAST plane of
    // the chunk; a point moving along a ray projects to a monotone path in T, so the hull of the two extreme quads bounds the planes in
    // between (one more texel of slack, and every lane re-checks its own centre against the staged window per plane: a lane that does
    // not fit takes the global-memory taps for that plane).  The alpha test of the centre no longer decides who contributes to the box
    // (alpha is only known once the window is staged: the centre is read from it).
    constexpr bool CHUNK_CAPABLE = PAIRED && FIXED8;
    bool chunkWin = false;
    TWindow Wc;
    Wc.ok = false;
    Wc.tooLarge = false;
    Wc.x0 = Wc.y0 = Wc.w = Wc.h = Wc.pitch = 0;
    float extX = 0.f, extY = 0.f; // half extent of my projected patch in T texels (the larger of the two extreme planes)
    bool lanePart = false;        // my patch was part of the hull (valid on the first or the last plane of the chunk)
    if(CHUNK_CAPABLE && paired && useChunkWindow)
    {
        const unsigned ka = z0 > zBegin ? z0 : zBegin, kbEnd = (z0 + kPlanesPerWg < zEnd) ? z0 + kPlanesPerWg : zEnd;
        if(ka < kbEnd) // uniform
        {
            float bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
            bool part = false, rIn = true;
#pragma unroll 1
            for(int e = 0; e < 2; ++e)
            {
                const unsigned vz = e == 0 ? ka : kbEnd - 1u;
                if(e == 1 && vz == ka)
                    break;
                PatchProj Q;
                float tpx, tpy;
                if(rValid && plane_geometry(vz, Q, tpx, tpy))
                {
                    float cx0, cy0, cx1, cy1;
                    bool ri;
                    corner_boxes(Q, A, wsh, R, cx0, cy0, cx1, cy1, ri);
                    bx0 = fminf(bx0, cx0);
                    by0 = fminf(by0, cy0);
                    bx1 = fmaxf(bx1, cx1);
                    by1 = fmaxf(by1, cy1);
                    extX = fmaxf(extX, 0.5f * (cx1 - cx0));
                    extY = fmaxf(extY, 0.5f * (cy1 - cy0));
                    rIn = rIn && ri;
                    part = true;
                }
            }
            publish_box(sh, AVDM_CHUNK_BOX, part, bx0 - 1.0f, by0 - 1.0f, bx1 + 1.0f, by1 + 1.0f, rIn);
            __syncthreads();
            Wc = stage_t_window(sT, sh, AVDM_CHUNK_BOX, A, R.ok, paired, halfPaired, LEAN, false, LEAN ? 20 : -1);
            __syncthreads();
            if(AVDM_REFINE_ANCHORED_WINDOW && listing && !Wc.ok && R.ok) // uniform
            {
                // The hull of ALL lanes is no window (it exceeds the LDS budget, leaves the image, or some lane's R taps leave the R tile) and there
                // is an outlier list: an ANCHORED window instead — as large as the budget allows, centred on where most of the workgroup's patches
                // land (the mean of the lanes' hull centres, taken once more over the lanes within 24 texels of it: a depth edge or a patch of wrong
                // SGM depths does not drag it into the gap between two clusters), clipped to the image.  Whoever fits takes the eight-plane pass
                // from it; whoever does not — plane by plane, plane_q — goes to the list.  The workgroup never falls back as a whole.
                const float hcx = 0.5f * (bx0 + bx1), hcy = 0.5f * (by0 + by1);
                float mx = 0.f, my = 0.f, nn = 0.f;
                const unsigned wv = threadIdx.x >> 6;
#pragma unroll 1
                for(int pass = 0; pass < 2; ++pass)
                {
                    const bool in = part && (pass == 0 || (fabsf(hcx - mx) <= 24.0f && fabsf(hcy - my) <= 24.0f));
                    const float n = wave_sum_f32(in ? 1.0f : 0.0f), sx = wave_sum_f32(in ? hcx : 0.0f), sy = wave_sum_f32(in ? hcy : 0.0f);
                    if((threadIdx.x & 63u) == 0u) // the per-plane boxes are idle while a chunk window is being sought: four of them carry the sums
                    {
                        sh.box[wv][0] = __float_as_int(n);
                        sh.box[wv][1] = __float_as_int(sx);
                        sh.box[wv][2] = __float_as_int(sy);
                    }
                    __syncthreads();
                    float N = 0.f, SX = 0.f, SY = 0.f;
#pragma unroll
                    for(int w = 0; w < 4; ++w)
                    {
                        N += __int_as_float(sh.box[w][0]);
                        SX += __int_as_float(sh.box[w][1]);
                        SY += __int_as_float(sh.box[w][2]);
                    }
                    __syncthreads();
                    if(N > 0.0f) // uniform (pass 1 with nobody near the mean keeps the mean of pass 0)
                    {
                        nn = N;
                        mx = SX / N;
                        my = SY / N;
                    }
                }
                init_shared(sh);
                __syncthreads();
                if(LEAN)
                    LEANSTAT_WG(12);
                if(nn > 0.0f) // uniform
                {
                    if(threadIdx.x == 0)
                    {
                        // box extent E (stage_t_window adds one texel before and two behind): the largest even E whose window fits the T budget
                        int E = 64;
                        while(E > 8 && lds_units(lds_pitch_for(E + 4) * (E + 4), paired, false) > A.tcap)
                            E -= 2;
                        const int cx = (int)floorf(mx), cy = (int)floorf(my);
                        const int x0b = min(max(cx - E / 2, 0), max(A.tcL.W - 2 - E, 0)), y0b = min(max(cy - E / 2, 0), max(A.tcL.H - 2 - E, 0));
                        sh.box[AVDM_CHUNK_BOX][0] = x0b;
                        sh.box[AVDM_CHUNK_BOX][1] = y0b;
                        sh.box[AVDM_CHUNK_BOX][2] = min(x0b + E, A.tcL.W - 2);
                        sh.box[AVDM_CHUNK_BOX][3] = min(y0b + E, A.tcL.H - 2);
                    }
                    __syncthreads();
                    Wc = stage_t_window(sT, sh, AVDM_CHUNK_BOX, A, R.ok, paired, halfPaired, LEAN);
                    __syncthreads();
                    part = part && rIn; // a lane whose R taps leave the staged R tile is an outlier too
                }
            }
            else if(Wc.tooLarge) // uniform
            {
                // the hull of ALL lanes does not fit: once more without the outliers (wave_inlier) — their waves then take the global-memory
                // taps on the planes they are valid on, the rest of the workgroup keeps the LDS path
                if(threadIdx.x < 4)
                    sh.box[AVDM_CHUNK_BOX][threadIdx.x] = (threadIdx.x & 2) ? INT_MIN : INT_MAX;
                __syncthreads();
                part = wave_inlier(part, 0.5f * (bx0 + bx1), 0.5f * (by0 + by1), 16.0f * (float)stepXY + 16.0f);
                publish_box(sh, AVDM_CHUNK_BOX, part, bx0 - 1.0f, by0 - 1.0f, bx1 + 1.0f, by1 + 1.0f, rIn);
                __syncthreads();
                Wc = stage_t_window(sT, sh, AVDM_CHUNK_BOX, A, R.ok, paired, halfPaired, LEAN);
                __syncthreads();
            }
            chunkWin = Wc.ok;
            lanePart = part;
        }
    }
    if(LEAN)
    {
        LEANSTAT_WG(13);
        if(!chunkWin)
            LEANSTAT_WG(14);
    }

#pragma unroll 1
    for(unsigned c = 0; c < kRefineChunksPerWg; ++c)
    {
    const unsigned zc = z0 + 8u * c;
    if(zc >= zEnd) // uniform
        break;
    __half* const pv = pv0 + 8u * c;
    uint4 packed = make_uint4(0u, 0u, 0u, 0u);
    if(pixActive)
        packed = *reinterpret_cast<const uint4*>(pv);
    if(!chunkWin && c > 0)
    { // the per-plane fall-back reuses the box entries of the previous chunk
        __syncthreads();
        init_shared(sh);
        __syncthreads();
    }
    // packed[k] += s without indexing the register quad dynamically (that would spill it to scratch)
    auto commit = [&](int k, float s) __attribute__((always_inline)) {
        const unsigned sel = (unsigned)k >> 1, hiHalf = (unsigned)k & 1u;
        unsigned word = sel == 0 ? packed.x : (sel == 1 ? packed.y : (sel == 2 ? packed.z : packed.w));
        const unsigned short hbits = (unsigned short)(hiHalf ? (word >> 16) : (word & 0xffffu));
        const __half hs = __float2half(__half2float(__ushort_as_half(hbits)) + s);
        const unsigned nb = (unsigned)__half_as_ushort(hs);
        word = hiHalf ? ((word & 0x0000ffffu) | (nb << 16)) : ((word & 0xffff0000u) | nb);
        packed.x = sel == 0 ? word : packed.x;
        packed.y = sel == 1 ? word : packed.y;
        packed.z = sel == 2 ? word : packed.z;
        packed.w = sel == 3 ? word : packed.w;
    };
    unsigned quadsDone = 0u; // bit q: planes 4 q ... 4 q + 3 of the chunk went through the four-plane pass
    if constexpr(PLANES >= 4 && CHUNK_CAPABLE)
    {
        if(chunkWin && usePlanePairs && !noPacked)
        {
            // four planes per pass over the patch (ncc_accumulate_lds_fixed8_quad), the planes' geometry in the form the pixel ray gives it:
            // ay — the normal of the epipolar plane of my ray — once, M_T (ax d) per plane
            const f3 v1 = f3{-dir.x, -dir.y, -dir.z};
            const f3 ay = normalize(cross(v1, ld3(tc.C) - C));
            const f3 Bt = M3x3mulV3(tc.P, ay) * RK.pixK, Br = M3x3mulV3(rc.P, ay) * RK.pixK;
            auto plane_q = [&](unsigned vz, QuadPlane& q, f3& raxOut, bool& valid, bool& laneLds, bool& misfit) __attribute__((always_inline)) {
                const bool inRange = vz >= zBegin && vz < zEnd; // uniform; a plane outside the range is an invalid plane of the pass
                valid = rValid && inRange;
                q.c = make_float4(0.f, 0.f, 0.f, 0.f);
                const int rel = (int)vz - ((volDimZ - 1) / 2);
                const float t = fmaf((float)rel, dps.y, dps.x);
                q.t = t;
                const f3 p = C + dir * t;
                const f3 v2 = normalize(ld3(tc.C) - p);
                f3 nn;
                if(sgmNormal != nullptr)
                {
                    const float* np_ = (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx;
                    nn = f3{np_[0], np_[1], np_[2]};
                }
                else
                    nn = normalize((v1 + v2) * 0.5f);
                const f3 axd = normalize(cross(ay, nn)) * (RK.pixK * t);
                q.tax = M3x3mulV3(tc.P, axd);
                raxOut = axd; // (projected into R only for the ONE plane that provides the R side of the pass: M3x3mulV3(rc.P, .) at the call)
                const f3 ht0 = fma3(t, RK.htB, RK.htA);
                const float it0 = proj_rcp(ht0.z);
                const float tpx = ht0.x * it0, tpy = ht0.y * it0;
                valid = valid && r_border_ok(vz) && !((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd));
                laneLds = true;
                misfit = false;
                if(valid)
                {
                    const float cxT = fmaf(tpx, A.tcSx, A.tcOx), cyT = fmaf(tpy, A.tcSy, A.tcOy);
                    laneLds = lanePart && (cxT - extX - 1.0f >= (float)Wc.x0) && (cxT + extX + 2.0f <= (float)(Wc.x0 + Wc.w - 1)) &&
                              (cyT - extY - 1.0f >= (float)Wc.y0) && (cyT + extY + 2.0f <= (float)(Wc.y0 + Wc.h - 1));
                    if(laneLds)
                        q.c = lds_center_paired(sT, Wc.pitch, Wc.x0, Wc.y0, cxT, cyT);
                    else if(!listing)
                        q.c = tex_bilinear_px<FIXED8>(A.tcL, cxT, cyT);
                    // (listing: a plane of mine outside the window sends me to the outlier list, where every plane is evaluated from scratch —
                    // its centre colour is not needed here)
                    if(laneLds || !listing)
                        valid = !(q.c.w < (255.f * 0.4f));
                    misfit = !laneLds;
                }
                laneLds = __ballot(valid && !laneLds) == 0ull; // wave-uniform choice of the tap source
            };
            // append the lanes of `outl` as units (my pixel, planes zFirst ... zFirst + nPlanes - 1) to the outlier list; wave-uniform result:
            // false = the list is full (the wave stays on the old path; slots it was granted below the capacity are written as empty units)
            auto list_outliers = [&](bool outl, unsigned zFirst, unsigned nPlanes) __attribute__((always_inline)) -> bool {
                const unsigned long long m = __ballot(outl);
                const unsigned n = (unsigned)__popcll(m);
                // a list that is already full refuses without another atomic (every later wave with a misfit lane used to pay up to three
                // failed global atomics per chunk before it dropped to the one-plane path — in exactly the many-outliers case that is slowest anyway)
                if(__builtin_nontemporal_load(outliers) >= listCap) // uniform
                    return false;
                unsigned base = 0u;
                if((threadIdx.x & 63u) == 0u)
                    base = atomicAdd(outliers, n);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                const bool granted = base + n <= listCap; // uniform
                if(outl)
                {
                    const unsigned idx = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if(idx < listCap)
                        reinterpret_cast<uint2*>(outliers)[1u + idx] = make_uint2(vx | (vy << 16), granted ? (zFirst | (nPlanes << 16)) : 0u);
                }
                return granted;
            };
            auto selP = [](bool c, const QuadPlane& a, const QuadPlane& b) __attribute__((always_inline)) -> QuadPlane {
                QuadPlane r;
                r.t = c ? a.t : b.t;
                r.tax = sel3(c, a.tax, b.tax);
                r.c = sel4(c, a.c, b.c);
                return r;
            };
            // EIGHT planes per pass (AVDM_REFINE_PLANES8=1, experimental): the whole chunk when all of it lies in the T camera's range (24 of the 31
            // planes of the default sweep); the R side comes from plane 3 of the eight (or its stand-in).  Falls through to the two four-plane
            // passes otherwise.  34 instead of 46 VALU instructions per plane and sample in the loop; measured (sessions r04_p, q): the Refine
            // sweep 267.2 against 276.4 ms, volumes within the fp16 quantum of the default's on all but 2e-5 of the entries.
            if constexpr(PLANES == 8)
            {
                // (AVDM_REFINE_OCTO_PARTIAL, the default since round 5: a chunk that only overlaps the range too — its planes outside are invalid planes
                // of the pass, like those of a four-plane pass; the 31 planes of the default sweep are then four passes of eight)
                if(AVDM_REFINE_OCTO_PARTIAL ? (zc + 8u > zBegin) : (zc >= zBegin && zc + 8u <= zEnd)) // uniform
                {
                    QuadPlane q[8];
                    f3 ra[8];
                    bool vv[8], ll[8], mf[8];
#pragma unroll
                    for(int k = 0; k < 8; ++k)
                        plane_q(zc + (unsigned)k, q[k], ra[k], vv[k], ll[k], mf[k]);
                    bool allLds = true, anyValid = false, outl = false;
#pragma unroll
                    for(int k = 0; k < 8; ++k)
                    {
                        allLds = allLds && ll[k];
                        outl = outl || (vv[k] && mf[k]);
                    }
                    if(listing && !allLds && list_outliers(outl, zc, 8u)) // wave-uniform
                    {
                        // my chunk is on the list: I take no part in the pass; the rest of the wave runs it from the window
#pragma unroll
                        for(int k = 0; k < 8; ++k)
                            vv[k] = vv[k] && !outl;
                        allLds = true;
                    }
#pragma unroll
                    for(int k = 0; k < 8; ++k)
                        anyValid = anyValid || vv[k];
                    if(allLds) // wave-uniform
                    {
                        LEANSTAT(8);
                        if(anyValid)
                        {
                            QuadPlane qf = q[7];
                            f3 raf = ra[7];
                            constexpr int order[7] = {0, 6, 1, 5, 2, 4, 3}; // the last one applied wins: 3, 4, 2, 5, 1, 6, 0, 7
#pragma unroll
                            for(int i = 0; i < 7; ++i)
                            {
                                qf = selP(vv[order[i]], q[order[i]], qf);
                                raf = sel3(vv[order[i]], ra[order[i]], raf);
                            }
                            QuadPlane qq[8];
#pragma unroll
                            for(int k = 0; k < 8; ++k)
                                qq[k] = selP(vv[k], q[k], qf);
                            float sim[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            const float tw = qf.t * RK.hrW;
                            ncc_accumulate_lds_fixed8_multi<WSH, true, PAIRED, RP, false, 4>(
                              M3x3mulV3(rc.P, raf), Br * qf.t, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z}, qq, Bt, RK.htB, RK.htA, A, tab,
                              make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wc.pitch, Wc.x0, Wc.y0, PAIRED ? 16 : 8), rcCenter, sim);
                            if(vv[0]) commit(0, sim[0]);
                            if(vv[1]) commit(1, sim[1]);
                            if(vv[2]) commit(2, sim[2]);
                            if(vv[3]) commit(3, sim[3]);
                            if(vv[4]) commit(4, sim[4]);
                            if(vv[5]) commit(5, sim[5]);
                            if(vv[6]) commit(6, sim[6]);
                            if(vv[7]) commit(7, sim[7]);
                        }
                        quadsDone = 3u;
                    }
                }
            }
#pragma unroll 1
            for(unsigned qd = 0; qd < 2u; ++qd)
            {
                if constexpr(PLANES == 8)
                    if((quadsDone >> qd) & 1u) // uniform per wave: the eight-plane pass took the chunk
                        continue;
                const unsigned zq = zc + 4u * qd;
                if(zq >= zEnd || zq + 4u <= zBegin) // uniform: nothing of this quad is in range
                {
                    quadsDone |= 1u << qd;
                    continue;
                }
                QuadPlane q0, q1, q2, q3;
                f3 ra0, ra1, ra2, ra3;
                bool v0, v1b, v2b, v3, l0, l1, l2, l3, m0, m1, m2, m3;
                plane_q(zq, q0, ra0, v0, l0, m0);
                plane_q(zq + 1u, q1, ra1, v1b, l1, m1);
                plane_q(zq + 2u, q2, ra2, v2b, l2, m2);
                plane_q(zq + 3u, q3, ra3, v3, l3, m3);
                bool quadLds = l0 && l1 && l2 && l3;
                if(listing && !quadLds)
                {
                    const bool outl = (v0 && m0) || (v1b && m1) || (v2b && m2) || (v3 && m3);
                    if(list_outliers(outl, zq, 4u)) // wave-uniform: the lanes outside the window are on the list, the others run the pass
                    {
                        v0 = v0 && !outl, v1b = v1b && !outl, v2b = v2b && !outl, v3 = v3 && !outl;
                        quadLds = true;
                    }
                }
                if(!quadLds) // wave-uniform: this wave runs the quad one plane per pass
                    continue;
                LEANSTAT(9);
                if(v0 || v1b || v2b || v3)
                {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    // a lane's invalid planes run a copy of one of its valid planes (their results are not committed); the R side comes from
                    // plane 1 of the quad (or its stand-in): at most two depth steps from every plane of the pass
                    const QuadPlane qf = selP(v1b, q1, selP(v2b, q2, selP(v0, q0, q3)));
                    const f3 raf = sel3(v1b, ra1, sel3(v2b, ra2, sel3(v0, ra0, ra3)));
                    const float tw = qf.t * RK.hrW;
                    ncc_accumulate_lds_fixed8_quad<WSH, true, PAIRED, RP>(M3x3mulV3(rc.P, raf), Br * qf.t, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z},
                                                                          selP(v0, q0, qf), selP(v1b, q1, qf), selP(v2b, q2, qf), selP(v3, q3, qf), Bt, RK.htB, RK.htA, A,
                                                                          tab, make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wc.pitch, Wc.x0, Wc.y0, PAIRED ? 16 : 8),
                                                                          rcCenter, s0, s1, s2, s3);
                    // (packed[k] is indexed with constants in commit(): qd is a loop counter, so both cases are spelled out)
                    if(qd == 0u)
                    {
                        if(v0) commit(0, s0);
                        if(v1b) commit(1, s1);
                        if(v2b) commit(2, s2);
                        if(v3) commit(3, s3);
                    }
                    else
                    {
                        if(v0) commit(4, s0);
                        if(v1b) commit(5, s1);
                        if(v2b) commit(6, s2);
                        if(v3) commit(7, s3);
                    }
                }
                quadsDone |= 1u << qd;
            }
        }
    }
#pragma unroll 1
    for(int k0 = 0; k0 < 8; k0 += 2)
    {
        if((quadsDone >> (k0 >> 2)) & 1u) // uniform per wave
            continue;
#pragma unroll 1
    for(int k = k0; k < k0 + 2; ++k)
    {
        const unsigned vz = zc + k;
        if(vz < zBegin || vz >= zEnd) // uniform
            continue;

        bool valid = rValid;
        PatchProj Q;
        float4 tcCenter = make_float4(0.f, 0.f, 0.f, 0.f);
        float tpx = 0.f, tpy = 0.f;
        if(valid)
            valid = plane_geometry(vz, Q, tpx, tpy);
        TWindow Wd = Wc;
        bool laneLds = true; // my taps of this plane lie inside the staged window
        if(chunkWin)
        {
            if(valid)
            {
                const float cxT = fmaf(tpx, A.tcSx, A.tcOx), cyT = fmaf(tpy, A.tcSy, A.tcOy);
                // taps = texels floor(.) and floor(.) + 1 of positions within +- ext of the centre; one more texel for rounding
                // (a lane that was valid on neither extreme plane has no extent on record: it never reads the window)
                laneLds = lanePart && (cxT - extX - 1.0f >= (float)Wd.x0) && (cxT + extX + 2.0f <= (float)(Wd.x0 + Wd.w - 1)) &&
                          (cyT - extY - 1.0f >= (float)Wd.y0) && (cyT + extY + 2.0f <= (float)(Wd.y0 + Wd.h - 1));
                if(laneLds)
                    tcCenter = lds_center_paired(sT, Wd.pitch, Wd.x0, Wd.y0, cxT, cyT);
                else
                    tcCenter = tex_bilinear_px<FIXED8>(A.tcL, cxT, cyT);
                valid = !(tcCenter.w < (255.f * 0.4f));
            }
            // the choice of the tap source stays WAVE-uniform (a scalar branch around the sample loop, as with one window per plane): if one
            // lane does not fit, its whole wave takes the global-memory taps for this plane
            laneLds = __ballot(valid && !laneLds) == 0ull;
        }
        else
        {
            float bx0 = 0.f, by0 = 0.f, bx1 = 0.f, by1 = 0.f;
            bool rInside = false;
            if(valid)
            {
                tcCenter = tex_bilinear_px<FIXED8>(A.tcL, fmaf(tpx, A.tcSx, A.tcOx), fmaf(tpy, A.tcSy, A.tcOy));
                valid = !(tcCenter.w < (255.f * 0.4f));
            }
            if(valid)
                corner_boxes(Q, A, wsh, R, bx0, by0, bx1, by1, rInside);
            publish_box(sh, k, valid, bx0, by0, bx1, by1, rInside);
            __syncthreads();
            Wd = stage_t_window(sT, sh, k, A, R.ok, paired, halfPaired, LEAN);
            __syncthreads();
        }

        if(LEAN && __any(valid))
        {
            if(Wd.ok && laneLds)
                LEANSTAT(10);
            else
                LEANSTAT(11);
        }
        if(valid)
        {
            float s;
            if(Wd.ok && laneLds && FIXED8 && !noPacked)
                s = ncc_accumulate_lds_fixed8<WSH, true, PAIRED, RP>(Q, A, tab, make_windows(smem, A.rcap, R.pitch, R.x0, R.y0, Wd.pitch, Wd.x0, Wd.y0, PAIRED ? 16 : 8), rcCenter,
                                                         tcCenter);
            else if(Wd.ok && laneLds)
                s = ncc_accumulate<FIXED8, WSH, true>(Q, A, tab, LdsTap{sR, R.pitch, R.x0, R.y0}, LdsTap{sT, Wd.pitch, Wd.x0, Wd.y0}, rcCenter, tcCenter);
            else
                s = ncc_accumulate<FIXED8, WSH, true>(Q, A, tab, GlobalTap{A.rcL}, GlobalTap{A.tcL}, rcCenter, tcCenter);
            commit(k, s);
        }
    }
    }
    if(pixActive)
        *reinterpret_cast<uint4*>(pv) = packed;
    }
}

// ---------------------------------------------------------------------------------------------
// The outlier list of refine_similarity_kernel, worked off: one LANE per unit (pixel, first plane, number of planes), every plane of the unit
// evaluated from scratch exactly as the per-plane fall-back of refine_similarity_kernel evaluates it for a lane whose taps leave the
// workgroup's T window — the pixel's ray, the plane's patch, the border tests (the reference's own on a knife-edge row), both centre colours
// and every tap through the software texture unit from global memory (clamp addressing), one plane per pass — and added to the fp16 volume.
// The units of a launch are scattered pixels (wrong SGM depths): no window would serve 64 of them.  Launched after the sweep kernel on the same
// stream with a fixed grid; the lanes stride over min(count, capacity) units.
// ---------------------------------------------------------------------------------------------
template <bool FIXED8, int WSH>
__global__ void __launch_bounds__(256)
  refine_outlier_kernel(__half* __restrict__ vol, long long pitch_y, int pitch_x, int volDimZ, const float2* __restrict__ sgmDepthPixSize, int map_pitch,
                        const float* __restrict__ sgmNormal, int normal_pitch, avdm_camera_t rc, avdm_camera_t tc, NccArgs A, PatchTable tab, int stepXY,
                        unsigned zBegin, unsigned zEnd, avdm_roi_t roi, const unsigned* __restrict__ list, unsigned listCap, unsigned* __restrict__ totals,
                        unsigned* __restrict__ refused)
{
    const int wsh = WSH > 0 ? WSH : A.wsh;
    const unsigned count = min(list[0], listCap);
    if(refused != nullptr && blockIdx.x == 0u && threadIdx.x == 0u && list[0] > listCap) // (always on: one compare per launch)
        atomicAdd(refused, list[0] - listCap);
    if(totals != nullptr && blockIdx.x == 0u && threadIdx.x == 0u) // AVDM_REFINE_OUTLIER_STATS=1: {units worked off, units that found the list full}
    {
        atomicAdd(&totals[0], count);
        atomicAdd(&totals[1], list[0] - count);
    }
    const uint2* const units = reinterpret_cast<const uint2*>(list) + 1;
    // one lane per PLANE of a unit (round 6; one lane per unit until round 5: eight planes x 49 samples x 8 global taps in sequence per lane, a
    // latency chain that made this kernel 0.3 ... 1.3 ms per launch with the machine a quarter full).  The eight lanes of a unit share its pixel
    // (the same R taps: one cache line) and add into different halfs of the volume.
#pragma unroll 1
    for(unsigned i = blockIdx.x * 256u + threadIdx.x; i < 8u * count; i += gridDim.x * 256u)
    {
        const uint2 u = units[i >> 3];
        const unsigned vx = u.x & 0xffffu, vy = u.x >> 16, zFirst = u.y & 0xffffu, nPlanes = u.y >> 16;
        if((i & 7u) >= nPlanes) // (nPlanes == 0: a slot of a wave that found the list full)
            continue;
        const float2 dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
        if(!(dps.x > 0.0f))
            continue;
        const float x = (float)(roi.x.begin + vx) * (float)stepXY;
        const float y = (float)(roi.y.begin + vy) * (float)stepXY;
        const f3 C = ld3(rc.C);
        const f3 rpv = normalize(M3x3mulV2(rc.iP, x, y));
        const f3 pMid = C + rpv * dps.x;
        const f3 dir = normalize(pMid - C); // kernels.cuh:17-24
        const float dd = (float)wsh + 2.0f;
        if((x < dd) || (x > A.rcW1 - dd) || (y < dd) || (y > A.rcH1 - dd))
            continue;
        const bool knife = AVDM_KNIFE_LITERAL && (x == dd || x == A.rcW1 - dd || y == dd || y == A.rcH1 - dd);
        const float4 rcCenter = tex_bilinear_px<FIXED8>(A.rcL, fmaf(x, A.rcSx, A.rcOx), fmaf(y, A.rcSy, A.rcOy));
        if(rcCenter.w < (255.f * 0.9f))
            continue;
        const RayConsts RK = make_ray_consts(rc, tc, C, dir, x, y);
        __half* const pv = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2;
        const unsigned vz = zFirst + (i & 7u);
        if(vz >= zBegin && vz < zEnd)
        {
            const int rel = (int)vz - ((volDimZ - 1) / 2);
            if(knife && !lit::refine_r_inside(rc, x, y, dps.x, dps.y, rel, dd, A.rcW1, A.rcH1))
                continue;
            // the plane's geometry: refine_similarity_kernel's plane_geometry
            const float t = fmaf((float)rel, dps.y, dps.x);
            const f3 p = C + dir * t;
            const float pd = RK.pixK * t;
            f3 ax, ay;
            {
                const f3 v1 = f3{-dir.x, -dir.y, -dir.z};
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
            const float tw = t * RK.hrW;
            const PatchProj Q = make_patch_proj_on_ray(rc, tc, f3{fmaf(tw, x, RK.hrA.x), fmaf(tw, y, RK.hrA.y), tw + RK.hrA.z}, fma3(t, RK.htB, RK.htA), ax, ay, pd);
            const float it0 = proj_rcp(Q.ht0.z);
            const float tpx = Q.ht0.x * it0, tpy = Q.ht0.y * it0;
            if((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd))
                continue;
            const float4 tcCenter = tex_bilinear_px<FIXED8>(A.tcL, fmaf(tpx, A.tcSx, A.tcOx), fmaf(tpy, A.tcSy, A.tcOy));
            if(tcCenter.w < (255.f * 0.4f))
                continue;
            const float sim = ncc_accumulate<FIXED8, WSH, true, GlobalTap, GlobalTap, AVDM_OUTLIER_UNROLL>(Q, A, tab, GlobalTap{A.rcL}, GlobalTap{A.tcL}, rcCenter, tcCenter);
            pv[vz] = __float2half(__half2float(pv[vz]) + sim);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// useConsistentScale (Patch.cuh:250-308, :499-505): per voxel, the R and T images are sampled at fractional mip levels chosen so that a
// pixel of either image covers the same surface area at the patch centre.  Off by default in the reference (SgmParams.hpp:50,
// RefineParams.hpp:41); built for completeness as one plain kernel for both stages: one lane per pixel, trilinear taps through the
// software texture unit straight from global memory (no LDS windows: the footprint changes with the level).
// ---------------------------------------------------------------------------------------------
struct LodTap
{
    Tex T;
    float lod, invW, invH; // 1 / dims of the nominal level's texel grid: u = (X + 0.5) * invW with X in that level's texel space
    template <bool FIXED8>
    __device__ __forceinline__ float4 fetch(float x, float y) const
    {
        return tex2DLod(T, (x + 0.5f) * invW, (y + 0.5f) * invH, lod);
    }
};

// Patch.cuh:250-308
__device__ __forceinline__ void computeRcTcMipmapLevels(float& out_rcMipmapLevel, float& out_tcMipmapLevel, float mipmapLevel, const avdm_camera_t& rc,
                                                        const avdm_camera_t& tc, float rp0x, float rp0y, float tp0x, float tp0y, f3 p0)
{
    const float rcDepth = size(ld3(rc.C) - p0);
    const float tcDepth = size(ld3(tc.C) - p0);
    const f3 prp1 = ld3(rc.C) + normalize(M3x3mulV2(rc.iP, rp0x + 1.f, rp0y + 0.f)) * rcDepth;
    const f3 ptp1 = ld3(tc.C) + normalize(M3x3mulV2(tc.iP, tp0x + 1.f, tp0y + 0.f)) * tcDepth;
    const float rcDist = size(p0 - prp1);
    const float tcDist = size(p0 - ptp1);
    const float distFactor = rcDist / tcDist;
    if(distFactor < 1.f)
    {
        // T camera has a lower resolution (1 Rc pixSize < 1 Tc pixSize)
        out_tcMipmapLevel = mipmapLevel - log2f(1.f / distFactor);
        if(out_tcMipmapLevel < 0.f)
        {
            out_rcMipmapLevel = mipmapLevel + fabsf(out_tcMipmapLevel);
            out_tcMipmapLevel = 0.f;
        }
    }
    else
    {
        out_rcMipmapLevel = mipmapLevel;
        out_tcMipmapLevel = mipmapLevel + log2f(distFactor);
    }
}

struct CsArgs
{
    Tex rcT, tcT;
    float mipmapLevel;
    int useConsistentScale;
    int useCustomPatchPattern;
    avdm_patch_pattern_t pattern;
};

// compNCCby3DptsYK_customPatchPattern (Patch.cuh:598-773): one weighted NCC per subpart (a full square at a coarser level, or circles of
// samples weighted by colour only), combined with the subparts' weights.  Returns INFINITY when invalid.
template <bool TInvert>
__device__ __forceinline__ float ncc_custom_pattern(const PatchProj& Q, const NccArgs& A, const CsArgs& S, const avdm_camera_t& rc, const avdm_camera_t& tc,
                                                    float rpx, float rpy, float tpx, float tpy, f3 p0)
{
    const float dd = 2.f; // Patch.cuh:617
    if((rpx < dd) || (rpx > A.rcW1 - dd) || (tpx < dd) || (tpx > A.tcW1 - dd) || (rpy < dd) || (rpy > A.rcH1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd))
        return INFINITY;
    LodTap rt{S.rcT, S.mipmapLevel, 1.0f / (float)A.rcL.W, 1.0f / (float)A.rcL.H}, tt{S.tcT, S.mipmapLevel, 1.0f / (float)A.tcL.W, 1.0f / (float)A.tcL.H};
    const float rX0 = fmaf(rpx, A.rcSx, A.rcOx), rY0 = fmaf(rpy, A.rcSy, A.rcOy), tX0 = fmaf(tpx, A.tcSx, A.tcOx), tY0 = fmaf(tpy, A.tcSy, A.tcOy);
    // centre alpha at the stage's level
    if(rt.fetch<true>(rX0, rY0).w < (255.f * 0.9f) || tt.fetch<true>(tX0, tY0).w < (255.f * 0.4f))
        return INFINITY;
    float rcLevel = S.mipmapLevel, tcLevel = S.mipmapLevel;
    if(S.useConsistentScale)
        computeRcTcMipmapLevels(rcLevel, tcLevel, S.mipmapLevel, rc, tc, rpx, rpy, tpx, tpy, p0);

    float fsim = 0.f, wsumParts = 0.f;
#pragma unroll 1
    for(int sp = 0; sp < S.pattern.nbSubparts; ++sp)
    {
        const avdm_patch_pattern_subpart_t& part = S.pattern.subparts[sp];
        rt.lod = rcLevel + part.level;
        tt.lod = tcLevel + part.level;
        const float4 rcCenter = rt.fetch<true>(rX0, rY0), tcCenter = tt.fetch<true>(tX0, tY0);
        float xsum = 0.f, ysum = 0.f, xxsum = 0.f, yysum = 0.f, xysum = 0.f, wsum = 0.f;
        const int n = part.isCircle ? part.nbCoordinates : (2 * part.wsh + 1) * (2 * part.wsh + 1);
#pragma unroll 1
        for(int c = 0; c < n; ++c)
        {
            float fx, fy, dP = 0.f;
            if(part.isCircle)
            {
                fx = part.coordinates[c][0];
                fy = part.coordinates[c][1];
            }
            else
            {
                const int side = 2 * part.wsh + 1;
                const int yp = c / side - part.wsh, xp = c % side - part.wsh;
                fx = (float)xp * part.downscale;
                fy = (float)yp * part.downscale;
                dP = sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP; // CostYKfromLab(dx, dy, ...): the unscaled offsets (Patch.cuh:724)
            }
            f3 hrRow, htRow;
            row_of(Q, fy, hrRow, htRow);
            float rX, rY, tX, tY;
            sample_pos(Q, A, hrRow, htRow, fx, rX, rY, tX, tY);
            const float4 rcC = rt.fetch<true>(rX, rY), tcC = tt.fetch<true>(tX, tY);
            const float drx = rcCenter.x - rcC.x, dry = rcCenter.y - rcC.y, drz = rcCenter.z - rcC.z;
            const float dtx = tcCenter.x - tcC.x, dty = tcCenter.y - tcC.y, dtz = tcCenter.z - tcC.z;
            const float dcr = sqrtf(fmaf(drx, drx, fmaf(dry, dry, drz * drz))), dct = sqrtf(fmaf(dtx, dtx, fmaf(dty, dty, dtz * dtz)));
            const float w = __expf(-(dcr * A.invGammaC + dP)) * __expf(-(dct * A.invGammaC + dP));
            // statistics of L shifted by the centre values, as in ncc_accumulate (shift invariant)
            const float gx = drx, gy = dtx, wgx = w * gx, wgy = w * gy;
            wsum += w;
            xsum += wgx;
            ysum += wgy;
            xxsum = fmaf(wgx, gx, xxsum);
            yysum = fmaf(wgy, gy, yysum);
            xysum = fmaf(wgx, gy, xysum);
        }
        const float iw = 1.0f / wsum;
        const float varXW = (xxsum - xsum * xsum * iw) * iw, varYW = (yysum - ysum * ysum * iw) * iw, varXYW = (xysum - xsum * ysum * iw) * iw;
        const float rawSim = varXYW / sqrtf(varXW * varYW);
        const float fsimSubpart = isfinite(rawSim) ? -rawSim : 1.0f;
        if(fsimSubpart < 0.f)
        {
            fsim += (TInvert ? sigmoid(0.0f, 1.0f, 0.7f, -0.7f, fsimSubpart) : fsimSubpart) * part.weight;
            wsumParts += part.weight;
        }
    }
    if(wsumParts == 0.f)
        return INFINITY;
    return TInvert ? fsim : fsim / wsumParts; // Refine does not average (Patch.cuh:764-768)
}

// REFINE = false: volume_computeSimilarity_kernel (kernels.cuh:109-233); REFINE = true: volume_refineSimilarity_kernel (:235-391)
template <bool REFINE>
__global__ void __launch_bounds__(256)
  similarity_cs_kernel(uint8_t* __restrict__ best, uint8_t* __restrict__ second, __half* __restrict__ vol, int volDimZ,
                       const float2* __restrict__ sgmDepthPixSize, int map_pitch, const float* __restrict__ sgmNormal, int normal_pitch,
                       long long pitch_y, int pitch_x, const float* __restrict__ depths, avdm_camera_t rc, avdm_camera_t tc, NccArgs A, PatchTable tab,
                       CsArgs S, int stepXY, unsigned zBegin, unsigned zEnd, avdm_roi_t roi)
{
    constexpr int CH = REFINE ? 8 : 4; // planes per lane, written as one 16-byte / 4-byte word like the main kernels
    const unsigned vx = blockIdx.x * 64 + (threadIdx.x & 63), vy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const unsigned z0 = ((zBegin / CH) + blockIdx.z) * CH;
    const int wsh = A.wsh;
    const float x = (float)(roi.x.begin + vx) * (float)stepXY;
    const float y = (float)(roi.y.begin + vy) * (float)stepXY;
    const f3 C = ld3(rc.C), Z = ld3(rc.ZVect);
    const f3 v = normalize(M3x3mulV2(rc.iP, x, y));
    const float dd = (float)wsh + 2.0f;
    const bool rInside = S.useCustomPatchPattern || !((x < dd) || (x > A.rcW1 - dd) || (y < dd) || (y > A.rcH1 - dd));
    // knife-edge rows: the reference's own border test per plane (lit::, see above); the custom pattern has its own 2-pixel margin on the exact pixel
    const bool knife = AVDM_KNIFE_LITERAL && !S.useCustomPatchPattern && rInside && (x == dd || x == A.rcW1 - dd || y == dd || y == A.rcH1 - dd);

    float2 dps = make_float2(-1.f, 0.f);
    f3 pMid = C, dir = v;
    bool pixActive = true;
    if(REFINE)
    {
        dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
        pixActive = dps.x > 0.0f;
        pMid = C + v * dps.x;
        dir = normalize(pMid - C);
    }
    const float dnC = dot(Z, C), dnv = dot(Z, v);

    unsigned wb = 0, ws = 0;
    uint4 packed = make_uint4(0u, 0u, 0u, 0u);
    uint8_t *pb = nullptr, *ps = nullptr;
    __half* pv = nullptr;
    if(REFINE)
    {
        pv = vol + ((long long)vy * pitch_y + (long long)vx * pitch_x) / 2 + z0;
        if(pixActive)
            packed = *reinterpret_cast<const uint4*>(pv);
    }
    else
    {
        pb = best + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
        ps = second + (long long)vy * pitch_y + (long long)vx * pitch_x + z0;
        wb = *reinterpret_cast<const unsigned*>(pb);
        ws = *reinterpret_cast<const unsigned*>(ps);
    }

    const LodTap rtap0{S.rcT, 0.f, 1.0f / (float)A.rcL.W, 1.0f / (float)A.rcL.H}, ttap0{S.tcT, 0.f, 1.0f / (float)A.tcL.W, 1.0f / (float)A.tcL.H};
#pragma unroll 1
    for(int k = 0; k < CH; ++k)
    {
        const unsigned vz = z0 + k;
        if(vz < zBegin || vz >= zEnd)
            continue;
        float s = INFINITY;
        if(pixActive && rInside)
        {
            f3 p;
            if(REFINE)
            {
                const int rel = (int)vz - ((volDimZ - 1) / 2);
                p = rel != 0 ? pMid + dir * ((float)rel * dps.y) : pMid;
            }
            else
            {
                const f3 planep = C + Z * depths[vz];
                p = C + v * ((dot(planep, Z) - dnC) / dnv);
            }
            const float pd = computePixSize(rc, p);
            f3 ax, ay;
            if(REFINE && sgmNormal != nullptr)
            {
                const f3 v1 = normalize(C - p), v2 = normalize(ld3(tc.C) - p);
                ay = normalize(cross(v1, v2));
                const float* nn = (const float*)((const char*)sgmNormal + (long long)vy * normal_pitch) + 3 * vx;
                ax = normalize(cross(ay, f3{nn[0], nn[1], nn[2]}));
            }
            else
                patch_axes(rc, tc, p, ax, ay);
            const PatchProj Q = make_patch_proj(rc, tc, p, ax, ay, pd);
            const float it0 = fast_rcp(Q.ht0.z);
            const float tpx = Q.ht0.x * it0, tpy = Q.ht0.y * it0;
            if(S.useCustomPatchPattern)
                s = ncc_custom_pattern<REFINE>(Q, A, S, rc, tc, x, y, tpx, tpy, p);
            else if(!((tpx < dd) || (tpx > A.tcW1 - dd) || (tpy < dd) || (tpy > A.tcH1 - dd)) &&
                    (!knife || (REFINE ? lit::refine_r_inside(rc, x, y, dps.x, dps.y, (int)vz - ((volDimZ - 1) / 2), dd, A.rcW1, A.rcH1)
                                       : lit::sgm_r_inside(rc, x, y, depths[REFINE ? 0 : vz], dd, A.rcW1, A.rcH1))))
            {
                float rcLevel = S.mipmapLevel, tcLevel = S.mipmapLevel;
                if(S.useConsistentScale)
                    computeRcTcMipmapLevels(rcLevel, tcLevel, S.mipmapLevel, rc, tc, x, y, tpx, tpy, p);
                LodTap rt = rtap0, tt = ttap0;
                rt.lod = rcLevel;
                tt.lod = tcLevel;
                const float4 rcCenter = rt.fetch<true>(fmaf(x, A.rcSx, A.rcOx), fmaf(y, A.rcSy, A.rcOy));
                const float4 tcCenter = tt.fetch<true>(fmaf(tpx, A.tcSx, A.tcOx), fmaf(tpy, A.tcSy, A.tcOy));
                if(!(rcCenter.w < (255.f * 0.9f) || tcCenter.w < (255.f * 0.4f)))
                    s = ncc_accumulate<true, 0, REFINE>(Q, A, tab, rt, tt, rcCenter, tcCenter);
            }
        }
        if(REFINE)
        {
            if(pixActive && s != INFINITY)
            { // packed[k] += s (invalid patches add nothing, kernels.cuh:373-381)
                const unsigned sel = (unsigned)k >> 1, hiHalf = (unsigned)k & 1u;
                unsigned word = sel == 0 ? packed.x : (sel == 1 ? packed.y : (sel == 2 ? packed.z : packed.w));
                const unsigned short hbits = (unsigned short)(hiHalf ? (word >> 16) : (word & 0xffffu));
                const __half hs = __float2half(__half2float(__ushort_as_half(hbits)) + s);
                const unsigned nb = (unsigned)__half_as_ushort(hs);
                word = hiHalf ? ((word & 0x0000ffffu) | (nb << 16)) : ((word & 0xffff0000u) | nb);
                packed.x = sel == 0 ? word : packed.x;
                packed.y = sel == 1 ? word : packed.y;
                packed.z = sel == 2 ? word : packed.z;
                packed.w = sel == 3 ? word : packed.w;
            }
        }
        else
        {
            float fsim = 255.0f;
            if(s != INFINITY)
            {
                s = (s + 1.0f) * 0.5f;
                s = fminf(1.0f, fmaxf(0.0f, s));
                fsim = s * 254.0f;
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
    }
    if(REFINE)
    {
        if(pixActive)
            *reinterpret_cast<uint4*>(pv) = packed;
    }
    else
    {
        *reinterpret_cast<unsigned*>(pb) = wb;
        *reinterpret_cast<unsigned*>(ps) = ws;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static avdm_patch_pattern_t g_patchPattern = {}; // the reference's constantPatchPattern_d (DevicePatchPattern.hpp:51)
static bool g_patchPatternSet = false;
static std::mutex g_patchPatternMutex;           // one host thread per device may build / read it (computeOnMultiGPUs)
static unsigned* g_stats = nullptr; // device counters, allocated on first use when AVDM_SIM_STATS=1
static unsigned* g_outlierTotals = nullptr; // {units worked off, units refused by a full list} of refine_outlier_kernel when AVDM_REFINE_OUTLIER_STATS=1
// units the outlier lists of this process could not take (a full list leaves a wave on the slower per-plane path; the results do not change):
// one counter per device, bumped by refine_outlier_kernel, read by avdm_refine_outlier_refused()
static unsigned* g_outlierRefused[64] = {};
static std::mutex g_outlierRefusedMutex;
static unsigned* outlier_refused_counter()
{
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_outlierRefusedMutex);
    unsigned*& p = g_outlierRefused[dev & 63];
    if(p == nullptr)
    {
        if(hipMalloc((void**)&p, sizeof(unsigned)) != hipSuccess)
            p = nullptr;
        else
            (void)hipMemset(p, 0, sizeof(unsigned));
    }
    return p;
}
// Capacity of the outlier list of a launch over nPix pixels and nchunks chunks of 8 planes: EVERY unit the sweep can append — one per (pixel,
// chunk) from the eight-plane pass, two from the four-plane form (a unit per quad) — so that a full list cannot happen (8 B per unit: 64 MB
// for a 1024 x 1024 tile, 768 MB for an undivided 12 MP frame, of 288 GB).  Rounds 5-6a sized it at a quarter of the pairs ("40 x the bench's
// lists"); the program's run on the wide-baseline scene of scripts/cli_e2e_cfg3.py then logged 348 670 refused units (session r06_g) — a
// refused wave takes the one-plane path, which is slower AND makes WHICH waves do so depend on the order of the atomics, i.e. the volume's
// last fp16 bit on the schedule.  With room for every unit the sweep's result is a function of its inputs again.
static unsigned outlier_list_capacity(size_t nPix, unsigned nchunks)
{
    return (unsigned)std::min<size_t>(std::max<size_t>(2 * nPix * nchunks, 4096), 0x7fffff00u);
}

static unsigned* outlier_totals()
{
    const char* e = getenv("AVDM_REFINE_OUTLIER_STATS");
    if(!(e != nullptr && e[0] == '1'))
        return nullptr;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if(g_outlierTotals == nullptr)
    {
        if(hipMalloc((void**)&g_outlierTotals, 2 * sizeof(unsigned)) != hipSuccess)
            g_outlierTotals = nullptr;
        else
            (void)hipMemset(g_outlierTotals, 0, 2 * sizeof(unsigned));
    }
    return g_outlierTotals;
}

// paired: in = the caller would like the 16-byte paired records (FIXED8 pyramids only); out = whether the LDS budget allows them
// fractional: out = the stage's level of detail is not an integral level of the pyramids (scales that are not a power-of-two multiple of the
// pyramid's first level, e.g. --sgmScale 3 --refineScale 1): the caller runs the plain trilinear kernel (similarity_cs_kernel), whose taps
// go through the software texture unit at A.mipmapLevel; the sample positions are then expressed in the texel space of the level BELOW
// budgetBytes: dynamic LDS a workgroup may use — a third of the compute unit's 160 KiB (three workgroups fit by LDS) unless the caller launches an
// instantiation it has raised the limit for (the default ones: two workgroups per compute unit by registers anyway, half of the LDS each);
// rec12: out (optional) = 12-byte records instead of the half-paired 8-byte ones (they fit and the caller's instantiation reads them)
constexpr int kLdsThird = 42 * 1280 - 256, kLdsHalf = 64 * 1280 - 256;
static bool fill_ncc_args(NccArgs& A, PatchTable& tab, const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, int scale, int stepXY, int wsh,
                          double gammaC, double gammaP, bool& paired, bool& fractional, int budgetBytes = kLdsThird, bool* rec12 = nullptr)
{
    int rl, tl;
    const bool rInt = lod_is_integral(rcPyr, scale, &rl), tInt = lod_is_integral(tcPyr, scale, &tl);
    fractional = !(rInt && tInt);
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
    if(fractional)
    { // DeviceMipmapImage::getLevel of the R image, clamped like tex2DLod clamps it; both images are sampled at it (Patch.cuh:499-505)
        const float maxl = (float)(rcPyr->levels - 1);
        const float l = tex_level_of(rcPyr, scale);
        A.mipmapLevel = !(l > 0.0f) ? 0.0f : (l > maxl ? maxl : l);
    }
    A.wsh = wsh;
    const int n = 2 * wsh + 1;
    for(int yp = -wsh; yp <= wsh; ++yp)
        for(int xp = -wsh; xp <= wsh; ++xp)
            tab.c[(yp + wsh) * n + (xp + wsh)] = 2.0f * sqrtf((float)(xp * xp + yp * yp)) * A.invGammaP * log2e;

    // LDS budget: R tile = 16 stage pixels * stepXY texels + halo; T window: the same footprint (a T view at a
    // markedly larger scale, or a depth edge inside the workgroup, overflows it and takes the generic path).  Row pitches are 8 (mod 16) texels.
    const int rw = 15 * stepXY + 2 * (wsh + 2) + 5;
    const int tw = rw + 1;
    // bytes of dynamic LDS per workgroup.  LDS is allocated in granules (1280 B on gfx950, 160 KiB / 128): 42 granules per workgroup =
    // 53 760 B, three of them 161 280 of the 163 840 B (64 granules: two workgroups).  The static shared state (BlockShared, < 256 B) must fit
    // in the same granules: when it grew by 20 B (round 2: one more box for the chunk window) the old budget (160 KiB / 3 - 1 KiB = 53 589 B)
    // tipped into a 43rd granule, only TWO workgroups fitted a CU and both kernels ran 20 % slower with identical sample loops.
    const int budget = budgetBytes;
    bool r12 = false;
    {
        const char* pe = getenv("AVDM_SIM_PAIRED");
        if(pe && pe[0] == '0')
            paired = false;
        // the paired layout doubles the bytes per texel: only when the R tile and a T window of 1.5 x the R footprint still fit
        if(paired && (lds_pitch_for(rw) * rw * 2 + lds_pitch_for(tw) * tw * 3) * 8 > budget)
            paired = false;
        // 12-byte records (the three dot2 operands of a tap, no alpha) under the same condition
        const char* p12 = getenv("AVDM_SIM_REC12");
        if(!paired && rec12 != nullptr && !(p12 && p12[0] == '0') && (lds_pitch_for(rw) * rw * 2 + lds_pitch_for(tw) * tw * 3) * 6 <= budget)
            r12 = true;
        if(rec12 != nullptr)
            *rec12 = r12;
    }
    auto units = [&](int n) { return paired ? 2 * n : (r12 ? (3 * n + 1) / 2 : n); }; // 8-byte units (lds_units in the kernels)
    A.rpitch = lds_pitch_for(rw);
    A.rcap = units(lds_pitch_for(rw) * rw);
    A.tcap = units(lds_pitch_for(tw) * tw);
    // the T window may use what is left of the budget — a T view at a larger scale or a slanted surface then still runs from LDS (3 % of
    // the plane-workgroups of cfg3 overflowed the R-sized window and paid the ~4x slower generic path)
    {
        const int room = budget / 8 - A.rcap;
        if(room > A.tcap)
            A.tcap = room;
    }
    if((A.rcap + A.tcap) * 8 > (budget > 60 * 1024 ? budget : 60 * 1024))
    { // keep >= 2 workgroups per CU; larger steps take the generic path
        A.rcap = 16;
        A.tcap = 16;
        A.forceGeneric = 1;
    }
    else
        A.forceGeneric = 0;
    const char* e = getenv("AVDM_SIM_LDS");
    if(e && e[0] == '0')
        A.forceGeneric = 1;
    const char* pk = getenv("AVDM_SIM_PACKED");
    A.noPacked = (pk && pk[0] == '0') ? 1 : 0;
    const char* cw = getenv("AVDM_SIM_CHUNK_WINDOW"); // 0: one T window per plane (the A/B reference of the chunk window)
    A.chunkWindow = (cw && cw[0] == '0') ? 0 : 1;
    const char* pp = getenv("AVDM_SIM_PLANE_PAIRS"); // 0: one plane per pass over the patch (the A/B reference of the plane pairs)
    A.planePairs = (pp && pp[0] == '0') ? 0 : 1;
    A.stats = nullptr;
    const char* st = getenv("AVDM_SIM_STATS");
    if(st && st[0] == '1')
    {
        if(g_stats == nullptr)
        {
            if(hipMalloc((void**)&g_stats, 4 * sizeof(unsigned)) != hipSuccess)
                g_stats = nullptr;
            else
                (void)hipMemset(g_stats, 0, 4 * sizeof(unsigned));
        }
        A.stats = g_stats;
    }
    return true;
}

// AVDM_SIM_LITERAL=1 (read at each call): both similarity entry points run the reference's arithmetic as written (avdm_literal.hip)
int literal_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                               const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_sgm_params_t* sp, avdm_range_t dr,
                               avdm_roi_t roi, void* stream);
int literal_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal,
                              int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                              const avdm_refine_params_t* rp, avdm_range_t dr, avdm_roi_t roi, void* stream);
static bool sim_literal_mode()
{
    const char* e = getenv("AVDM_SIM_LITERAL");
    return e != nullptr && e[0] == '1';
}
// avdm_sgm_params_t::referenceArithmetic / avdm_refine_params_t::referenceArithmetic: the reference's arithmetic as written, to the bits of the
// pinned reference build, from LDS windows (avdm_literal.hip: strict_sgm_kernel / strict_refine_kernel) — the product's parity mode
int strict_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                              const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr, const avdm_sgm_params_t* sp, avdm_range_t dr,
                              avdm_roi_t roi, void* stream);
int strict_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ, const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal,
                             int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                             const avdm_refine_params_t* rp, avdm_range_t dr, avdm_roi_t roi, void* stream);

} // namespace avdm

using namespace avdm;

extern "C" {

int avdm_build_custom_patch_pattern(int n_subparts, const avdm_patch_subpart_params_t* subparts, int group, avdm_patch_pattern_t* out)
{
    // patchPattern.cpp:18-80: checks
    if(n_subparts <= 0 || subparts == nullptr)
        return set_error_msg(1, "Cannot build custom patch pattern: No patch pattern subpart given.");
    std::map<int, int> nbCoordsPerSubparts; // <level or subpart index, nb coordinates>
    for(int i = 0; i < n_subparts; ++i)
    {
        const avdm_patch_subpart_params_t& sp = subparts[i];
        if(sp.radius <= 0.f)
            return set_error_msg(1, "Cannot build custom patch pattern: A patch pattern subpart radius is incorrect.");
        if(sp.isCircle && sp.nbCoordinates <= 0)
            return set_error_msg(1, "Cannot build custom patch pattern: A patch pattern subpart circle number of coordinates is incorrect.");
        if(group)
        {
            if(!sp.isCircle && nbCoordsPerSubparts.find(sp.level) != nbCoordsPerSubparts.end())
                return set_error_msg(1, "Cannot build custom patch pattern: Cannot group more than one full patch pattern subpart.");
            nbCoordsPerSubparts[sp.level] += sp.isCircle ? sp.nbCoordinates : 0;
        }
        else
            nbCoordsPerSubparts[i] += sp.isCircle ? sp.nbCoordinates : 0;
    }
    int maxSubpartCoords = 0;
    for(const auto& kv : nbCoordsPerSubparts)
        maxSubpartCoords = std::max(maxSubpartCoords, kv.second);
    const int nbSubparts = (int)nbCoordsPerSubparts.size();
    if(nbSubparts > AVDM_PATCH_MAX_SUBPARTS)
        return set_error_msg(1, "Cannot build custom patch pattern: Too many patch pattern subpart given.");
    if(maxSubpartCoords > AVDM_PATCH_MAX_COORDS_PER_SUBPART)
        return set_error_msg(1, "Cannot build custom patch pattern: Too many patch pattern subpart coordinates given.");

    avdm_patch_pattern_t pp = {};
    pp.nbSubparts = nbSubparts;
    auto fillCircle = [](avdm_patch_pattern_subpart_t& part, int first, const avdm_patch_subpart_params_t& sp) {
        const float angleDifference = (float)((M_PI * 2.f) / sp.nbCoordinates); // double division, then float (patchPattern.cpp:138)
        for(int i = 0; i < sp.nbCoordinates; ++i)
        {
            const float radians = angleDifference * (float)i;
            part.coordinates[first + i][0] = std::cos(radians) * sp.radius;
            part.coordinates[first + i][1] = std::sin(radians) * sp.radius;
        }
    };
    if(group)
    {
        for(int i = 0; i < n_subparts; ++i)
        {
            const avdm_patch_subpart_params_t& sp = subparts[i];
            avdm_patch_pattern_subpart_t& part = pp.subparts[std::distance(nbCoordsPerSubparts.begin(), nbCoordsPerSubparts.find(sp.level))];
            if(sp.isCircle)
            {
                fillCircle(part, part.nbCoordinates, sp);
                part.wsh = std::max(part.wsh, int(sp.radius + std::pow(2.f, (float)sp.level - 1.f)));
                part.nbCoordinates += sp.nbCoordinates;
            }
            else
                part.wsh = std::max(part.wsh, int(sp.radius));
            part.level = (float)sp.level;
            part.downscale = std::pow(2.f, part.level);
            part.weight = sp.weight;
            part.isCircle = sp.isCircle ? 1 : 0;
        }
    }
    else
    {
        for(int i = 0; i < nbSubparts; ++i)
        {
            const avdm_patch_subpart_params_t& sp = subparts[i];
            avdm_patch_pattern_subpart_t& part = pp.subparts[i];
            if(sp.isCircle)
            {
                fillCircle(part, 0, sp); // the reference divides by and loops over the not yet assigned subpart.nbCoordinates (:196-201)
                part.wsh = int(sp.radius + std::pow(2.f, (float)sp.level - 1.f));
                part.nbCoordinates = sp.nbCoordinates;
            }
            else
            {
                part.wsh = int(sp.radius);
                part.nbCoordinates = 0;
            }
            part.level = (float)sp.level;
            part.downscale = std::pow(2.f, part.level);
            part.weight = sp.weight;
            part.isCircle = sp.isCircle ? 1 : 0;
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_patchPatternMutex);
        g_patchPattern = pp;
        g_patchPatternSet = true;
    }
    if(out != nullptr)
        *out = pp;
    return 0;
}

/* debugging aid (not part of avdm.h): plane-workgroups since the last call {LDS path, generic: R tile, generic: T outside, generic: T too large} */
int avdm_debug_similarity_stats(unsigned out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if(g_stats == nullptr)
        return 0;
    unsigned h[4];
    if(hipMemcpy(h, g_stats, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    (void)hipMemset(g_stats, 0, sizeof(h));
    for(int i = 0; i < 4; ++i)
        out[i] = h[i];
    return 0;
}

/* debugging aid (not part of avdm.h): the pass counters of a -DAVDM_LEAN_STATS=1 variant build (zeros in the shipped library), read and cleared */
int avdm_debug_lean_stats(unsigned out[32])
{
    for(int i = 0; i < 32; ++i)
        out[i] = 0;
#if AVDM_LEAN_STATS
    if(hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(g_leanStats), 32 * sizeof(unsigned)) != hipSuccess)
        return 1;
    const unsigned zero[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_leanStats), zero, sizeof(zero));
#endif
    return 0;
}

/* debugging aid (not part of avdm.h; AVDM_REFINE_OUTLIER_STATS=1): units of the Refine outlier list since the last call {worked off, refused by a full list} */
int avdm_debug_refine_outlier_units(unsigned out[2])
{
    out[0] = out[1] = 0;
    if(g_outlierTotals == nullptr)
        return 0;
    if(hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, g_outlierTotals, 2 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    (void)hipMemset(g_outlierTotals, 0, 2 * sizeof(unsigned));
    return 0;
}

/* Bytes of per-stream scratch avdm_volume_refine_similarity takes from the library's block for its outlier list (the default kernels) when it
 * sweeps `n_pixels` pixels x `n_planes` planes: what a scheduler prices a tile slot with (host/DepthMapEstimator.cpp: getNbSimultaneousTiles). */
size_t avdm_refine_similarity_scratch_bytes(size_t n_pixels, int n_planes)
{
    if(n_pixels == 0 || n_planes <= 0)
        return 0;
    return 8 + (size_t)outlier_list_capacity(n_pixels, (unsigned)((n_planes + 7) / 8)) * sizeof(uint2);
}

/* Units (pixel, chunk of planes) that found an outlier list FULL on the current device since the last call (they ran on the slower per-plane
 * path).  Waits for the device.  Cannot happen since the capacity holds every unit a sweep can append (outlier_list_capacity) unless the
 * sweep exceeds 2^31 units; a scheduler logs it (host/DepthMapEstimator.cpp) so that it cannot go unnoticed if it ever does. */
int avdm_refine_outlier_refused(unsigned* out)
{
    *out = 0;
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return 1;
    unsigned* p;
    {
        std::lock_guard<std::mutex> lock(g_outlierRefusedMutex);
        p = g_outlierRefused[dev & 63];
    }
    if(p == nullptr)
        return 0;
    if(hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, p, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess)
        return set_error_msg(1, "avdm_refine_outlier_refused: reading the counter failed");
    (void)hipMemset(p, 0, sizeof(unsigned));
    return 0;
}

int avdm_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                                   const avdm_camera_t* tc, const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                                   const avdm_sgm_params_t* sp, avdm_range_t dr, avdm_roi_t roi, void* stream)
{
    if(dr.end <= dr.begin || roi.x.end <= roi.x.begin || roi.y.end <= roi.y.begin)
        return 0;
    if(sp->wsh < 1 || sp->wsh > 4)
        return set_error_msg(1, "avdm_volume_compute_similarity: wsh must be in [1, 4]");
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)best & 3) || ((uintptr_t)second & 3))
        return set_error_msg(1, "avdm_volume_compute_similarity: volume base / pitches must be multiples of 4 bytes");
    if(sim_literal_mode() && !sp->useConsistentScale && !sp->useCustomPatchPattern)
        return avdm::literal_compute_similarity(best, second, pitch_y, pitch_x, depths, rc, tc, rc_pyr, tc_pyr, sp, dr, roi, stream);
    if(sp->referenceArithmetic)
    {
        if(sp->useConsistentScale || sp->useCustomPatchPattern)
            return set_error_msg(1, "avdm_volume_compute_similarity: referenceArithmetic is built for the wsh-square patch at one level of detail "
                                    "(not with useConsistentScale / useCustomPatchPattern)");
        return avdm::strict_compute_similarity(best, second, pitch_y, pitch_x, depths, rc, tc, rc_pyr, tc_pyr, sp, dr, roi, stream);
    }
    NccArgs A;
    PatchTable tab;
    const bool fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
    bool paired = fixed8, fractional = false, rec12 = false;
    // the default instantiation (scale 2, stepXY 2, wsh 4) runs two workgroups per compute unit by registers: it gets half of the LDS and
    // 12-byte records; everything else keeps a third (fill_ncc_args).  Whether the call runs it is known before the LDS layout, except for the
    // A/B switches: those fall back to the general instantiations and their layout.
    const bool wantDefault = fixed8 && sp->wsh == 4 && lds_pitch_for(15 * sp->stepXY + 2 * (sp->wsh + 2) + 5) == 56;
    fill_ncc_args(A, tab, rc_pyr, tc_pyr, sp->scale, sp->stepXY, sp->wsh, sp->gammaC, sp->gammaP, paired, fractional, wantDefault ? kLdsHalf : kLdsThird,
                  wantDefault ? &rec12 : nullptr);
    bool runDefault = wantDefault && !paired && !A.noPacked && !A.forceGeneric && A.chunkWindow && A.planePairs && A.stats == nullptr && !fractional &&
                      !sp->useConsistentScale && !sp->useCustomPatchPattern;
    if(wantDefault && !runDefault)
    {
        paired = fixed8;
        rec12 = false;
        fill_ncc_args(A, tab, rc_pyr, tc_pyr, sp->scale, sp->stepXY, sp->wsh, sp->gammaC, sp->gammaP, paired, fractional);
    }
    const unsigned nchunks = ((dr.end + 3) >> 2) - (dr.begin >> 2);
    if(((dr.end + 3) & ~3u) > (unsigned)pitch_x)
        return set_error_msg(1, "avdm_volume_compute_similarity: pitch_x too small for the depth range (must cover the 4-aligned range)");
    if(sp->useCustomPatchPattern && !g_patchPatternSet)
        return set_error_msg(1, "avdm_volume_compute_similarity: useCustomPatchPattern without a pattern (avdm_build_custom_patch_pattern)");
    if(sp->useConsistentScale || sp->useCustomPatchPattern || fractional)
    {
        CsArgs S;
        S.rcT = make_tex(rc_pyr);
        S.tcT = make_tex(tc_pyr);
        S.mipmapLevel = A.mipmapLevel;
        S.useConsistentScale = sp->useConsistentScale;
        S.useCustomPatchPattern = sp->useCustomPatchPattern;
        {
            std::lock_guard<std::mutex> lock(g_patchPatternMutex);
            S.pattern = g_patchPattern;
        }
        hipLaunchKernelGGL((similarity_cs_kernel<false>), dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), nchunks), dim3(256), 0,
                           (hipStream_t)stream, best, second, (__half*)nullptr, 0, (const float2*)nullptr, 0, (const float*)nullptr, 0, pitch_y, pitch_x,
                           depths, *rc, *tc, A, tab, S, sp->stepXY, dr.begin, dr.end, roi);
        AVDM_LAUNCH_CHECK("avdm_volume_compute_similarity(consistent scale)");
    }
    dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), divUp(nchunks, kSgmChunksPerWg));
    const size_t lds = (size_t)(A.rcap + A.tcap) * sizeof(uint2);
#define LAUNCH(F8, W, PR)                                                                                                                                 \
    hipLaunchKernelGGL((similarity_kernel<F8, W, PR>), grid, dim3(256), lds, (hipStream_t)stream, best, second, pitch_y, pitch_x, depths, *rc, *tc, A, tab, \
                       sp->stepXY, dr.begin, dr.end, roi)
    if(runDefault)
    {
        // the default: scale 2, stepXY 2, wsh 4 — eight planes per pass (AVDM_SIM_PLANES8=0: four), 12-byte records, up to half of the compute
        // unit's LDS; one launch
        int dev = 0;
        (void)hipGetDevice(&dev);
#define AVDM_SGM_DEFAULT_LAUNCH(P, R12)                                                                                                                   \
    {                                                                                                                                                     \
        static std::once_flag once[64]; /* the attribute belongs to the function on ONE device */                                                        \
        std::call_once(once[dev & 63], [&] {                                                                                                              \
            (void)hipFuncSetAttribute((const void*)similarity_kernel<true, 4, false, 56, P, R12>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalf);  \
        });                                                                                                                                               \
        hipLaunchKernelGGL((similarity_kernel<true, 4, false, 56, P, R12>), grid, dim3(256), lds, (hipStream_t)stream, best, second, pitch_y, pitch_x,     \
                           depths, *rc, *tc, A, tab, sp->stepXY, dr.begin, dr.end, roi);                                                                 \
    }
        // read at each call (12-byte records only): AVDM_SIM_PLANES8=0 — four planes per pass everywhere (the round-4 default)
        const char* p8 = getenv("AVDM_SIM_PLANES8");
        const bool planes8 = !(p8 != nullptr && p8[0] == '0'); // eight planes per pass: the default since round 5
        if(rec12 && planes8)
            AVDM_SGM_DEFAULT_LAUNCH(8, true)
        else if(rec12)
            AVDM_SGM_DEFAULT_LAUNCH(4, true)
        else
            AVDM_SGM_DEFAULT_LAUNCH(4, false)
#undef AVDM_SGM_DEFAULT_LAUNCH
        return ::avdm::set_error(hipGetLastError(), "avdm_volume_compute_similarity"); // the one launch of the default path: done
    }
    else if(fixed8 && paired)
    {
        if(sp->wsh == 4) LAUNCH(true, 4, true);
        else if(sp->wsh == 3) LAUNCH(true, 3, true);
        else LAUNCH(true, 0, true);
    }
    else if(fixed8)
    {
        if(sp->wsh == 4) LAUNCH(true, 4, false);
        else if(sp->wsh == 3) LAUNCH(true, 3, false);
        else LAUNCH(true, 0, false);
    }
    else
    {
        if(sp->wsh == 4) LAUNCH(false, 4, false);
        else if(sp->wsh == 3) LAUNCH(false, 3, false);
        else LAUNCH(false, 0, false);
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
    if((pitch_x & 15) || (pitch_y & 15) || ((uintptr_t)vol_f16 & 15))
        return set_error_msg(1, "avdm_volume_refine_similarity: volume base / pitches must be multiples of 16 bytes");
    if((int)(((dr.end + 7) & ~7u) * 2) > pitch_x)
        return set_error_msg(1, "avdm_volume_refine_similarity: pitch_x too small (must cover the 8-aligned depth range)");
    if(sim_literal_mode() && !rp->useConsistentScale && !rp->useCustomPatchPattern)
        return avdm::literal_refine_similarity(vol_f16, pitch_y, pitch_x, dimZ, sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, rc, tc, rc_pyr, tc_pyr, rp,
                                               dr, roi, stream);
    if(rp->referenceArithmetic)
    {
        if(rp->useConsistentScale || rp->useCustomPatchPattern)
            return set_error_msg(1, "avdm_volume_refine_similarity: referenceArithmetic is built for the wsh-square patch at one level of detail "
                                    "(not with useConsistentScale / useCustomPatchPattern)");
        return avdm::strict_refine_similarity(vol_f16, pitch_y, pitch_x, dimZ, sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, rc, tc, rc_pyr, tc_pyr, rp,
                                              dr, roi, stream);
    }
    NccArgs A;
    PatchTable tab;
    const bool fixed8 = rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8;
    bool paired = fixed8, fractional = false;
    // the default instantiation (scale 1, stepXY 1, wsh 3) gets half of the compute unit's LDS like the SGM one (see there): its T window —
    // the hull of a chunk of 31 planes, or of one plane when SGM outliers inside the workgroup stretch it — then holds 62 x 62 instead of
    // 46 x 46 paired records before the workgroup has to take its taps from global memory
    const bool wantDefault = fixed8 && rp->wsh == 3 && lds_pitch_for(15 * rp->stepXY + 2 * (rp->wsh + 2) + 5) == 40;
    fill_ncc_args(A, tab, rc_pyr, tc_pyr, rp->scale, rp->stepXY, rp->wsh, rp->gammaC, rp->gammaP, paired, fractional, wantDefault ? kLdsHalf : kLdsThird);
    const bool runDefault = wantDefault && paired && !A.noPacked && !A.forceGeneric && A.chunkWindow && A.planePairs && A.stats == nullptr && !fractional &&
                            !rp->useConsistentScale && !rp->useCustomPatchPattern;
    if(wantDefault && !runDefault)
    {
        paired = fixed8;
        fill_ncc_args(A, tab, rc_pyr, tc_pyr, rp->scale, rp->stepXY, rp->wsh, rp->gammaC, rp->gammaP, paired, fractional);
    }
    const unsigned nchunks = ((dr.end + 7) >> 3) - (dr.begin >> 3);
    if(rp->useCustomPatchPattern && !g_patchPatternSet)
        return set_error_msg(1, "avdm_volume_refine_similarity: useCustomPatchPattern without a pattern (avdm_build_custom_patch_pattern)");
    if(rp->useConsistentScale || rp->useCustomPatchPattern || fractional)
    {
        CsArgs S;
        S.rcT = make_tex(rc_pyr);
        S.tcT = make_tex(tc_pyr);
        S.mipmapLevel = A.mipmapLevel;
        S.useConsistentScale = rp->useConsistentScale;
        S.useCustomPatchPattern = rp->useCustomPatchPattern;
        {
            std::lock_guard<std::mutex> lock(g_patchPatternMutex);
            S.pattern = g_patchPattern;
        }
        hipLaunchKernelGGL((similarity_cs_kernel<true>), dim3(divUp(roi.x.end - roi.x.begin, 64), divUp(roi.y.end - roi.y.begin, 4), nchunks), dim3(256), 0,
                           (hipStream_t)stream, (uint8_t*)nullptr, (uint8_t*)nullptr, (__half*)vol_f16, dimZ, (const float2*)sgm_depth_pixsize, map_pitch,
                           sgm_normal, normal_pitch, pitch_y, pitch_x, (const float*)nullptr, *rc, *tc, A, tab, S, rp->stepXY, dr.begin, dr.end, roi);
        AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity(consistent scale)");
    }
    dim3 grid(divUp(roi.x.end - roi.x.begin, 16), divUp(roi.y.end - roi.y.begin, 16), divUp(nchunks, kRefineChunksPerWg));
    const size_t lds = (size_t)(A.rcap + A.tcap) * sizeof(uint2);
#define LAUNCH(F8, W, PR)                                                                                                                          \
    hipLaunchKernelGGL((refine_similarity_kernel<F8, W, PR>), grid, dim3(256), lds, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x, dimZ,   \
                       (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, A, tab, rp->stepXY, dr.begin, dr.end, roi, \
                       (unsigned*)nullptr, 0u)
    if(runDefault)
    {
        // the default: scale 1, stepXY 1, wsh 3 — eight planes per pass (AVDM_REFINE_PLANES8=0: four), 16-byte records, up to half of the compute
        // unit's LDS; the sweep kernel + the kernel that works off its outlier list
        {
            static std::once_flag once0[64]; // the attribute belongs to the function on ONE device
            int dev0 = 0;
            (void)hipGetDevice(&dev0);
            std::call_once(once0[dev0 & 63], [&] {
                (void)hipFuncSetAttribute((const void*)refine_similarity_kernel<true, 3, true, 40, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalf);
            });
            // AVDM_REFINE_PLANES8=1 (experimental, read at each call): eight planes per pass on the chunks that lie in the T camera's range
            const char* p8 = getenv("AVDM_REFINE_PLANES8");
            // The outlier list (AVDM_REFINE_OUTLIER_LIST, default on; refine_similarity_kernel): lanes whose taps leave their workgroup's T window are
            // appended to a list in the stream's scratch block and worked off by refine_outlier_kernel right after the sweep — {count, pad,
            // (pixel, planes) units}; capacity = every unit the launch can append (outlier_list_capacity): the list cannot be full
            const char* ol = getenv("AVDM_REFINE_OUTLIER_LIST");
            const bool useList = !(ol != nullptr && ol[0] == '0');
            const size_t nPix = (size_t)(roi.x.end - roi.x.begin) * (roi.y.end - roi.y.begin);
            const unsigned listCap = useList ? outlier_list_capacity(nPix, nchunks) : 0u;
            std::unique_ptr<StreamScratch> lease;
            if(useList)
                lease.reset(new StreamScratch((hipStream_t)stream, 8 + (size_t)listCap * sizeof(uint2)));
            unsigned* list = useList ? (unsigned*)lease->ptr() : nullptr;
            if(useList && list == nullptr)
                return set_error_msg(2, "avdm_volume_refine_similarity: scratch allocation failed");
            if(useList)
            {
                const hipError_t me = hipMemsetAsync(list, 0, 8, (hipStream_t)stream);
                if(me != hipSuccess)
                    return set_error(me, "avdm_volume_refine_similarity");
            }
            if(!(p8 != nullptr && p8[0] == '0')) // the default since round 5 (parity tables: profiles/r05_a_parity_*.json)
            {
                static std::once_flag once8[64];
                std::call_once(once8[dev0 & 63], [&] {
                    (void)hipFuncSetAttribute((const void*)refine_similarity_kernel<true, 3, true, 40, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsHalf);
                });
                hipLaunchKernelGGL((refine_similarity_kernel<true, 3, true, 40, 8>), grid, dim3(256), lds, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x,
                                   dimZ, (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, A, tab, rp->stepXY, dr.begin, dr.end,
                                   roi, list, listCap);
            }
            else
                hipLaunchKernelGGL((refine_similarity_kernel<true, 3, true, 40, 4>), grid, dim3(256), lds, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x,
                                   dimZ, (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, A, tab, rp->stepXY, dr.begin, dr.end,
                                   roi, list, listCap);
            if(useList)
                hipLaunchKernelGGL((refine_outlier_kernel<true, 3>), dim3(kOutlierGrid), dim3(256), 0, (hipStream_t)stream, (__half*)vol_f16, pitch_y, pitch_x, dimZ,
                                   (const float2*)sgm_depth_pixsize, map_pitch, sgm_normal, normal_pitch, *rc, *tc, A, tab, rp->stepXY, dr.begin, dr.end, roi,
                                   (const unsigned*)list, listCap, outlier_totals(), outlier_refused_counter());
            return ::avdm::set_error(hipGetLastError(), "avdm_volume_refine_similarity"); // the launches of the default path: done
        }
    }
    else if(fixed8 && paired)
    {
        if(rp->wsh == 3) LAUNCH(true, 3, true);
        else if(rp->wsh == 4) LAUNCH(true, 4, true);
        else LAUNCH(true, 0, true);
    }
    else if(fixed8)
    {
        if(rp->wsh == 3) LAUNCH(true, 3, false);
        else if(rp->wsh == 4) LAUNCH(true, 4, false);
        else LAUNCH(true, 0, false);
    }
    else
    {
        if(rp->wsh == 3) LAUNCH(false, 3, false);
        else if(rp->wsh == 4) LAUNCH(false, 4, false);
        else LAUNCH(false, 0, false);
    }
#undef LAUNCH
    AVDM_LAUNCH_CHECK("avdm_volume_refine_similarity");
}

} // extern "C"
