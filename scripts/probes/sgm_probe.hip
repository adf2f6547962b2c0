// sgm_probe — what bounds the SGM pair kernel for ONE cfg3 volume (1000 x 750 x 256)?  Stand-alone (compiles in seconds, unlike
// avdm_sgm.hip); the step arithmetic is the production packed-uint16 step (csrc/avdm_sgm.hip: sgm_lstep_u16 + output stage), FULL
// dwords, NW = 1, fast path only.  Not bit-checked here: the probe answers performance questions only.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/sgm_probe.hip -o scripts/probes/sgm_probe
// Designs:
//   ring  : the production structure (two waves per column, 4 x 8 register ring, loads and stores on the same wave)
//           with switches: no stores / no loads / no arithmetic (which part of the time is whose)
//   lds   : dedicated LOADER waves stream the slices into an LDS ring with direct-to-LDS 16-byte loads (each loader has its own
//           vmcnt: no store shares it, a load instruction moves 4 slices), the compute waves read LDS and only ever issue stores
// Geometries: first launch (K = 0) walks Y (1000 columns, slice stride X*256), second (K = 2) walks X (750 columns, column stride X*256);
//             the X walk also with the row pitch padded to an odd number of 256-byte units (HBM channel spread).
// Every run also reports the shader clock seen by the kernel: clock64() ticks per wall_clock64() tick (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CK(x)                                                                                                                                                 \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        hipError_t e = (x);                                                                                                                                   \
        if(e != hipSuccess)                                                                                                                                   \
        {                                                                                                                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                                                                                                     \
            exit(1);                                                                                                                                          \
        }                                                                                                                                                     \
    } while(0)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_pk(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return as_u32(__builtin_elementwise_min(as_pk(a), as_pk(b))); }
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return as_u32(as_pk(a) + as_pk(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return as_u32(as_pk(a) - as_pk(b)); }
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned oldv, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)oldv, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
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
template <int KP1>
__device__ __forceinline__ unsigned pk_div(unsigned n)
{
    if(KP1 == 1)
        return n;
    if(KP1 == 2)
        return as_u32(as_pk(n) >> (unsigned short)1);
    if(KP1 == 4)
        return as_u32(as_pk(n) >> (unsigned short)2);
    const unsigned lo = ((n & 0xffffu) * 683u) >> 11, hi = ((n >> 16) * 683u) >> 11;
    return lo | (hi << 16);
}
template <int K>
__device__ __forceinline__ unsigned pk_avg(unsigned o, unsigned c)
{
    if(K == 0)
        return c;
    return pk_div<K + 1>(as_u32(as_pk(o) * (unsigned short)K + as_pk(c)));
}
// production step (NW = 1): updates P[2], returns min(L, 255) per plane pair in q[2]
__device__ __forceinline__ void lstep(unsigned (&P)[2], unsigned inw, unsigned iP2Pair, unsigned P1Pair, const unsigned (&keepM)[2],
                                      const unsigned (&forceV)[2], unsigned (&q)[2])
{
    unsigned m = pk_min(P[0], P[1]);
    const unsigned mlo = (unsigned)(unsigned short)m, mhi = m >> 16;
    const unsigned bestPair = wave_min_bits(min(mlo, mhi)) * 0x00010001u;
    const unsigned FPair = bestPair + iP2Pair;
    unsigned Lp[3];
    Lp[0] = __builtin_amdgcn_alignbit(P[0], (unsigned)__builtin_amdgcn_mov_dpp((int)P[1], 0x138, 0xf, 0xf, true), 16);
    Lp[1] = __builtin_amdgcn_alignbit(P[1], P[0], 16);
    Lp[2] = __builtin_amdgcn_alignbit((unsigned)__builtin_amdgcn_mov_dpp((int)P[0], 0x130, 0xf, 0xf, true), P[1], 16);
#pragma unroll
    for(int r = 0; r < 2; ++r)
    {
        const unsigned nb = pk_min(Lp[r], Lp[r + 1]);
        const unsigned mF = pk_min(pk_min(P[r], pk_add(nb, P1Pair)), FPair);
        const unsigned cur = __builtin_amdgcn_perm(0u, inw, r ? 0x0c030c02u : 0x0c010c00u);
        unsigned L = pk_add(cur, pk_sub(mF, bestPair));
        L = (L & keepM[r]) | forceV[r];
        P[r] = L;
        q[r] = pk_min(L, 0x00ff00ffu);
    }
}

enum { FIRST_FWD = 0, FIRST_REV = 1, SECOND_FWD = 2, SECOND_REV = 3 };
enum { M_NOSTORE = 1, M_NOLOAD = 2, M_NOCOMPUTE = 4 };

struct Vol
{
    const unsigned char* in;
    unsigned char* out;
    unsigned char* tmp;
    const float* p2; // [A][B]
    long long strideA, strideB;
    int A, B;
    unsigned long long* clk; // [4]: clock64 / wall_clock64 at the start and at the end of block 0 (the counters are per XCD: same block)
};

template <int ROLE, int K>
__device__ __forceinline__ unsigned out_stage(const unsigned (&q)[2], unsigned ow, unsigned tw)
{
    constexpr bool LOAD_OUT = (ROLE == SECOND_FWD) || (ROLE == SECOND_REV) || (ROLE == FIRST_FWD && K > 0);
    unsigned res[2];
#pragma unroll
    for(int h = 0; h < 2; ++h)
    {
        const unsigned c = q[h];
        const unsigned o = LOAD_OUT ? __builtin_amdgcn_perm(0u, ow, h ? 0x0c030c02u : 0x0c010c00u) : 0u;
        if(ROLE == FIRST_FWD)
            res[h] = pk_avg<K>(o, c);
        else if(ROLE == FIRST_REV)
            res[h] = c;
        else if(ROLE == SECOND_REV)
            res[h] = pk_avg<K + 1>(o, c);
        else if(K == 0)
            res[h] = pk_avg<1>(o, c);
        else
        {
            const unsigned t = __builtin_amdgcn_perm(0u, tw, h ? 0x0c030c02u : 0x0c010c00u);
            res[h] = pk_avg<K + 1>(pk_avg<K>(o, c), t);
        }
    }
    return __builtin_amdgcn_perm(res[1], res[0], 0x06040200u);
}

__device__ __forceinline__ void clk_begin(const Vol& V)
{
    if(V.clk && blockIdx.x == 0 && threadIdx.x == 0)
    {
        V.clk[0] = clock64();
        V.clk[1] = wall_clock64();
    }
}
__device__ __forceinline__ void clk_end(const Vol& V)
{
    if(V.clk && blockIdx.x == 0 && threadIdx.x == 0)
    {
        V.clk[2] = clock64();
        V.clk[3] = wall_clock64();
    }
}

// =====================================================================================================================
// design "ring": the production structure
// =====================================================================================================================
#define WPB 4
template <int K, int MODE>
__global__ void __launch_bounds__(128 * WPB) ring_kernel(Vol V)
{
    constexpr int PF = 8, NS = 4;
    clk_begin(V);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= WPB ? 1 : 0;
    const int a = (int)blockIdx.x * WPB + (wv - rev * WPB);
    const bool active = a < V.A;
    const int B = V.B;
    const int lane = threadIdx.x & 63;
    const unsigned offw = lane * 4;
    const int aa = active ? a : 0;
    const unsigned char* __restrict__ inCol = V.in + (long long)aa * V.strideA;
    unsigned char* outCol = V.out + (long long)aa * V.strideA;
    unsigned char* tmpCol = V.tmp + (long long)aa * V.strideA;
    const long long strideB = V.strideB;
    unsigned P[2], keepM[2], forceV[2];
    {
        const unsigned v = *reinterpret_cast<const unsigned*>(inCol + offw);
        P[0] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u);
        P[1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
#pragma unroll
        for(int r = 0; r < 2; ++r)
        {
            unsigned keep = 0, force = 0;
#pragma unroll
            for(int h = 0; h < 2; ++h)
            {
                const int z = lane * 4 + 2 * r + h;
                const bool border = (z == 0) || (z >= 255);
                keep |= (border ? 0u : 0xffffu) << (16 * h);
                force |= (border ? 255u : 0u) << (16 * h);
            }
            keepM[r] = keep;
            forceV[r] = force;
        }
    }
    const unsigned P1Pair = 10u * 0x00010001u;
    const float* __restrict__ p2col = V.p2 + (long long)aa * B;
    const long long dirStride = rev ? -strideB : strideB;

    auto walk = [&](auto roleTag, int ib0, int ib1) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(roleTag)::value;
        constexpr bool STORE_TMP = (ROLE == FIRST_REV) && (K > 0);
        constexpr bool LOAD_OUT = (ROLE == SECOND_FWD) || (ROLE == SECOND_REV) || (ROLE == FIRST_FWD && K > 0);
        constexpr bool LOAD_TMP = (ROLE == SECOND_FWD) && (K > 0);
        const int nSteps = ib1 - ib0;
        if(nSteps <= 0)
            return;
        const long long slice0 = rev ? (long long)(B - 1 - ib0) : (long long)ib0;
        unsigned inLoad = (unsigned)(slice0 * strideB);
        unsigned outStore = inLoad;
        int nLoaded = 0;
        unsigned rin[NS][PF], rout[NS][PF], rtmp[NS][PF];
        auto load_group = [&](unsigned (&ri)[PF], unsigned (&ro)[PF], unsigned (&rt)[PF]) __attribute__((always_inline)) {
#pragma unroll
            for(int t = 0; t < PF; ++t)
            {
                if(MODE & M_NOLOAD)
                {
                    ri[t] = 0x20402040u + inLoad;
                    ro[t] = 0x30303030u;
                    rt[t] = 0x50505050u;
                }
                else
                {
                    ri[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(inCol + (size_t)inLoad + offw));
                    if(LOAD_OUT)
                        ro[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(outCol + (size_t)inLoad + offw));
                    if(LOAD_TMP)
                        rt[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(tmpCol + (size_t)inLoad + offw));
                }
                const bool more = nLoaded + 1 < nSteps;
                inLoad += more ? (unsigned)dirStride : 0u;
                nLoaded += more ? 1 : 0;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto load_p2 = [&](int blk) __attribute__((always_inline)) -> float {
            const int ib = min(ib0 + blk * 64 + lane, ib1 - 1);
            return p2col[rev ? B - ib : ib];
        };
        unsigned ip2vec = 0;
        auto set_p2_block = [&](float v) __attribute__((always_inline)) { ip2vec = (unsigned)(int)floorf(v) * 0x00010001u; };
        auto step = [&](int i, unsigned inw, unsigned ow, unsigned tw) __attribute__((always_inline)) {
            unsigned neww;
            if(MODE & M_NOCOMPUTE)
                neww = inw ^ (LOAD_OUT ? ow : 0u) ^ (LOAD_TMP ? tw : 0u);
            else
            {
                unsigned q[2];
                const unsigned iP2Pair = (unsigned)__builtin_amdgcn_readlane((int)ip2vec, i & 63);
                lstep(P, inw, iP2Pair, P1Pair, keepM, forceV, q);
                neww = out_stage<ROLE, K>(q, ow, tw);
            }
            if(!(MODE & M_NOSTORE))
                __builtin_nontemporal_store(neww, reinterpret_cast<unsigned*>((STORE_TMP ? tmpCol : outCol) + (size_t)outStore + offw));
            else if(neww == 0x12345678u) // keep the arithmetic alive
                *reinterpret_cast<unsigned*>(outCol + offw) = neww;
            outStore += (unsigned)dirStride;
            __builtin_amdgcn_sched_barrier(0);
        };
        const int nGroups = (nSteps + PF - 1) / PF;
        auto group = [&](int g, unsigned (&ri)[PF], unsigned (&ro)[PF], unsigned (&rt)[PF]) __attribute__((always_inline)) {
            if(g * PF + PF <= nSteps)
            {
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    step(g * PF + t, ri[t], ro[t], rt[t]);
            }
            else
            {
#pragma unroll
                for(int t = 0; t < PF; ++t)
                    if(g * PF + t < nSteps)
                        step(g * PF + t, ri[t], ro[t], rt[t]);
            }
            load_group(ri, ro, rt);
        };
#pragma unroll
        for(int s = 0; s < NS; ++s)
            load_group(rin[s], rout[s], rtmp[s]);
        set_p2_block(load_p2(0));
        float p2next = load_p2(1);
        int G = 0;
        for(; G + NS <= nGroups; G += NS)
        {
            if(G > 0 && ((G * PF) & 63) == 0)
            {
                set_p2_block(p2next);
                p2next = load_p2((G * PF) / 64 + 1);
            }
#pragma unroll
            for(int s = 0; s < NS; ++s)
                group(G + s, rin[s], rout[s], rtmp[s]);
        }
        if(G < nGroups)
        {
            if(G > 0 && ((G * PF) & 63) == 0)
                set_p2_block(p2next);
#pragma unroll
            for(int s = 0; s < NS - 1; ++s)
                if(G + s < nGroups)
                    group(G + s, rin[s], rout[s], rtmp[s]);
        }
    };
    const int M = max(1, B / 2);
    if(active && B > 1)
    {
        if(!rev)
            walk(std::integral_constant<int, FIRST_FWD>{}, 1, M);
        else
            walk(std::integral_constant<int, FIRST_REV>{}, 1, B - M);
    }
    __syncthreads();
    if(active && B > 1)
    {
        if(!rev)
        {
            walk(std::integral_constant<int, SECOND_FWD>{}, M, B - 1);
            walk(std::integral_constant<int, FIRST_FWD>{}, B - 1, B);
        }
        else
            walk(std::integral_constant<int, SECOND_REV>{}, B - M, B);
    }
    clk_end(V);
}

// =====================================================================================================================
// design "lds": loader waves + LDS ring
//   workgroup = CPB columns: waves [0, CPB) forward chains, [CPB, 2 CPB) reverse chains, [2 CPB, 4 CPB) their loaders
//   one epoch = E steps; a chain's LDS ring holds D epochs of S streams; the loader runs D - 1 epochs ahead; one barrier per epoch
// =====================================================================================================================
#define CPB 4
#define EPOCH 8
#define SLOTS_PER_CHAIN 72 // slice buffers (256 B) per chain: 8 chains x 72 x 256 B = 144 KB of the 160 KB
template <int S>
struct RingCfg
{
    static constexpr int D = SLOTS_PER_CHAIN / (EPOCH * S); // S=1: 9, S=2: 4, S=3: 3
    static constexpr int LOADS_PER_EPOCH = S * (EPOCH / 4);
};
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

template <int K, int MODE>
__global__ void __launch_bounds__(256 * CPB) lds_kernel(Vol V)
{
    extern __shared__ unsigned char lds[];
    clk_begin(V);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool loader = wv >= 2 * CPB;
    const int chain = loader ? wv - 2 * CPB : wv; // 0 .. 2 CPB - 1
    const int rev = chain >= CPB ? 1 : 0;
    const int a = (int)blockIdx.x * CPB + (chain - rev * CPB);
    const bool active = a < V.A;
    const int B = V.B;
    const int lane = threadIdx.x & 63;
    const int aa = active ? a : 0;
    const unsigned char* __restrict__ inCol = V.in + (long long)aa * V.strideA;
    unsigned char* outCol = V.out + (long long)aa * V.strideA;
    unsigned char* tmpCol = V.tmp + (long long)aa * V.strideA;
    const long long strideB = V.strideB;
    const long long dirStride = rev ? -strideB : strideB;
    unsigned char* ring = lds + chain * (SLOTS_PER_CHAIN * 256);
    const int M = max(1, B / 2);

    // every wave of the workgroup runs the same number of epochs per phase (barriers must match)
    const int steps1 = max(M - 1, B - M - 1), steps2 = max(B - M, M); // phase 2 forward: SECOND_FWD (B-1-M) + the last slice
    const int epochs1 = (steps1 + EPOCH - 1) / EPOCH, epochs2 = (steps2 + EPOCH - 1) / EPOCH;

    unsigned P[2] = {0, 0}, keepM[2] = {0, 0}, forceV[2] = {0, 0};
    if(!loader)
    {
        const unsigned v = *reinterpret_cast<const unsigned*>(inCol + lane * 4);
        P[0] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u);
        P[1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
#pragma unroll
        for(int r = 0; r < 2; ++r)
        {
            unsigned keep = 0, force = 0;
#pragma unroll
            for(int h = 0; h < 2; ++h)
            {
                const int z = lane * 4 + 2 * r + h;
                const bool border = (z == 0) || (z >= 255);
                keep |= (border ? 0u : 0xffffu) << (16 * h);
                force |= (border ? 255u : 0u) << (16 * h);
            }
            keepM[r] = keep;
            forceV[r] = force;
        }
    }
    const unsigned P1Pair = 10u * 0x00010001u;
    const float* __restrict__ p2col = V.p2 + (long long)aa * B;

    // ---- loader side: epochs [0, nEpochs) of the walk ib0 <= ib < ib1 of my chain in the given role ----
    auto load_walk = [&](auto roleTag, int ib0, int ib1, int nEpochs) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(roleTag)::value;
        constexpr bool LOAD_OUT = (ROLE == SECOND_FWD) || (ROLE == SECOND_REV) || (ROLE == FIRST_FWD && K > 0);
        constexpr bool LOAD_TMP = (ROLE == SECOND_FWD) && (K > 0);
        constexpr int S = 1 + (LOAD_OUT ? 1 : 0) + (LOAD_TMP ? 1 : 0);
        constexpr int D = RingCfg<S>::D;
        const int nSteps = max(ib1 - ib0, 1);
        const long long slice0 = rev ? (long long)(B - 1 - ib0) : (long long)ib0;
        const int sub = lane >> 4;          // which of the 4 slices of a batch
        const unsigned col16 = (lane & 15) * 16;
        auto issue_epoch = [&](int e) __attribute__((always_inline)) {
            unsigned char* part = ring + (e % D) * (EPOCH * S * 256);
#pragma unroll
            for(int b = 0; b < EPOCH / 4; ++b)
            {
                const int st = min(e * EPOCH + b * 4 + sub, nSteps - 1); // past the end: re-read the last slice
                const long long off = (slice0 * strideB) + (long long)st * dirStride + col16;
                // LDS layout of a part: [step][stream][256 B]; a 16-byte direct load writes lane l at base + 16 l, i.e. 4 consecutive
                // steps need a step pitch of 256 B per stream -> [stream][step][256 B]
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(inCol + off),
                                                 (__attribute__((address_space(3))) void*)(part + (0 * EPOCH + b * 4) * 256), 16, 0, 2);
                if(LOAD_OUT)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(outCol + off),
                                                     (__attribute__((address_space(3))) void*)(part + (1 * EPOCH + b * 4) * 256), 16, 0, 2);
                if(LOAD_TMP)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tmpCol + off),
                                                     (__attribute__((address_space(3))) void*)(part + (2 * EPOCH + b * 4) * 256), 16, 0, 2);
            }
        };
        // prologue: D - 1 epochs in flight, epoch 0 landed
#pragma unroll
        for(int e = 0; e < D - 1; ++e)
            issue_epoch(e);
        wait_vmcnt<(D - 2) * RingCfg<S>::LOADS_PER_EPOCH>();
        // bare s_barrier: __syncthreads() carries workgroup fences, which make the compiler drain vmcnt to 0 — the loads in flight
        // are the point of this wave.  The data of an epoch is complete (vmcnt above) before the barrier that publishes it.
        __builtin_amdgcn_s_barrier(); // epoch 0 visible
        for(int e = 0; e < nEpochs; ++e)
        {
            issue_epoch(e + D - 1); // into the part consumed during epoch e - 1
            wait_vmcnt<(D - 2) * RingCfg<S>::LOADS_PER_EPOCH>(); // epoch e + 1 has landed
            __builtin_amdgcn_s_barrier(); // end of epoch e
        }
        wait_vmcnt<0>();
    };
    // ---- compute side ----
    // integer P2 pairs of an epoch come through the SCALAR cache (one s_load_dwordx8 per epoch, wave-uniform address): a vector load
    // here would share vmcnt with the stores and drain them
    const unsigned* __restrict__ p2i = reinterpret_cast<const unsigned*>(V.p2) + (long long)aa * B;
    auto compute_walk = [&](auto roleTag, int ib0, int ib1, int nEpochs) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(roleTag)::value;
        constexpr bool REV = (ROLE == FIRST_REV) || (ROLE == SECOND_REV);
        constexpr bool STORE_TMP = (ROLE == FIRST_REV) && (K > 0);
        constexpr bool LOAD_OUT = (ROLE == SECOND_FWD) || (ROLE == SECOND_REV) || (ROLE == FIRST_FWD && K > 0);
        constexpr bool LOAD_TMP = (ROLE == SECOND_FWD) && (K > 0);
        constexpr int S = 1 + (LOAD_OUT ? 1 : 0) + (LOAD_TMP ? 1 : 0);
        constexpr int D = RingCfg<S>::D;
        const int nSteps = active ? ib1 - ib0 : 0;
        const long long slice0 = REV ? (long long)(B - 1 - ib0) : (long long)ib0;
        unsigned outStore = (unsigned)(slice0 * strideB);
        unsigned char* dstCol = STORE_TMP ? tmpCol : outCol;
        const int nFull = nSteps / EPOCH;
        auto epoch = [&](auto fullTag, int e) __attribute__((always_inline)) {
            constexpr bool FULLE = decltype(fullTag)::value;
            const int i0 = __builtin_amdgcn_readfirstlane(ib0 + e * EPOCH);
            const unsigned* pe = p2i + (REV ? max(B - i0 - (EPOCH - 1), 0) : i0);
            // inline asm: the compiler cannot prove the map is never stored to, and would use a vector load (vmcnt) here
            typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
            u32x8 pv;
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(pv) : "s"(pe));
            const unsigned* part = reinterpret_cast<const unsigned*>(ring + (e % D) * (EPOCH * S * 256));
            unsigned inw[EPOCH], ow[EPOCH], tw[EPOCH];
#pragma unroll
            for(int t = 0; t < EPOCH; ++t)
            {
                inw[t] = part[(0 * EPOCH + t) * 64 + lane];
                ow[t] = LOAD_OUT ? part[(1 * EPOCH + t) * 64 + lane] : 0u;
                tw[t] = LOAD_TMP ? part[(2 * EPOCH + t) * 64 + lane] : 0u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pv)); // the scalar load above is invisible to the compiler's counters
#pragma unroll
            for(int t = 0; t < EPOCH; ++t)
            {
                if(FULLE || e * EPOCH + t < nSteps)
                {
                    unsigned neww;
                    if(MODE & M_NOCOMPUTE)
                        neww = inw[t] ^ ow[t] ^ tw[t];
                    else
                    {
                        unsigned q[2];
                        const unsigned iP2Pair = pv[REV ? EPOCH - 1 - t : t];
                        lstep(P, inw[t], iP2Pair, P1Pair, keepM, forceV, q);
                        neww = out_stage<ROLE, K>(q, ow[t], tw[t]);
                    }
                    if(!(MODE & M_NOSTORE))
                        __builtin_nontemporal_store(neww, reinterpret_cast<unsigned*>(dstCol + (size_t)outStore + lane * 4));
                    else if(neww == 0x12345678u)
                        *reinterpret_cast<unsigned*>(outCol + lane * 4) = neww;
                    outStore += (unsigned)dirStride;
                }
            }
        };
        __builtin_amdgcn_s_barrier(); // epoch 0 visible (bare barrier: my own stores in flight need not drain)
        int e = 0;
        for(; e < nFull; ++e)
        {
            epoch(std::true_type{}, e);
            __builtin_amdgcn_s_barrier(); // end of epoch e
        }
        for(; e < nEpochs; ++e)
        {
            if(e * EPOCH < nSteps)
                epoch(std::false_type{}, e);
            __builtin_amdgcn_s_barrier();
        }
    };

    if(loader)
    {
        if(!rev)
            load_walk(std::integral_constant<int, FIRST_FWD>{}, 1, M, epochs1);
        else
            load_walk(std::integral_constant<int, FIRST_REV>{}, 1, B - M, epochs1);
    }
    else
    {
        if(!rev)
            compute_walk(std::integral_constant<int, FIRST_FWD>{}, 1, M, epochs1);
        else
            compute_walk(std::integral_constant<int, FIRST_REV>{}, 1, B - M, epochs1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads(); // phase 1 stores visible to the whole workgroup
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if(loader)
    {
        if(!rev)
            load_walk(std::integral_constant<int, SECOND_FWD>{}, M, B, epochs2); // (the last slice's role differs only in the output stage)
        else
            load_walk(std::integral_constant<int, SECOND_REV>{}, B - M, B, epochs2);
    }
    else
    {
        if(!rev)
            compute_walk(std::integral_constant<int, SECOND_FWD>{}, M, B, epochs2);
        else
            compute_walk(std::integral_constant<int, SECOND_REV>{}, B - M, B, epochs2);
    }
    clk_end(V);
}

// ---------------------------------------------------------------------------------------------------------------------
struct Bench
{
    unsigned char *in, *out, *tmp;
    float* p2;
    unsigned* p2i;
    unsigned long long* clk;
    hipEvent_t e0, e1;
};

template <typename F>
static void timeit(Bench& Bn, const char* name, double algBytes, F launch)
{
    for(int i = 0; i < 2; ++i)
        launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(Bn.e0));
    const int reps = 5;
    for(int i = 0; i < reps; ++i)
        launch();
    CK(hipEventRecord(Bn.e1));
    CK(hipEventSynchronize(Bn.e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, Bn.e0, Bn.e1));
    unsigned long long c[4];
    CK(hipMemcpy(c, Bn.clk, sizeof(c), hipMemcpyDeviceToHost));
    const double mhz = (c[3] > c[1]) ? (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 100.0 : 0.0;
    printf("%-58s %8.1f us  %6.0f GB/s alg  frac %.3f  sclk~%5.0f MHz\n", name, ms / reps * 1e3, algBytes / (ms / reps) / 1e6,
           algBytes / (ms / reps) / 1e6 / 8000.0, mhz);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int X = 1000, Y = 750, Z = 256;
    const int XP = 1001; // padded row: odd number of 256-byte units
    const size_t bytes = (size_t)XP * Y * Z;
    Bench Bn;
    CK(hipMalloc(&Bn.in, bytes));
    CK(hipMalloc(&Bn.out, bytes));
    CK(hipMalloc(&Bn.tmp, bytes));
    CK(hipMalloc(&Bn.p2, (size_t)1024 * 1024 * 4));
    CK(hipMalloc(&Bn.p2i, (size_t)1024 * 1024 * 4));
    CK(hipMalloc(&Bn.clk, 64));
    CK(hipMemset(Bn.clk, 0, 64));
    {
        std::vector<unsigned char> h(bytes);
        unsigned s = 12345;
        for(size_t i = 0; i < bytes; ++i)
        {
            s = s * 1664525u + 1013904223u;
            h[i] = (unsigned char)(s >> 24);
        }
        CK(hipMemcpy(Bn.in, h.data(), bytes, hipMemcpyHostToDevice));
        CK(hipMemset(Bn.out, 7, bytes));
        CK(hipMemset(Bn.tmp, 9, bytes));
        std::vector<float> p(1024 * 1024);
        std::vector<unsigned> pint(1024 * 1024);
        for(size_t i = 0; i < p.size(); ++i)
        {
            s = s * 1664525u + 1013904223u;
            const unsigned ip = 80u + ((s >> 24) * 175u) / 256u; // floor(P2) in [80, 255)
            p[i] = (float)ip + 0.25f;
            pint[i] = ip * 0x00010001u;
        }
        CK(hipMemcpy(Bn.p2, p.data(), p.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Bn.p2i, pint.data(), pint.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipEventCreate(&Bn.e0));
    CK(hipEventCreate(&Bn.e1));
    const double alg0 = 5.0 * X * Y * Z + 32.0 * X * Y, alg2 = 6.0 * X * Y * Z + 32.0 * X * Y; // 11 B/voxel + 64 B/pixel over the two launches
    auto volY = [&](int xp) { return Vol{Bn.in, Bn.out, Bn.tmp, Bn.p2, (long long)Z, (long long)xp * Z, X, Y, Bn.clk}; };          // walk along Y
    auto volX = [&](int xp) { return Vol{Bn.in, Bn.out, Bn.tmp, Bn.p2, (long long)xp * Z, (long long)Z, Y, X, Bn.clk}; };          // walk along X
    const size_t ldsBytes = (size_t)2 * CPB * SLOTS_PER_CHAIN * 256;
#define RING(K, MODE, vol, name, alg)                                                                                                                        \
    timeit(Bn, name, alg, [&] {                                                                                                                               \
        Vol v = vol;                                                                                                                                          \
        hipLaunchKernelGGL((ring_kernel<K, MODE>), dim3((v.A + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, v);                                                    \
    })
#define LDSK(K, MODE, vol, name, alg)                                                                                                                        \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        CK(hipFuncSetAttribute((const void*)lds_kernel<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));                                 \
        timeit(Bn, name, alg, [&] {                                                                                                                           \
            Vol v = vol;                                                                                                                                      \
            v.p2 = reinterpret_cast<const float*>(Bn.p2i);                                                                                                    \
            hipLaunchKernelGGL((lds_kernel<K, MODE>), dim3((v.A + CPB - 1) / CPB), dim3(256 * CPB), ldsBytes, 0, v);                                          \
        });                                                                                                                                                   \
    } while(0)

    // functional cross-check of the two designs (same arithmetic, same inputs): identical output volumes
    {
        std::vector<unsigned char> a(bytes), b(bytes);
        auto run = [&](bool ldsDesign, int K, Vol v, std::vector<unsigned char>& dst) {
            CK(hipMemset(Bn.out, 7, bytes));
            CK(hipMemset(Bn.tmp, 9, bytes));
            if(ldsDesign)
            {
                v.p2 = reinterpret_cast<const float*>(Bn.p2i);
                if(K == 0)
                {
                    CK(hipFuncSetAttribute((const void*)lds_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                    hipLaunchKernelGGL((lds_kernel<0, 0>), dim3((v.A + CPB - 1) / CPB), dim3(256 * CPB), ldsBytes, 0, v);
                }
                else
                {
                    CK(hipFuncSetAttribute((const void*)lds_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
                    hipLaunchKernelGGL((lds_kernel<2, 0>), dim3((v.A + CPB - 1) / CPB), dim3(256 * CPB), ldsBytes, 0, v);
                }
            }
            else if(K == 0)
                hipLaunchKernelGGL((ring_kernel<0, 0>), dim3((v.A + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, v);
            else
                hipLaunchKernelGGL((ring_kernel<2, 0>), dim3((v.A + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, v);
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            CK(hipMemcpy(dst.data(), Bn.out, bytes, hipMemcpyDeviceToHost));
        };
        for(int K = 0; K <= 2; K += 2)
        {
            Vol v = K == 0 ? volY(X) : volX(X);
            run(false, K, v, a);
            run(true, K, v, b);
            size_t diff = 0;
            for(size_t i = 0; i < bytes; ++i)
                diff += a[i] != b[i];
            printf("check K=%d: ring vs lds design differ on %zu of %zu bytes\n", K, diff, bytes);
        }
        fflush(stdout);
    }
    RING(0, 0, volY(X), "ring  K=0 walk Y (production launch 1)", alg0);
    RING(2, 0, volX(X), "ring  K=2 walk X (production launch 2)", alg2);
    RING(2, 0, volX(XP), "ring  K=2 walk X, row pitch 1001 x 256", alg2);
    RING(0, 0, volY(XP), "ring  K=0 walk Y, row pitch 1001 x 256", alg0);
    RING(0, M_NOSTORE, volY(X), "ring  K=0 walk Y, no stores", alg0);
    RING(0, M_NOLOAD, volY(X), "ring  K=0 walk Y, no loads", alg0);
    RING(0, M_NOCOMPUTE, volY(X), "ring  K=0 walk Y, no arithmetic", alg0);
    RING(0, M_NOLOAD | M_NOSTORE, volY(X), "ring  K=0 walk Y, arithmetic only", alg0);
    RING(2, M_NOSTORE, volX(X), "ring  K=2 walk X, no stores", alg2);
    RING(2, M_NOLOAD, volX(X), "ring  K=2 walk X, no loads", alg2);
    RING(2, M_NOCOMPUTE, volX(X), "ring  K=2 walk X, no arithmetic", alg2);
    RING(2, M_NOCOMPUTE, volX(XP), "ring  K=2 walk X, no arithmetic, pitch 1001", alg2);
    RING(2, M_NOLOAD | M_NOSTORE, volX(X), "ring  K=2 walk X, arithmetic only", alg2);
    RING(2, 0, volY(X), "ring  K=2 roles on the Y walk (geometry vs roles)", alg2);
    RING(0, 0, volX(X), "ring  K=0 roles on the X walk", alg0);
    LDSK(0, 0, volY(X), "lds   K=0 walk Y", alg0);
    LDSK(2, 0, volX(X), "lds   K=2 walk X", alg2);
    LDSK(2, 0, volX(XP), "lds   K=2 walk X, row pitch 1001 x 256", alg2);
    LDSK(0, M_NOCOMPUTE, volY(X), "lds   K=0 walk Y, no arithmetic", alg0);
    LDSK(2, M_NOCOMPUTE, volX(X), "lds   K=2 walk X, no arithmetic", alg2);
    LDSK(0, M_NOSTORE, volY(X), "lds   K=0 walk Y, no stores", alg0);
    LDSK(2, M_NOSTORE, volX(X), "lds   K=2 walk X, no stores", alg2);
    return 0;
}
