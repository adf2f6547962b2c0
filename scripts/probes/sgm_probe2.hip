// sgm_probe2 — second SGM probe (round 2): what do (a) scalar-base addressing and (b) a different split of the four paths over the two
// launches buy for ONE cfg3 volume?  Re-uses the helpers and the "ring" baseline of sgm_probe.hip.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/sgm_probe2.hip -o scripts/probes/sgm_probe2
// Designs:
//   ring   : production structure (sgm_probe.hip): launch 1 = Y pair (K = 0), launch 2 = X pair (K = 2)
//   ring-s : the same two launches with wave-uniform base pointers (SGPR base + 32-bit lane offset: no 64-bit VALU address arithmetic,
//            the walk counters stay on the scalar unit)
//   merged : launch 1 = Y pair (K = 0)  +  first halves of the two X paths, which only read the input volume and leave their raw costs
//            in the scratch volume (3500 waves instead of 2000);  launch 2 = second halves of the X paths (in, out, scratch -> out).
//            Same 11 B/voxel as the production split (5 + 2 in launch 1, 4 in launch 2).
#define main sgm_probe_main
#include "sgm_probe.hip"
#undef main

// n div 3 for packed uint16 pairs, n <= 1020: (n * 21856) >> 16 == (n * 683) >> 11; the two products by SDWA word selects, the two
// high halves gathered by one v_perm_b32
__device__ __forceinline__ unsigned pk_div3_sdwa(unsigned n)
{
    unsigned lo, hi;
    const unsigned mul = 21856u;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(lo) : "v"(n), "v"(mul));
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(hi) : "v"(n), "v"(mul));
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
template <int K>
__device__ __forceinline__ unsigned pk_avg2(unsigned o, unsigned c)
{
    if(K == 0)
        return c;
    const unsigned n = as_u32(as_pk(o) * (unsigned short)K + as_pk(c));
    if(K == 1)
        return as_u32(as_pk(n) >> (unsigned short)1);
    if(K == 3)
        return as_u32(as_pk(n) >> (unsigned short)2);
    return pk_div3_sdwa(n);
}

// roles of a walk: which volumes it reads besides the input, what it computes, where it stores
enum
{
    R_STORE_OUT = 0, // out = L                                   (Y pair, phase 1, both directions; last slice of the Y forward path)
    R_AVG1_OUT,      // out = (out + L) >> 1                       (Y pair, phase 2)
    R_RAW_TMP,       // tmp = L                                   (X paths, first halves)
    R_FIN_F,         // out = avg3(avg2(out, L), tmp)              (X forward path, second half)
    R_FIN_R,         // out = avg3(avg2(out, tmp), L)              (X reverse path, second half)
    R_LAST_F,        // out = avg2(out, L)                         (X forward path, slice B-1)
    R_LAST_R,        // out = avg3(255, L)                         (X reverse path, slice 0)
    R_K2_FIRST_FWD,  // production K = 2 roles (ring-s)
    R_K2_FIRST_REV,
    R_K2_SECOND_FWD,
    R_K2_SECOND_REV
};

template <int ROLE>
__device__ __forceinline__ unsigned out_stage2(const unsigned (&q)[2], unsigned ow, unsigned tw)
{
    unsigned res[2];
#pragma unroll
    for(int h = 0; h < 2; ++h)
    {
        const unsigned c = q[h];
        const unsigned o = __builtin_amdgcn_perm(0u, ow, h ? 0x0c030c02u : 0x0c010c00u);
        const unsigned t = __builtin_amdgcn_perm(0u, tw, h ? 0x0c030c02u : 0x0c010c00u);
        if(ROLE == R_STORE_OUT || ROLE == R_RAW_TMP || ROLE == R_K2_FIRST_REV)
            res[h] = c;
        else if(ROLE == R_AVG1_OUT)
            res[h] = pk_avg2<1>(o, c);
        else if(ROLE == R_FIN_F || ROLE == R_K2_SECOND_FWD)
            res[h] = pk_avg2<3>(pk_avg2<2>(o, c), t);
        else if(ROLE == R_FIN_R)
            res[h] = pk_avg2<3>(pk_avg2<2>(o, t), c);
        else if(ROLE == R_LAST_F || ROLE == R_K2_FIRST_FWD)
            res[h] = pk_avg2<2>(o, c);
        else if(ROLE == R_K2_SECOND_REV)
            res[h] = pk_avg2<3>(o, c);
        else
            res[h] = pk_avg2<3>(0x00ff00ffu, c);
    }
    return __builtin_amdgcn_perm(res[1], res[0], 0x06040200u);
}

struct LaneState
{
    unsigned P[2], keepM[2], forceV[2];
    unsigned laneOff;
    int lane;
};

__device__ __forceinline__ void lane_init(LaneState& S, const unsigned char* inCol)
{
    S.lane = threadIdx.x & 63;
    S.laneOff = (unsigned)S.lane * 4u;
    const unsigned v = *reinterpret_cast<const unsigned*>(inCol + S.laneOff);
    S.P[0] = __builtin_amdgcn_perm(0u, v, 0x0c010c00u);
    S.P[1] = __builtin_amdgcn_perm(0u, v, 0x0c030c02u);
#pragma unroll
    for(int r = 0; r < 2; ++r)
    {
        unsigned keep = 0, force = 0;
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
            const int z = S.lane * 4 + 2 * r + h;
            const bool border = (z == 0) || (z >= 255);
            keep |= (border ? 0u : 0xffffu) << (16 * h);
            force |= (border ? 255u : 0u) << (16 * h);
        }
        S.keepM[r] = keep;
        S.forceV[r] = force;
    }
}

// One walk of nSteps steps starting at slice `slice0` (signed stride dstride between steps), all base pointers wave-uniform.
// p2col: this column's P2 row; p2i0 / p2dir: map index of step 0 and its direction.
template <int ROLE>
__device__ __forceinline__ void walk_s(LaneState& S, const unsigned char* inB, unsigned char* outB, unsigned char* tmpB, long long dstride, int nSteps,
                                       const float* __restrict__ p2col, int p2i0, int p2dir, int p2max)
{
    constexpr int PF = 8, NS = 4;
    constexpr bool LOAD_OUT = (ROLE == R_AVG1_OUT) || (ROLE == R_FIN_F) || (ROLE == R_FIN_R) || (ROLE == R_LAST_F) || (ROLE == R_K2_FIRST_FWD) ||
                              (ROLE == R_K2_SECOND_FWD) || (ROLE == R_K2_SECOND_REV);
    constexpr bool LOAD_TMP = (ROLE == R_FIN_F) || (ROLE == R_FIN_R) || (ROLE == R_K2_SECOND_FWD);
    constexpr bool STORE_TMP = (ROLE == R_RAW_TMP) || (ROLE == R_K2_FIRST_REV);
    if(nSteps <= 0)
        return;
    const unsigned char* inL = inB;
    const unsigned char* outL = outB;
    const unsigned char* tmpL = tmpB;
    unsigned char* stB = STORE_TMP ? tmpB : outB;
    int nLoaded = 0;
    unsigned rin[NS][PF], rout[NS][PF], rtmp[NS][PF];
    auto load_group = [&](unsigned (&ri)[PF], unsigned (&ro)[PF], unsigned (&rt)[PF]) __attribute__((always_inline)) {
#pragma unroll
        for(int t = 0; t < PF; ++t)
        {
            // the 32-bit lane offset is made opaque HERE so that its zero extension stays in this block: instruction selection then sees
            // (uniform base + zext(i32)) and uses the SGPR-base form of global_load (no 64-bit VALU add per access)
            unsigned lo = S.laneOff;
            asm volatile("" : "+v"(lo));
            ri[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(inL + lo));
            if(LOAD_OUT)
                ro[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(outL + lo));
            if(LOAD_TMP)
                rt[t] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(tmpL + lo));
            const long long adv = (nLoaded + 1 < nSteps) ? dstride : 0ll; // past the end: keep re-reading the last slice
            inL += adv;
            outL += adv;
            tmpL += adv;
            ++nLoaded;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_p2 = [&](int blk) __attribute__((always_inline)) -> float {
        const int i = min(blk * 64 + S.lane, nSteps - 1);
        return p2col[min(max(p2i0 + p2dir * i, 0), p2max)];
    };
    unsigned ip2vec = 0;
    auto set_p2_block = [&](float v) __attribute__((always_inline)) { ip2vec = (unsigned)(int)floorf(v) * 0x00010001u; };
    auto step = [&](int i, unsigned inw, unsigned ow, unsigned tw) __attribute__((always_inline)) {
        unsigned q[2];
        const unsigned iP2Pair = (unsigned)__builtin_amdgcn_readlane((int)ip2vec, i & 63);
        lstep(S.P, inw, iP2Pair, 10u * 0x00010001u, S.keepM, S.forceV, q);
        const unsigned neww = out_stage2<ROLE>(q, ow, tw);
        unsigned lo = S.laneOff;
        asm volatile("" : "+v"(lo));
        __builtin_nontemporal_store(neww, reinterpret_cast<unsigned*>(stB + lo));
        stB += dstride;
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
}

struct Vol2
{
    Vol y, x;            // the Y walk (columns = x positions) and the X walk (columns = y positions) of the same volume
    unsigned* state;     // [x.A][2][64][2] path costs of the X paths at the cut
    int nWgY;            // workgroups of the Y pair in the merged launch
};

// Y pair (K = 0) of column a by the 2 * WPB waves of a workgroup (production structure, scalar bases)
__device__ __forceinline__ void y_pair_body(const Vol& V, int wgIdx)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= WPB ? 1 : 0;
    const int a = wgIdx * WPB + (wv - rev * WPB);
    const bool active = a < V.A;
    const int B = V.B;
    const int aa = active ? a : 0;
    const unsigned char* inCol = V.in + (long long)aa * V.strideA;
    unsigned char* outCol = V.out + (long long)aa * V.strideA;
    const long long sB = V.strideB;
    const float* p2col = V.p2 + (long long)aa * B;
    LaneState S;
    lane_init(S, inCol);
    const int M = max(1, B / 2);
    if(active && B > 1)
    {
        if(!rev)
        {
            *reinterpret_cast<unsigned*>(outCol + S.laneOff) = 0xffffffffu; // slice 0
            walk_s<R_STORE_OUT>(S, inCol + sB, outCol + sB, nullptr, sB, M - 1, p2col, 1, 1, B - 1);
        }
        else
            walk_s<R_STORE_OUT>(S, inCol + (long long)(B - 2) * sB, outCol + (long long)(B - 2) * sB, nullptr, -sB, B - M - 1, p2col, B - 1, -1, B);
    }
    __syncthreads();
    if(active && B > 1)
    {
        if(!rev)
        {
            walk_s<R_AVG1_OUT>(S, inCol + (long long)M * sB, outCol + (long long)M * sB, nullptr, sB, B - 1 - M, p2col, M, 1, B);
            walk_s<R_STORE_OUT>(S, inCol + (long long)(B - 1) * sB, outCol + (long long)(B - 1) * sB, nullptr, sB, 1, p2col, B - 1, 1, B);
        }
        else
            walk_s<R_AVG1_OUT>(S, inCol + (long long)(M - 1) * sB, outCol + (long long)(M - 1) * sB, nullptr, -sB, M, p2col, M, -1, B);
    }
}

// production K = 2 pair with scalar bases (ring-s)
__device__ __forceinline__ void x_pair_body(const Vol& V, int wgIdx)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= WPB ? 1 : 0;
    const int a = wgIdx * WPB + (wv - rev * WPB);
    const bool active = a < V.A;
    const int B = V.B;
    const int aa = active ? a : 0;
    const unsigned char* inCol = V.in + (long long)aa * V.strideA;
    unsigned char* outCol = V.out + (long long)aa * V.strideA;
    unsigned char* tmpCol = V.tmp + (long long)aa * V.strideA;
    const long long sB = V.strideB;
    const float* p2col = V.p2 + (long long)aa * B;
    LaneState S;
    lane_init(S, inCol);
    const int M = max(1, B / 2);
    if(active && B > 1)
    {
        if(!rev)
        {
            *reinterpret_cast<unsigned*>(outCol + S.laneOff) = 0xffffffffu;
            walk_s<R_K2_FIRST_FWD>(S, inCol + sB, outCol + sB, tmpCol + sB, sB, M - 1, p2col, 1, 1, B - 1);
        }
        else
            walk_s<R_K2_FIRST_REV>(S, inCol + (long long)(B - 2) * sB, outCol + (long long)(B - 2) * sB, tmpCol + (long long)(B - 2) * sB, -sB, B - M - 1, p2col,
                                   B - 1, -1, B);
    }
    __syncthreads();
    if(active && B > 1)
    {
        if(!rev)
        {
            walk_s<R_K2_SECOND_FWD>(S, inCol + (long long)M * sB, outCol + (long long)M * sB, tmpCol + (long long)M * sB, sB, B - 1 - M, p2col, M, 1, B);
            walk_s<R_K2_FIRST_FWD>(S, inCol + (long long)(B - 1) * sB, outCol + (long long)(B - 1) * sB, nullptr, sB, 1, p2col, B - 1, 1, B);
        }
        else
            walk_s<R_K2_SECOND_REV>(S, inCol + (long long)(M - 1) * sB, outCol + (long long)(M - 1) * sB, nullptr, -sB, M, p2col, M, -1, B);
    }
}

// first halves of the X paths: raw costs into the scratch volume, path costs at the cut into `state`
__device__ __forceinline__ void x_first_body(const Vol& V, unsigned* state, int wgIdx)
{
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= WPB ? 1 : 0;
    const int a = wgIdx * WPB + (wv - rev * WPB);
    if(a >= V.A || V.B <= 1)
        return;
    const int B = V.B;
    const unsigned char* inCol = V.in + (long long)a * V.strideA;
    unsigned char* tmpCol = V.tmp + (long long)a * V.strideA;
    const long long sB = V.strideB;
    const float* p2col = V.p2 + (long long)a * B;
    LaneState S;
    lane_init(S, inCol);
    const int M = max(1, B / 2);
    if(!rev)
        walk_s<R_RAW_TMP>(S, inCol + sB, nullptr, tmpCol + sB, sB, M - 1, p2col, 1, 1, B - 1); // slices 1 .. M-1
    else
        walk_s<R_RAW_TMP>(S, inCol + (long long)(B - 2) * sB, nullptr, tmpCol + (long long)(B - 2) * sB, -sB, B - M - 1, p2col, B - 1, -1, B); // B-2 .. M
    unsigned* st = state + ((long long)a * 2 + rev) * 128 + S.lane * 2;
    st[0] = S.P[0];
    st[1] = S.P[1];
}

template <int DUMMY>
__global__ void __launch_bounds__(128 * WPB) merged1_kernel(Vol2 V)
{
    if((int)blockIdx.x < V.nWgY)
        y_pair_body(V.y, (int)blockIdx.x);
    else
        x_first_body(V.x, V.state, (int)blockIdx.x - V.nWgY);
}

// second halves of the X paths
template <int WPB2>
__global__ void __launch_bounds__(128 * WPB2) merged2_kernel(Vol2 V2)
{
    const Vol& V = V2.x;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rev = wv >= WPB2 ? 1 : 0;
    const int a = (int)blockIdx.x * WPB2 + (wv - rev * WPB2);
    if(a >= V.A || V.B <= 1)
        return;
    const int B = V.B;
    const unsigned char* inCol = V.in + (long long)a * V.strideA;
    unsigned char* outCol = V.out + (long long)a * V.strideA;
    unsigned char* tmpCol = V.tmp + (long long)a * V.strideA;
    const long long sB = V.strideB;
    const float* p2col = V.p2 + (long long)a * B;
    LaneState S;
    lane_init(S, inCol);
    const unsigned* st = V2.state + ((long long)a * 2 + rev) * 128 + S.lane * 2;
    S.P[0] = st[0];
    S.P[1] = st[1];
    const int M = max(1, B / 2);
    if(!rev)
    {
        walk_s<R_FIN_F>(S, inCol + (long long)M * sB, outCol + (long long)M * sB, tmpCol + (long long)M * sB, sB, B - 1 - M, p2col, M, 1, B); // M .. B-2
        walk_s<R_LAST_F>(S, inCol + (long long)(B - 1) * sB, outCol + (long long)(B - 1) * sB, nullptr, sB, 1, p2col, B - 1, 1, B);
    }
    else
    {
        walk_s<R_FIN_R>(S, inCol + (long long)(M - 1) * sB, outCol + (long long)(M - 1) * sB, tmpCol + (long long)(M - 1) * sB, -sB, M - 1, p2col, M, -1, B); // M-1 .. 1
        walk_s<R_LAST_R>(S, inCol, outCol, nullptr, -sB, 1, p2col, 1, -1, B); // slice 0
    }
}

__global__ void __launch_bounds__(128 * WPB) rings_y_kernel(Vol V) { y_pair_body(V, (int)blockIdx.x); }
__global__ void __launch_bounds__(128 * WPB) rings_x_kernel(Vol V) { x_pair_body(V, (int)blockIdx.x); }

int main(int argc, char** argv)
{
    const int X = 1000, Y = 750, Z = 256;
    const size_t bytes = (size_t)X * Y * Z;
    Bench Bn;
    CK(hipMalloc(&Bn.in, bytes));
    CK(hipMalloc(&Bn.out, bytes));
    CK(hipMalloc(&Bn.tmp, bytes));
    CK(hipMalloc(&Bn.p2, (size_t)1024 * 1024 * 4));
    CK(hipMalloc(&Bn.clk, 64));
    CK(hipMemset(Bn.clk, 0, 64));
    unsigned* state;
    CK(hipMalloc(&state, (size_t)1024 * 2 * 128 * 4));
    {
        std::vector<unsigned char> h(bytes);
        unsigned s = 12345;
        for(size_t i = 0; i < bytes; ++i)
        {
            s = s * 1664525u + 1013904223u;
            h[i] = (unsigned char)(s >> 24);
        }
        CK(hipMemcpy(Bn.in, h.data(), bytes, hipMemcpyHostToDevice));
        std::vector<float> p(1024 * 1024);
        for(size_t i = 0; i < p.size(); ++i)
        {
            s = s * 1664525u + 1013904223u;
            p[i] = (float)(80u + ((s >> 24) * 175u) / 256u) + 0.25f;
        }
        CK(hipMemcpy(Bn.p2, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipEventCreate(&Bn.e0));
    CK(hipEventCreate(&Bn.e1));
    const double alg0 = 5.0 * X * Y * Z + 32.0 * X * Y, alg2 = 6.0 * X * Y * Z + 32.0 * X * Y;
    const double algM1 = 7.0 * X * Y * Z + 48.0 * X * Y, algM2 = 4.0 * X * Y * Z + 16.0 * X * Y;
    const Vol vY{Bn.in, Bn.out, Bn.tmp, Bn.p2, (long long)Z, (long long)X * Z, X, Y, nullptr};
    const Vol vX{Bn.in, Bn.out, Bn.tmp, Bn.p2, (long long)X * Z, (long long)Z, Y, X, nullptr};
    Vol2 v2{vY, vX, state, (X + WPB - 1) / WPB};
    const int nWgX = (Y + WPB - 1) / WPB;

    auto run_ring = [&] {
        hipLaunchKernelGGL((ring_kernel<0, 0>), dim3((X + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, vY);
        hipLaunchKernelGGL((ring_kernel<2, 0>), dim3(nWgX), dim3(128 * WPB), 0, 0, vX);
    };
    auto run_rings = [&] {
        hipLaunchKernelGGL(rings_y_kernel, dim3((X + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, vY);
        hipLaunchKernelGGL(rings_x_kernel, dim3(nWgX), dim3(128 * WPB), 0, 0, vX);
    };
    auto run_merged1 = [&] { hipLaunchKernelGGL((merged1_kernel<0>), dim3(v2.nWgY + nWgX), dim3(128 * WPB), 0, 0, v2); };
    auto run_merged2_4 = [&] { hipLaunchKernelGGL((merged2_kernel<4>), dim3((Y + 3) / 4), dim3(128 * 4), 0, 0, v2); };
    auto run_merged2_2 = [&] { hipLaunchKernelGGL((merged2_kernel<2>), dim3((Y + 1) / 2), dim3(128 * 2), 0, 0, v2); };
    auto run_merged2_1 = [&] { hipLaunchKernelGGL((merged2_kernel<1>), dim3(Y), dim3(128), 0, 0, v2); };

    // cross-check: the three designs produce the same output volume
    {
        std::vector<unsigned char> a(bytes), b(bytes);
        auto result = [&](auto run, std::vector<unsigned char>& dst) {
            CK(hipMemset(Bn.out, 7, bytes));
            CK(hipMemset(Bn.tmp, 9, bytes));
            run();
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            CK(hipMemcpy(dst.data(), Bn.out, bytes, hipMemcpyDeviceToHost));
        };
        result(run_ring, a);
        result(run_rings, b);
        size_t diff = 0;
        for(size_t i = 0; i < bytes; ++i)
            diff += a[i] != b[i];
        printf("check: ring vs ring-s differ on %zu of %zu bytes\n", diff, bytes);
        result([&] { run_merged1(); run_merged2_4(); }, b);
        diff = 0;
        for(size_t i = 0; i < bytes; ++i)
            diff += a[i] != b[i];
        printf("check: ring vs merged differ on %zu of %zu bytes\n", diff, bytes);
        fflush(stdout);
    }
    timeit(Bn, "ring    launch 1 (Y pair, K=0)", alg0, [&] { hipLaunchKernelGGL((ring_kernel<0, 0>), dim3((X + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, vY); });
    timeit(Bn, "ring    launch 2 (X pair, K=2)", alg2, [&] { hipLaunchKernelGGL((ring_kernel<2, 0>), dim3(nWgX), dim3(128 * WPB), 0, 0, vX); });
    timeit(Bn, "ring    both launches", alg0 + alg2, run_ring);
    timeit(Bn, "ring-s  launch 1 (Y pair, K=0)", alg0, [&] { hipLaunchKernelGGL(rings_y_kernel, dim3((X + WPB - 1) / WPB), dim3(128 * WPB), 0, 0, vY); });
    timeit(Bn, "ring-s  launch 2 (X pair, K=2)", alg2, [&] { hipLaunchKernelGGL(rings_x_kernel, dim3(nWgX), dim3(128 * WPB), 0, 0, vX); });
    timeit(Bn, "ring-s  both launches", alg0 + alg2, run_rings);
    timeit(Bn, "merged  launch 1 (Y pair + X first halves)", algM1, run_merged1);
    timeit(Bn, "merged  launch 2 (X second halves), 4 columns / workgroup", algM2, run_merged2_4);
    timeit(Bn, "merged  launch 2 (X second halves), 2 columns / workgroup", algM2, run_merged2_2);
    timeit(Bn, "merged  launch 2 (X second halves), 1 column / workgroup", algM2, run_merged2_1);
    timeit(Bn, "merged  both launches (4 columns / workgroup)", algM1 + algM2, [&] { run_merged1(); run_merged2_4(); });
    timeit(Bn, "merged  both launches (1 column / workgroup)", algM1 + algM2, [&] { run_merged1(); run_merged2_1(); });
    return 0;
}
