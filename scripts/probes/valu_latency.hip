// VALU issue / dependency probe for gfx950: shader-clock cycles (s_memtime) per instruction of ONE wave64 for a chain of DEPENDENT instructions and for
// four INDEPENDENT chains, per instruction kind (what the eight-plane pass of avdm_similarity.hip is made of), with 1, 2 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip && ./valu_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>

#define S4(x) x x x x
#define S16(x) S4(x) S4(x) S4(x) S4(x)
#define S64(x) S16(x) S16(x) S16(x) S16(x)
#define LOOPS 64

typedef float v2f __attribute__((ext_vector_type(2)));

// DEP: 64 repetitions of a dependent group; IND: 16 repetitions of a group of independent chains
#define KERNEL(name, DEPSTR, INDSTR)                                                                                                     \
    __global__ void name##_dep(long long* out, float seed)                                                                               \
    {                                                                                                                                    \
        v2f a = {seed + threadIdx.x, seed * 0.5f}, b = {1.0001f, 0.9999f}, c = {0.001f, 0.002f};                                         \
        unsigned ua = (unsigned)threadIdx.x * 2654435761u, ub = 0x3c003c00u;                                                             \
        float f = seed + 1.5f, g = seed + 0.25f;                                                                                         \
        long long t0 = clock64(), w0 = wall_clock64();                                                                                   \
        _Pragma("unroll 1") for(int it = 0; it < LOOPS; ++it)                                                                            \
            asm volatile(S64(DEPSTR) : "+v"(a), "+v"(f), "+v"(ua), "+v"(g) : "v"(b), "v"(c), "v"(ub));                                   \
        long long t1 = clock64(), w1 = wall_clock64();                                                                                   \
        if((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&out[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&out[2], (unsigned long long)(w1 - w0)); }                                                                     \
        if(a.x + a.y + f + g + (float)ua == 12345.678f) out[1] = 1;                                                                      \
    }                                                                                                                                    \
    __global__ void name##_ind(long long* out, float seed)                                                                               \
    {                                                                                                                                    \
        v2f a = {seed + threadIdx.x, seed * 0.5f}, a2 = a * 1.1f, a3 = a * 1.2f, a4 = a * 1.3f, b = {1.0001f, 0.9999f}, c = {0.001f, 0.002f}; \
        unsigned ua = (unsigned)threadIdx.x * 2654435761u, ua2 = ua + 1, ua3 = ua + 2, ua4 = ua + 3, ub = 0x3c003c00u;                    \
        float f = seed + 1.5f, f2 = f + 1.f, f3 = f + 2.f, f4 = f + 3.f;                                                                 \
        long long t0 = clock64(), w0 = wall_clock64();                                                                                   \
        _Pragma("unroll 1") for(int it = 0; it < LOOPS; ++it)                                                                            \
            asm volatile(S16(INDSTR) : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(f), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(ua), "+v"(ua2), "+v"(ua3), "+v"(ua4) \
                         : "v"(b), "v"(c), "v"(ub));                                                                                     \
        long long t1 = clock64(), w1 = wall_clock64();                                                                                   \
        if((threadIdx.x & 63) == 0) { atomicMax((unsigned long long*)&out[0], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long*)&out[2], (unsigned long long)(w1 - w0)); }                                                                     \
        if(a.x + a2.x + a3.x + a4.y + f + f2 + f3 + f4 + (float)(ua + ua2 + ua3 + ua4) == 12345.678f) out[1] = 1;                        \
    }
// dep operands: %0 a (v2f), %1 f, %2 ua, %3 g, %4 b (v2f), %5 c (v2f), %6 ub
// ind operands: %0-%3 a..a4, %4-%7 f..f4, %8-%11 ua..ua4, %12 b, %13 c, %14 ub
KERNEL(pkfma, "v_pk_fma_f32 %0, %0, %4, %5\n", "v_pk_fma_f32 %0, %0, %12, %13\n v_pk_fma_f32 %1, %1, %12, %13\n v_pk_fma_f32 %2, %2, %12, %13\n v_pk_fma_f32 %3, %3, %12, %13\n")
KERNEL(pkadd, "v_pk_add_f32 %0, %0, %5\n", "v_pk_add_f32 %0, %0, %13\n v_pk_add_f32 %1, %1, %13\n v_pk_add_f32 %2, %2, %13\n v_pk_add_f32 %3, %3, %13\n")
KERNEL(fma32, "v_fma_f32 %1, %1, %6, %3\n", "v_fma_f32 %4, %4, %14, %8\n v_fma_f32 %5, %5, %14, %8\n v_fma_f32 %6, %6, %14, %8\n v_fma_f32 %7, %7, %14, %8\n")
KERNEL(floor32, "v_floor_f32 %1, %1\n", "v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n")
KERNEL(rcp32, "v_rcp_f32 %1, %1\n", "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL(sqrt32, "v_sqrt_f32 %1, %1\n", "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
KERNEL(exp32, "v_exp_f32 %1, %1\n", "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
KERNEL(rcp_fma, "v_rcp_f32 %1, %1\n s_nop 0\n v_fma_f32 %1, %1, %6, %3\n",
       "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n v_fma_f32 %4, %4, %14, %8\n v_fma_f32 %5, %5, %14, %8\n v_fma_f32 %6, %6, %14, %8\n v_fma_f32 %7, %7, %14, %8\n")
KERNEL(dot2, "v_dot2_f32_f16 %1, %2, %6, %1\n s_nop 2\n", "v_dot2_f32_f16 %4, %8, %14, %4\n v_dot2_f32_f16 %5, %9, %14, %5\n v_dot2_f32_f16 %6, %10, %14, %6\n v_dot2_f32_f16 %7, %11, %14, %7\n")
KERNEL(cvtpk, "v_cvt_pkrtz_f16_f32 %2, %1, %1\n v_cvt_f32_f16 %1, %2\n", "v_cvt_pkrtz_f16_f32 %8, %4, %4\n v_cvt_pkrtz_f16_f32 %9, %5, %5\n v_cvt_pkrtz_f16_f32 %10, %6, %6\n v_cvt_pkrtz_f16_f32 %11, %7, %7\n")
KERNEL(cvti, "v_cvt_i32_f32 %2, %1\n v_cvt_f32_i32 %1, %2\n", "v_cvt_i32_f32 %8, %4\n v_cvt_i32_f32 %9, %5\n v_cvt_i32_f32 %10, %6\n v_cvt_i32_f32 %11, %7\n")
// the sample loop's mix as one dependent group of 8 (1 transcendental in 8) and as 8 instructions on independent registers
KERNEL(mix8, "v_rcp_f32 %1, %1\n s_nop 0\n v_pk_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %6, %3\n v_floor_f32 %1, %1\n v_pk_add_f32 %0, %0, %5\n v_cvt_i32_f32 %2, %1\n v_pk_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %6, %3\n",
       "v_rcp_f32 %4, %4\n v_pk_fma_f32 %0, %0, %12, %13\n v_floor_f32 %5, %5\n v_pk_add_f32 %1, %1, %13\n v_cvt_i32_f32 %8, %6\n v_pk_fma_f32 %2, %2, %12, %13\n v_fma_f32 %7, %7, %14, %9\n v_pk_fma_f32 %3, %3, %12, %13\n")

// issue-slot probes: VALU interleaved with s_nop 0 / SALU / LDS reads (does a non-VALU instruction cost a wave an issue slot, and the SIMD a turn?)
KERNEL(valu_nop, "v_pk_fma_f32 %0, %0, %4, %5\n s_nop 0\n", "v_pk_fma_f32 %0, %0, %12, %13\n s_nop 0\n v_pk_fma_f32 %1, %1, %12, %13\n s_nop 0\n")
KERNEL(valu_salu, "v_pk_fma_f32 %0, %0, %4, %5\n s_add_u32 s50, s50, 0\n", "v_pk_fma_f32 %0, %0, %12, %13\n s_add_u32 s50, s50, 0\n v_pk_fma_f32 %1, %1, %12, %13\n s_add_u32 s51, s51, 0\n")
KERNEL(valu_lds, "v_pk_fma_f32 %0, %0, %4, %5\n ds_read_b32 %3, %2\n", "v_pk_fma_f32 %0, %0, %12, %13\n ds_read_b32 %4, %8\n v_pk_fma_f32 %1, %1, %12, %13\n ds_read_b32 %5, %8\n")

struct Entry { const char* name; void (*dep)(long long*, float); void (*ind)(long long*, float); int depI, indI; }; // instructions per repetition

int main()
{
    long long* d;
    hipMalloc((void**)&d, 32);
    Entry es[] = {
        {"v_pk_fma_f32", pkfma_dep, pkfma_ind, 1, 4}, {"v_pk_add_f32", pkadd_dep, pkadd_ind, 1, 4}, {"v_fma_f32", fma32_dep, fma32_ind, 1, 4},
        {"v_floor_f32", floor32_dep, floor32_ind, 1, 4}, {"v_rcp_f32", rcp32_dep, rcp32_ind, 1, 4}, {"v_sqrt_f32", sqrt32_dep, sqrt32_ind, 1, 4},
        {"v_exp_f32", exp32_dep, exp32_ind, 1, 4}, {"v_rcp_f32, s_nop 0, v_fma_f32", rcp_fma_dep, rcp_fma_ind, 2, 8},
        {"v_dot2_f32_f16 (dependent: + s_nop 2)", dot2_dep, dot2_ind, 1, 4}, {"v_cvt_pkrtz_f16_f32 / v_cvt_f32_f16", cvtpk_dep, cvtpk_ind, 2, 4},
        {"v_cvt_i32_f32 / v_cvt_f32_i32", cvti_dep, cvti_ind, 2, 4}, {"mix of 8 (1 rcp, 3 pk, fma, floor, cvt)", mix8_dep, mix8_ind, 8, 8},
        {"v_pk_fma_f32 + s_nop 0 (pairs)", valu_nop_dep, valu_nop_ind, 2, 4}, {"v_pk_fma_f32 + s_add_u32 (pairs)", valu_salu_dep, valu_salu_ind, 2, 4},
        {"v_pk_fma_f32 + ds_read_b32 (pairs)", valu_lds_dep, valu_lds_ind, 2, 4},
    };
    for(int waves = 1; waves <= 3; ++waves)
    {
        printf("== %d wave(s) per SIMD (one block of %d threads): shader cycles per INSTRUCTION of the SLOWEST wave - dependent chain | independent chains   [100 MHz ticks of the dependent run]\n", waves, 256 * waves);
        for(auto& e : es)
        {
            long long h[4];
            double dep = 1e30, ind = 1e30, wdep = 1e30;
            for(int it = 0; it < 5; ++it)
            {
                hipMemset(d, 0, 32);
                hipLaunchKernelGGL(e.dep, dim3(1), dim3(256 * waves), 0, 0, d, 1.0f);
                hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
                dep = std::min(dep, (double)h[0] / (64.0 * LOOPS * e.depI));
                wdep = std::min(wdep, (double)h[2]);
                hipMemset(d, 0, 32);
                hipLaunchKernelGGL(e.ind, dim3(1), dim3(256 * waves), 0, 0, d, 1.0f);
                hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
                ind = std::min(ind, (double)h[0] / (16.0 * LOOPS * e.indI));
            }
            printf("%-40s dependent %6.2f   independent %6.2f   [%.0f]\n", e.name, dep, ind, wdep);
        }
    }
    return 0;
}
