// avdm_libm.h — the two libm functions of the pinned reference build, restated so that the device evaluates them to the SAME BITS.
//
// The oracle's pin (oracle/_ref: the reference's kernel layer compiled for the CPU) calls the C library for the two
// transcendental functions that reach the depth map through the similarity path:
//     expf   — CostYKfromLab (color.cuh:167-210: two per patch sample), sigmoid (matrix.cuh:334-337: Refine), with `__expf -> expf` in the shim
//     cbrtf  — xyz2lab (color.cuh:124-141: three per texel of the Lab pyramid)
// The device library (ocml) evaluates both to within 1 ulp of the C library, not to its bits; in the reference's arithmetic — unshifted fp32
// NCC sums whose difference is rounding noise in low-texture patches — that last bit re-draws winner-take-all flips (DESIGN.md section 2,
// "where the bar was missed").  Both functions have short, fully specified algorithms in glibc 2.35 (the C library of this image, here and
// on the GPU box); they are restated below in plain C++ over IEEE double arithmetic, one operation per line of the source they follow:
//     sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c   (the ARM optimized-routines expf: 32-entry table, cubic in double precision;
//                                                          the x86-64 build runs its FMA variant, sysdeps/x86_64/fpu/multiarch/e_expf.c,
//                                                          on every CPU with FMA + AVX2: all five multiply-adds below are fused)
//     sysdeps/ieee754/flt-32/s_cbrtf.c                   (quadratic start value, one Halley step in double precision, table of 2^(k/3))
// tests/test_libm.py compiles THIS text for the host and holds it to the C library's own expf / cbrtf bit for bit: expf on every float in
// [-104, 89] (the range a weight can take, and beyond), cbrtf on every positive normal float.
//
// Test infrastructure it is not: the pyramid kernel (avdm_image.hip) and the reference-arithmetic similarity kernels (avdm_literal.hip)
// call these in the product.  Nothing here reads or links oracle/.
#pragma once

#include <stdint.h>
#include <string.h>

#ifndef AVDM_LIBM_FN
#ifdef __HIPCC__
#define AVDM_LIBM_FN __host__ __device__ __forceinline__
#else
#define AVDM_LIBM_FN static inline
#endif
#endif

namespace avdm {
namespace glibc {

AVDM_LIBM_FN double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

// e_exp2f_data.c: tab[i] = bits(2^(i/32)) - (i << 52) / 32
#define AVDM_EXP2F_TAB                                                                                                                   \
    {                                                                                                                                    \
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,                \
          0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,              \
          0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,              \
          0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,              \
          0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,              \
          0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,              \
          0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull                                                                                     \
    }

// __expf (e_expf.c:36-101).  Outside [-103.97, 88.72] the C library returns 0 / +inf through its error paths; the same values here,
// selected after the straight-line evaluation (no branch in the sample loops that call this twice per patch sample); a NaN propagates
// through the arithmetic.  T = the 32-entry table (a kernel keeps a copy in LDS: the index differs from lane to lane).
AVDM_LIBM_FN float expf_tab(float x, const uint64_t* T)
{
#ifdef __clang__
#pragma clang fp contract(off) // only the multiply-adds spelled fma_d are fused
#endif
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = (double)x;
    // x * N / ln2 = k + r with r in [-1/2, 1/2] and k an integer
    // (GCC fuses BOTH uses of the product InvLn2N * xd in the FMA build: the rounding to an integer and the remainder, which is then exact)
    double kd = fma_d(InvLn2N, xd, SHIFT);
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    const double r = fma_d(InvLn2N, xd, -kd);
    // exp(x) = 2^(k/N) * 2^(r/N) ~= s * (C0 r^3 + C1 r^2 + C2 r + 1)
    uint64_t t = T[ki % 32u];
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    const double z = fma_d(C0, r, C1);
    const double r2 = r * r;
    double y = fma_d(C2, r, 1.0);
    y = fma_d(z, r2, y);
    y = y * s;
    float res = (float)y;
    res = x < -0x1.9fe368p6f ? 0.0f : res;            // x < log(0x1p-150)
    res = x > 0x1.62e42ep6f ? __builtin_inff() : res; // x > log(0x1p128)
    return res;
}
AVDM_LIBM_FN float expf(float x)
{
    const uint64_t T[32] = AVDM_EXP2F_TAB;
    return expf_tab(x, T);
}

// __cbrtf (s_cbrtf.c:36-63) for finite x > 0 (the only arguments xyz2lab passes: r > 216 / 24389)
AVDM_LIBM_FN float cbrtf_pos(float x)
{
#ifdef __clang__
#pragma clang fp contract(off) // the baseline x86-64 build of s_cbrtf.c has no fused operation
#endif
    const double CBRT2 = 1.2599210498948731648, SQR_CBRT2 = 1.5874010519681994748;
    const double factor[5] = {1.0 / SQR_CBRT2, 1.0 / CBRT2, 1.0, CBRT2, SQR_CBRT2};
    // frexpf: x = xm * 2^xe, xm in [0.5, 1)
    uint32_t u;
    memcpy(&u, &x, 4);
    int xe = (int)(u >> 23) - 126;
    if((u >> 23) == 0u)
    { // subnormal: normalise like frexpf does
        const float xs = x * 0x1p25f;
        memcpy(&u, &xs, 4);
        xe = (int)(u >> 23) - 126 - 25;
    }
    u = (u & 0x807fffffu) | 0x3f000000u;
    float xm;
    memcpy(&xm, &u, 4);
    const float uu = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = uu * uu * uu;
    const int rem = xe % 3; // C remainder: the sign of xe
    const float ym = (float)((double)uu * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * factor[2 + rem]);
    // ldexpf(ym, xe / 3): ym in [0.5, 1.6), |xe / 3| <= 50: exact scaling by a power of two
    const int e3 = xe / 3;
    uint32_t sb = (uint32_t)(127 + e3) << 23;
    float sc;
    memcpy(&sc, &sb, 4);
    return ym * sc;
}

} // namespace glibc
} // namespace avdm
