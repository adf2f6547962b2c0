/* STAND-IN for CUDA's cuda_fp16.h (oracle/_ref test infrastructure; see cuda_runtime.h in this directory):
 * IEEE binary16 storage with round-to-nearest-even conversions; arithmetic goes through float like __hadd on the device
 * (one rounding of the exact float sum, which is exact for two halfs). */
#ifndef AVDM_REF_SHIM_CUDA_FP16_H
#define AVDM_REF_SHIM_CUDA_FP16_H
#include <cstdint>
#include <cstring>
namespace shim {
static inline uint16_t f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if(x >= 0x7f800000u)
        return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if(x >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if(x < 0x33000001u) /* < 2^-25 (or == 2^-25: ties to even -> 0) */
        return (uint16_t)sign;
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t he;
    if(e < -14)
    {
        shift = 13 + (-14 - e);
        he = 0;
    }
    else
    {
        shift = 13;
        he = (uint32_t)(e + 15);
    }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if(rem > half || (rem == half && (hm & 1u)))
        ++hm;
    uint32_t h = (e < -14) ? hm : ((he << 10) + (hm - 0x400u)); /* mantissa carry propagates into the exponent */
    return (uint16_t)(sign | h);
}
static inline float f16_to_f32(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if(e == 0)
    {
        if(m == 0)
            x = sign;
        else
        {
            int s = 0;
            while(!(m & 0x400u))
            {
                m <<= 1;
                ++s;
            }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 + 1 - s) << 23) | (m << 13);
        }
    }
    else if(e == 31)
        x = sign | 0x7f800000u | (m << 13);
    else
        x = sign | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
}
struct __half
{
    uint16_t bits;
    __half() = default;
    __half(float f) : bits(shim::f32_to_f16(f)) {}
    __half(double f) : bits(shim::f32_to_f16((float)f)) {}
    __half(int f) : bits(shim::f32_to_f16((float)f)) {}
    operator float() const { return shim::f16_to_f32(bits); }
};
static inline __half __float2half(float f) { return __half(f); }
static inline float __half2float(__half h) { return shim::f16_to_f32(h.bits); }
static inline __half __hadd(__half a, __half b) { return __half(__half2float(a) + __half2float(b)); }
#endif
