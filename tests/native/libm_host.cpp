// tests/native/libm_host.cpp — alicevision_amd/csrc/avdm_libm.h compiled for the HOST, text unchanged, against the C library's own expf / cbrtf
// (tests/test_libm.py).  Returns the number of arguments on which the restatement and the C library differ in any bit.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../alicevision_amd/csrc/avdm_libm.h"

static inline uint32_t bits_of(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float float_of(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

extern "C" {

// every `stride`-th float bit pattern whose value lies in [lo, hi] (both signs are walked); first differing argument into *where
long libm_check_expf(float lo, float hi, unsigned stride, float* where)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
    for(long i = 0; i < (1L << 32); i += stride)
    {
        const float x = float_of((uint32_t)i);
        if(!(x >= lo && x <= hi))
            continue;
        if(bits_of(expf(x)) != bits_of(avdm::glibc::expf(x)))
        {
            if(bad == 0 && where)
                *where = x;
            bad += 1;
        }
    }
    return bad;
}

long libm_check_cbrtf(float lo, float hi, unsigned stride, float* where)
{
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
    for(long i = 1; i < 0x7f800000L; i += stride)
    {
        const float x = float_of((uint32_t)i);
        if(!(x >= lo && x <= hi))
            continue;
        if(bits_of(cbrtf(x)) != bits_of(avdm::glibc::cbrtf_pos(x)))
        {
            if(bad == 0 && where)
                *where = x;
            bad += 1;
        }
    }
    return bad;
}

float libm_expf(float x) { return avdm::glibc::expf(x); }
float libm_cbrtf(float x) { return avdm::glibc::cbrtf_pos(x); }
}
