"""The two C-library functions of the pinned reference build as the DEVICE evaluates them (alicevision_amd/csrc/avdm_libm.h: glibc 2.35's expf and
cbrtf restated over IEEE double arithmetic) against the C library itself, bit for bit, on the CPU.  The header is compiled for the host text
unchanged; the device compiles the same text (avdm_literal.hip: the reference-arithmetic similarity kernels; avdm_image.hip: the Lab pyramid), and
the GPU suite closes the loop (tests/test_gpu_parity.py::test_pyramid_parity — texels identical to the oracle's, ::test_reference_arithmetic_* —
similarity volumes identical to the oracle's, which tests/test_oracle_ref.py pins to the reference's own code).

What the bit-equality rests on, and what would break it: the x86-64 C library runs the FMA build of expf on every CPU with FMA + AVX2 (this
image's hosts), in which the compiler fused all five multiply-adds of the routine; a host without FMA would run the unfused build, which differs
from it on ~1e-9 of the arguments — this test then fails HERE, on that host, before any GPU number is believed."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("libm") / "liblibm_host.so")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "libm_host.cpp"), "-o", out, "-lm"],
                   check=True)
    lib = C.CDLL(out)
    for f in (lib.libm_check_expf, lib.libm_check_cbrtf):
        f.argtypes = [C.c_float, C.c_float, C.c_uint, C.POINTER(C.c_float)]
        f.restype = C.c_long
    lib.libm_expf.argtypes = lib.libm_cbrtf.argtypes = [C.c_float]
    lib.libm_expf.restype = lib.libm_cbrtf.restype = C.c_float
    return lib


def test_expf_equals_the_c_library_bit_for_bit(libm):
    where = C.c_float(0.0)
    # every float of the range a Yoon-Kweon weight's exponent can take (-(dC / gammaC + dP / gammaP) <= 0, down to the underflow threshold and
    # beyond) and of the sigmoid's (|10 (x - mid) / width| <= ~30): 1.1e9 arguments, ~5 s on 8 cores
    assert libm.libm_check_expf(-110.0, 0.0, 1, C.byref(where)) == 0, where.value
    assert libm.libm_check_expf(0.0, 40.0, 1, C.byref(where)) == 0, where.value
    # the rest of the finite range up to the overflow threshold, every 3rd bit pattern
    assert libm.libm_check_expf(40.0, 90.0, 3, C.byref(where)) == 0, where.value
    assert libm.libm_expf(0.0) == 1.0 and libm.libm_expf(-200.0) == 0.0 and libm.libm_expf(100.0) == float("inf")


def test_cbrtf_equals_the_c_library_bit_for_bit(libm):
    where = C.c_float(0.0)
    # xyz2lab passes r in (216 / 24389, ~1.1]: every float of [2^-8, 4]
    assert libm.libm_check_cbrtf(2.0 ** -8, 4.0, 1, C.byref(where)) == 0, where.value
    # every 5th positive float, subnormals included
    assert libm.libm_check_cbrtf(0.0, 3.0e38, 5, C.byref(where)) == 0, where.value
    assert libm.libm_cbrtf(27.0) == 3.0
