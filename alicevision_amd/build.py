"""Build the gfx950 shared library (alicevision_amd/csrc/libavdm.so) with hipcc — in-tree, no JIT cache.

    python -m alicevision_amd.build [--force]

Cross-compiles without a GPU.  Per-file flags matter: the integer-valued / order-sensitive stages are compiled with
-ffp-contract=off so they can be compared bit-exactly with the oracle (DESIGN.md "parity classes").
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libavdm.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(HERE, "..", "include")]
SOURCES = {
    "avdm_image.hip": ["-ffp-contract=off"],         # bit-exact class since round 6 (the Lab pyramid: glibc's cbrtf restated, no contraction)
    "avdm_similarity.hip": [],                       # fast-math intrinsics + FMA contraction allowed (tolerance class)
    "avdm_sgm.hip": ["-ffp-contract=off"],           # bit-exact class
    "avdm_maps.hip": ["-ffp-contract=off"],          # bit-exact / order-preserving class (the colour optimisation inside it: tolerance class, AVDM_OPT_FAST)
    "avdm_fuse.hip": ["-ffp-contract=off"],          # bit-exact class (double arithmetic in the reference's order)
    "avdm_jpeg.hip": ["-ffp-contract=off"],          # bit-exact class (integers only)
    "avdm_literal.hip": ["-ffp-contract=off"],       # the reference's similarity arithmetic as written: referenceArithmetic mode + AVDM_SIM_LITERAL=1
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "avdm_device.h"), os.path.join(CSRC, "avdm_knife.h"), os.path.join(CSRC, "avdm_libm.h"),
               os.path.join(HERE, "..", "include", "avdm.h")]
    extra_headers = {"avdm_fuse.hip": [os.path.join(HERE, "..", "include", "avdm_fuse.h")]}
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers + extra_headers.get(src, [])):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
