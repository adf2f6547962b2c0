"""Per-axis launch time of the SGM path aggregation for several volume shapes in one process (HIP events of the library's own timer,
avdm_debug_sgm_kernel_timing): how the per-step time of a wave depends on the number of columns in flight and on the planes per wave.

    python scripts/sgm_axis_probe.py 1000x750x256 500x750x256 1000x376x256 1000x750x512 250x750x256

Prints, per shape: ms of the Y launch (X columns, Y steps) and of the X launch (Y columns, X steps), ns per step and wave, algorithmic GB/s."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alicevision_amd import abi
from alicevision_amd.pipeline import DevicePyramid
from alicevision_amd.synthetic import make_scene

lib = abi.load()
lib.avdm_debug_sgm_kernel_timing.argtypes = [C.c_int]
sc = make_scene(1, 4000, 3000, seed=3, device="cuda")
pyr = DevicePyramid(sc.images[0], 1, 128, abi.FILTER_CUDA_FIXED8)
sgm = abi.SgmParams.default()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
REPS = 6
for shape in (sys.argv[1:] or ["1000x750x256"]):
    X, Y, Z = (int(v) for v in shape.split("x"))
    vin = torch.randint(0, 255, (Y, X, Z), dtype=torch.uint8, device="cuda")
    vout = torch.empty_like(vin)
    tiles = (abi.SgmTile * 1)()
    tiles[0] = abi.SgmTile(vout.data_ptr(), vin.data_ptr(), X * Z, Z, Z, abi.ROI.make(0, X, 0, Y), C.pointer(pyr.desc))
    scratch = torch.empty(int(lib.avdm_volume_optimize_scratch_bytes(X, Y, Z)), dtype=torch.uint8, device="cuda")
    for _ in range(2):  # warm-up
        abi.check(lib.avdm_volume_optimize_tiles(1, tiles, C.c_void_p(scratch.data_ptr()), C.byref(sgm), st))
    torch.cuda.synchronize()
    lib.avdm_debug_sgm_kernel_timing(1)
    for _ in range(REPS):
        abi.check(lib.avdm_volume_optimize_tiles(1, tiles, C.c_void_p(scratch.data_ptr()), C.byref(sgm), st))
    torch.cuda.synchronize()
    k_ms, k_n = C.c_double(), C.c_long()
    lib.avdm_debug_sgm_kernel_timing_read(C.byref(k_ms), C.byref(k_n), 1)
    ms, n = (C.c_double * 4)(), (C.c_long * 4)()
    lib.avdm_debug_sgm_kernel_timing_read_paths(ms, n)
    lib.avdm_debug_sgm_kernel_timing(0)
    y_ms, x_ms = ms[0] / max(n[0], 1), ms[2] / max(n[2], 1)
    alg = 11.0 * X * Y * Z
    print(f"{shape}: Y launch {y_ms:.4f} ms ({X} columns, {1e6 * y_ms / Y:.0f} ns/step)  X launch {x_ms:.4f} ms ({Y} columns, {1e6 * x_ms / X:.0f} ns/step)  "
          f"both {y_ms + x_ms:.4f} ms = {alg / (y_ms + x_ms) / 1e6:.0f} GB/s algorithmic = {alg / (y_ms + x_ms) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
    del vin, vout, scratch
