"""SGM path-aggregation micro-benchmark: n independent cfg3-sized volumes (1000 x 750 x 256) in ONE batched call.
Prints ms per call and algorithmic GB/s (11 B/voxel) — shows how the column-parallel recurrence scales with the batch size."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alicevision_amd import abi
from alicevision_amd.pipeline import DevicePyramid
from alicevision_amd.synthetic import make_scene

X, Y, Z = int(os.environ.get("SGM_X", 1000)), int(os.environ.get("SGM_Y", 750)), int(os.environ.get("SGM_Z", 256))
lib = abi.load()
sc = make_scene(1, 4000, 3000, seed=3, device="cuda")
pyr = DevicePyramid(sc.images[0], 1, 128, abi.FILTER_CUDA_FIXED8)
sgm = abi.SgmParams.default()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for n in [int(v) for v in (sys.argv[1:] or ["1", "2", "4"])]:
    vin = [torch.randint(0, 255, (Y, X, Z), dtype=torch.uint8, device="cuda") for _ in range(n)]
    vout = [torch.empty_like(v) for v in vin]
    tiles = (abi.SgmTile * n)()
    for i in range(n):
        tiles[i] = abi.SgmTile(vout[i].data_ptr(), vin[i].data_ptr(), X * Z, Z, Z, abi.ROI.make(0, X, 0, Y), C.pointer(pyr.desc))
    scratch = torch.empty(n * int(lib.avdm_volume_optimize_scratch_bytes(X, Y, Z)), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            abi.check(lib.avdm_volume_optimize_tiles(n, tiles, C.c_void_p(scratch.data_ptr()), C.byref(sgm), st))
        b.record()
        torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"tiles={n} {X}x{Y}x{Z}: {ms:.3f} ms per call, {11.0 * X * Y * Z * n / ms / 1e6:.0f} GB/s algorithmic, frac of 8 TB/s = {11.0 * X * Y * Z * n / ms / 1e6 / 8000:.3f}", flush=True)
