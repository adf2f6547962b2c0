"""Micro-benchmark of the depth-map filtering kernels (include/avdm_fuse.h) on the GPU box: one 12 MP reference camera against
N T cameras (default 10, like --nNearestCams), exact depth maps of the analytic scene; the CPU restatement timed on a bounded sample.

    python scripts/fuse_microbench.py [n_tcams] [width height] [--no-cpu]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from alicevision_amd import fuse
from fuse_scene import camera_structs, make_fuse_scene

args = [a for a in sys.argv[1:] if not a.startswith("--")]
nT = int(args[0]) if args else 10
W, H = (int(args[1]), int(args[2])) if len(args) > 2 else (4000, 3000)
dev = torch.device("cuda:0")
fs = make_fuse_scene(nT + 1, W, H, seed=2, noise=1e-4, outliers=0.03, weak=0.1, device="cuda:0")
cams = camera_structs(fs, fuse.fuse_camera)
maps = [torch.from_numpy(x).to(dev) for x in fs.depth]
sim = torch.from_numpy(fs.sim[0]).to(dev)
out = fuse.filter_groups(maps[0], sim, cams[0], cams[1:], maps[1:])
torch.cuda.synchronize()
reps = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fuse.filter_groups(maps[0], sim, cams[0], cams[1:], maps[1:], out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
items = float(sum(int((m > 0).sum().item()) for m in maps[1:]))
print("filter_groups %dx%d, %d T cameras: %.3f ms per reference camera = %.2f G items/s (%.1f M depth values projected), HBM-resident inputs"
      % (W, H, nT, ms, items / ms / 1e6, items / 1e6))
d, s = maps[0].clone(), sim.clone()
e0.record()
for _ in range(reps):
    fuse.filter_depth_maps(d, s, out)
e1.record()
torch.cuda.synchronize()
msf = e0.elapsed_time(e1) / reps
print("filter_depth_maps: %.3f ms = %.0f GB/s over 9 B per pixel read + 8 B written" % (msf, W * H * 17 / msf / 1e6))
print("modal counts: %s" % np.bincount(out.cpu().numpy().ravel(), minlength=nT + 1).tolist())

if "--no-cpu" not in sys.argv:
    from oracle import fuse_oracle as fo
    small = make_fuse_scene(nT + 1, 1000, 750, seed=2, noise=1e-4, outliers=0.03, weak=0.1)
    oc = camera_structs(small, fo.fuse_cam)
    t0 = time.time()
    fo.filter_groups_rc(small.depth[0], small.sim[0], oc[0], oc[1:], small.depth[1:])
    dt = time.time() - t0
    it = float(sum(int((m > 0).sum()) for m in small.depth[1:]))
    print("CPU restatement (1 core, 1000x750, %d T cameras): %.2f s = %.3f M items/s -> GPU/CPU-core = %.0fx" % (nT, dt, it / dt / 1e6, (items / ms * 1e3) / (it / dt)))
