"""End-to-end run of the C++ host (aliceVision_depthMapEstimation) at cfg3 scale on the GPU box: 11 synthetic 12 MP views written
as linear float EXR + .sfm, default 1024 tiling (20 tiles per camera, batched SGM aggregation), N reference cameras.
Prints wall times and the depth error against the analytic ground truth.

    python scripts/cli_e2e_cfg3.py [n_cameras] [width height]
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4000, 3000)
d = "/tmp/avdm_cfg3"
os.makedirs(os.path.join(d, "images"), exist_ok=True)
t0 = time.time()
sc = make_scene(11, W, H, seed=3, device="cuda" if torch.cuda.is_available() else "cpu", baseline=0.9, amp=0.6)
lms = scene_io.sample_landmarks(sc, 3000, amp=0.6)
json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), open(os.path.join(d, "scene.sfm"), "w"))
for i in range(11):
    im = sc.images[i].cpu().numpy()
    exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]}, compression=0)
print("scene written in %.1f s" % (time.time() - t0), flush=True)
out = os.path.join(d, "out")
args = [CLI, "-i", os.path.join(d, "scene.sfm"), "--imagesFolder", os.path.join(d, "images"), "-o", out, "--downscale", "1", "--rangeStart", "0", "--rangeSize", str(ncam),
        "--sgmMaxDepths", "256", "--maxTCams", "10", "--sgmMaxTCamsPerTile", "10", "--refineMaxTCamsPerTile", "10", "-v", "info"]
args += os.environ.get("AVDM_E2E_ARGS", "").split()  # e.g. "--tileBufferWidth 4000 --tileBufferHeight 3000": one tile per camera
t0 = time.time()
r = subprocess.run(args, capture_output=True, text=True)
wall = time.time() - t0
if os.environ.get("AVDM_E2E_LOG"):  # the program's whole log (timestamped lines): where the seconds outside the tiles go
    open(os.environ["AVDM_E2E_LOG"], "w").write(r.stdout + "\n---- stderr ----\n" + r.stderr)
print("exit", r.returncode, "wall %.2f s for %d camera(s) -> %.3f depth-maps/s (including image decode, upload, EXR output)" % (wall, ncam, ncam / wall))
for l in r.stdout.splitlines():
    if any(k in l for k in ("Task done", "simultaneous", "tiles per image", "Optimizing volume of", "Batch ", "Found ", "Device memory (")):
        print("  ", l.strip()[:160])
print(r.stderr[-1500:])
if r.returncode == 0:
    dm, info = exr_io.read_exr(os.path.join(out, "%d_depthMap.exr" % scene_io.view_id(0)))
    depth = dm["Y"]
    gt = sc.gt_depth.cpu().numpy()
    m = depth > 0
    m[:32] = m[-32:] = False
    m[:, :32] = m[:, -32:] = False
    rel = np.abs(depth - gt)[m] / gt[m]
    print("valid %.4f, median |depth - gt| / gt = %.2e, 90th percentile %.2e" % (float((depth > 0).mean()), float(np.median(rel)), float(np.percentile(rel, 90))))

# optional: the filtering step on the maps just written (aliceVision_depthMapFiltering), when every camera was estimated
if r.returncode == 0 and ncam == 11 and os.environ.get("AVDM_E2E_FILTER", "1") == "1":
    FCLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapFiltering")
    flt = os.path.join(d, "filtered")
    t0 = time.time()
    r2 = subprocess.run([FCLI, "-i", os.path.join(d, "scene.sfm"), "--depthMapsFolder", out, "-o", flt, "--computeNormalMaps", "1", "-v", "info"],
                        capture_output=True, text=True)
    wall = time.time() - t0
    print("filtering: exit", r2.returncode, "wall %.2f s for 11 cameras (10 nearest cameras each, normal maps included)" % wall)
    for l in r2.stdout.splitlines():
        if any(k in l for k in ("Task done", "computed", "filtered", "normal maps of")):
            print("  ", l.strip()[:160])
    print(r2.stderr[-1500:])
    if r2.returncode == 0:
        fd = exr_io.read_exr(os.path.join(flt, "%d_depthMap.exr" % scene_io.view_id(0)))[0]["Y"]
        kept = (fd > 0) & m
        rel2 = np.abs(fd - gt)[kept] / gt[kept]
        print("filtered: kept %.4f of the valid interior depths, median error %.2e, 99.9th percentile %.2e (before: %.2e)"
              % (float(kept.sum()) / float(m.sum()), float(np.median(rel2)), float(np.percentile(rel2, 99.9)), float(np.percentile(rel, 99.9))))
