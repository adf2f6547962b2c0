"""A/B of the C++ program (aliceVision_depthMapEstimation) on the bench's own scene (cfg3: 11 views 4000 x 3000, 256 planes, 10 T cameras, default
1024 tiling) under several environments: the scene is written ONCE, every variant runs REPEAT times, the program's timestamped log of each
run goes to OUT/<variant>_<k>.log.

    python scripts/cli_e2e_ab.py OUT "name:ENV=val,ENV=val" "name2:" ...
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
out_dir = sys.argv[1]
variants = sys.argv[2:] or ["default:"]
REPEAT = int(os.environ.get("REPEAT", "2"))
NCAM = int(os.environ.get("NCAM", "11"))
os.makedirs(out_dir, exist_ok=True)
d = tempfile.mkdtemp(prefix="avdm_e2e_ab_")
os.makedirs(os.path.join(d, "images"))
sc = make_scene(11, 4000, 3000, seed=3, device="cuda" if torch.cuda.is_available() else "cpu")
lms = scene_io.sample_landmarks(sc, 3000)
json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), open(os.path.join(d, "scene.sfm"), "w"))
for i in range(11):
    im = sc.images[i].cpu().numpy()
    exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]}, compression=0)
del sc
torch.cuda.empty_cache() if torch.cuda.is_available() else None
num = r"([0-9.eE+-]+)"
ref_maps = None
for v in variants:
    name, _, envs = v.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, val = kv.partition("=")
        env[k] = val
    for k in range(REPEAT):
        out = os.path.join(d, "out")
        shutil.rmtree(out, ignore_errors=True)
        args = [CLI, "-i", os.path.join(d, "scene.sfm"), "--imagesFolder", os.path.join(d, "images"), "-o", out, "--downscale", "1", "--rangeStart", "0",
                "--rangeSize", str(NCAM), "--sgmMaxDepths", "256", "--maxTCams", "10", "--sgmMaxTCamsPerTile", "10", "--refineMaxTCamsPerTile", "10", "-v", "info"]
        t0 = time.time()
        r = subprocess.run(args, capture_output=True, text=True, env=env)
        wall = time.time() - t0
        log = r.stdout + "\n---- stderr ----\n" + r.stderr
        open(os.path.join(out_dir, "%s_%d.log" % (name, k)), "w").write(log)
        task = re.findall(r"Task done in \(s\): " + num, log)
        setup = re.findall(r"set-up \(streams[^)]*\) in " + num + " s", log)
        dec = re.findall(r"Batch 1/\d+: images decoded, uploaded and converted to pyramids in " + num + " s", log)
        rel = re.findall(r"device buffers of the tile slots released in " + num + " s", log)
        workers = re.findall(num + r" s for \d+ camera\(s\) in all", log)
        # process start (spawn -> the program's first log line) and exit (its "Task done" line -> the process gone), from the log's own clock
        stamps = re.findall(r"^\[(\d\d):(\d\d):(\d\d\.\d+)\]", log, re.M)
        def clock(t):
            lt = time.localtime(t)
            return lt.tm_hour * 3600 + lt.tm_min * 60 + lt.tm_sec + (t - int(t))
        if stamps:
            first = int(stamps[0][0]) * 3600 + int(stamps[0][1]) * 60 + float(stamps[0][2])
            last = int(stamps[-1][0]) * 3600 + int(stamps[-1][1]) * 60 + float(stamps[-1][2])
            print("    spawn -> first log line %.3f s, last log line -> process gone %.3f s" % (first - clock(t0), clock(t0 + wall) - last))
        print("%-28s run %d: exit %d wall %.3f s  task %s  set-up %s  batch-1 ingest %s  worker %s  slot buffers released in %s" %
              (name, k, r.returncode, wall, task[-1] if task else "-", setup[0] if setup else "-", dec[0] if dec else "-", workers[0] if workers else "-",
               rel[0] if rel else "-"), flush=True)
        if r.returncode != 0:
            print(log[-1500:])
            continue
        # every variant must write the same maps (byte for byte) as the first one
        maps = {}
        for f in sorted(os.listdir(out)):
            if f.endswith(".exr"):
                maps[f] = open(os.path.join(out, f), "rb").read()
        import hashlib
        digest = hashlib.sha256(b"".join(maps[f] for f in sorted(maps))).hexdigest()[:16]
        if ref_maps is None:
            ref_maps = digest
        print("    %d map files, digest %s%s" % (len(maps), digest, "" if digest == ref_maps else "  *** DIFFERS from the first run ***"), flush=True)
shutil.rmtree(d, ignore_errors=True)
