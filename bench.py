#!/usr/bin/env python
"""bench.py — depth-maps/sec of the MI355X-native depth-map estimation hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload = "cfg3"): synthetic 11-view scene at 12 MP (4000x3000), 1 reference camera + 10 neighbours,
256 depth planes, full path per step = R-image pyramid build (+ all-gather of it across ranks when N > 1) ->
similarity volume x10 -> 4-path SGM aggregation -> WTA -> thickness smoothing -> upscale -> Refine volume x10 ->
sub-sample arg-min -> 100 optimisation iterations.  One step = one depth map; every rank computes K depth maps of
different reference cameras (weak scaling), value = N*K / max-over-ranks time.  Images are resident in HBM before the
timed region.  All compute goes through the C ABI of alicevision_amd/csrc/libavdm.so (hand-written HIP); the oracle is only
used for the `cpu_baseline` leg on rank 0 at N = 1.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from alicevision_amd import abi
from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
from alicevision_amd.sharding import cameras_of_rank, exchange_pyramid
from alicevision_amd.synthetic import make_scene, plane_depths

WORKLOADS = {
    # name: (views, width, height, planes, tcams)
    "cfg3": (11, 4000, 3000, 256, 10),
    "cfg2": (5, 1920, 1080, 128, 4),
    "cfg1": (3, 640, 480, 64, 2),
}


def cpu_baseline(sc_small, sgm, ref, n_planes, full_px, full_t):
    """Oracle (CPU restatement, OpenMP over all host cores) on a bounded sample of the same workload, scaled linearly to one
    full depth map.  kind = "port": the reference has no CPU path and cannot be built here."""
    from oracle import oracle
    t = {}
    imgs = sc_small.images.cpu().numpy()
    o = oracle.OracleDepthMap(imgs, sc_small.K, sc_small.R, sc_small.C, sgm, ref)
    depths = plane_depths(sc_small, n_planes)
    tcs = [1, 2]
    lib = oracle.load()
    t0 = time.time()
    o.run_sgm(0, tcs, depths)
    t["sgm"] = time.time() - t0
    t0 = time.time()
    o.run_refine(0, tcs)
    t["refine"] = time.time() - t0
    px = sc_small.width * sc_small.height
    # both stages are dominated by terms proportional to pixels x T cameras (similarity / refine volumes)
    scale = (full_px / px) * (full_t / len(tcs))
    total = (t["sgm"] + t["refine"]) * scale
    cores = os.cpu_count() or 1
    return {"value": 1.0 / total, "unit": "depth-maps/s", "cores": cores, "kind": "port",
            "sample": f"oracle SGM+Refine on {sc_small.width}x{sc_small.height}, {n_planes} planes, {len(tcs)} T cams "
                      f"({t['sgm'] + t['refine']:.1f} s), scaled x{scale:.0f} (pixels x T cams) to one 12 MP / 10 T depth map"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a launcher with WORLD_SIZE={args.gpus} (torch.distributed.run); got {world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    V, W, H, Z, T = WORKLOADS[args.workload]
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default()

    # ---- scene: every rank renders the views it owns, builds their pyramids, then the pyramids are exchanged over RCCL ----
    sc = make_scene(V, W, H, seed=3, device=dev)  # deterministic: identical on every rank (cameras needed everywhere)
    images = sc.images  # (V, H, W, 4) fp32, resident in HBM
    min_ds, max_ds = min(sgm.scale, ref.scale), max(sgm.scale, ref.scale) * 64
    pyr = []
    for v in range(V):
        if world > 1 and v % world != rank:
            pyr.append(DevicePyramid.allocate(W, H, min_ds, max_ds, abi.FILTER_CUDA_FIXED8, device=dev))  # received below
        else:
            pyr.append(DevicePyramid(images[v], min_ds, max_ds, abi.FILTER_CUDA_FIXED8, device=dev))
    torch.cuda.synchronize()
    t_ex = 0.0
    if world > 1:
        dist.barrier()
        t0 = time.time()
        for v in range(V):
            exchange_pyramid(pyr[v].buf, src=v % world, dist=dist)  # broadcast from the owner over xGMI
        torch.cuda.synchronize()
        t_ex = time.time() - t0
    depths = plane_depths(sc, Z)
    tile = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, device=dev)
    tile.enable_timers(True)
    lib = abi.load()
    lib.avdm_debug_sgm_kernel_timing.argtypes = [ctypes.c_int]
    lib.avdm_debug_sgm_kernel_timing_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.c_int]
    my_cams = cameras_of_rank(list(range(V)), rank, world)  # reference cameras of this rank (round-robin), cycled over steps

    def step(i):
        rc = my_cams[i % len(my_cams)]
        tcs = [v for v in range(V) if v != rc][:T]
        with tile.timers.range("image_pyramid"):
            pyr[rc].fill(images[rc])              # image -> Lab pyramid (DeviceCache::addMipmapImage)
        if world > 1:
            exchange_pyramid(pyr[rc].buf, src=rank, dist=dist, all_ranks=True)
        tile.run_sgm(rc, tcs, depths)
        return tile.run_refine(rc, tcs)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    tile.reset_timers()
    lib.avdm_debug_sgm_kernel_timing(1)  # HIP events on the launch stream around every path-aggregation kernel launch
    if world > 1:
        dist.barrier()
    t0 = time.time()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.time() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stages = tile.stage_ms()  # mean ms per step and stage (HIP events on the launch stream)
    k_ms, k_n = ctypes.c_double(0.0), ctypes.c_long(0)
    abi.check(lib.avdm_debug_sgm_kernel_timing_read(ctypes.byref(k_ms), ctypes.byref(k_n), 1), "avdm_debug_sgm_kernel_timing_read")
    lib.avdm_debug_sgm_kernel_timing(0)
    valid = float((out[..., 0] > 0).float().mean().item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * args.steps / elapsed
        # roofline of the SGM path-aggregation kernel (BASELINE.json: "SGM HBM GB/s vs roofline")
        ds = sgm.scale * sgm.stepXY
        X, Y = (W + ds - 1) // ds, (H + ds - 1) // ds
        # 11 B/voxel + 64 B/pixel for the four paths = two launches of sgm_pair_kernel (forward + reverse path of one
        # axis per launch); AVDM_SGM_PAIR=0 runs the four sequential sgm_path_kernel launches instead
        n_launches = 4 if os.environ.get("AVDM_SGM_PAIR") == "0" else 2
        alg_bytes_per_launch = (11.0 * X * Y * Z + 64.0 * X * Y) / n_launches
        # average duration of one path-aggregation kernel launch (events around the launches alone); stages["sgm_optimize"] is the
        # whole avdm_volume_optimize call, i.e. these launches + the adaptive-P2 map kernel
        if k_n.value != n_launches * args.steps:
            raise SystemExit(f"expected {n_launches * args.steps} path-kernel launches in the timed region, the library timed {k_n.value}")
        sgm_ms_per_launch = k_ms.value / k_n.value
        achieved = alg_bytes_per_launch / (sgm_ms_per_launch * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "sgm_pair_kernel" if n_launches == 2 else "sgm_path_kernel", "achieved": achieved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None, "alg_bytes_per_launch": alg_bytes_per_launch,
                "ms_per_launch": sgm_ms_per_launch, "launches_per_volume": n_launches,
                "ms_whole_call_per_volume": stages["sgm_optimize"]}
        pmc = os.path.join(ROOT, "profiles", "r01_sgm_pmc.json")
        if os.path.exists(pmc):
            try:
                roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        line = {
            "metric": "depth-maps/sec (12 MP, 256 depth hyp, 10 neighbours)", "value": value, "unit": "depth-maps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (fp16 texels, u8 cost volume)", "data": "synthetic",
            "config": {"workload": args.workload, "views": V, "width": W, "height": H, "depth_planes": Z, "t_cams": T,
                       "sgm": "scale 2 stepXY 2 wsh 4, 4 paths", "refine": "scale 1 stepXY 1 wsh 3, 31 planes, 100 opt iters",
                       "sharding": f"round-robin reference cameras over {world} rank(s)", "pyramid_exchange_s": t_ex},
            "roofline": roof,
            # the similarity kernels are VALU-issue bound, HBM fraction is not their figure: voxel x T camera rates instead
            # (a voxel-T is (2 wsh + 1)^2 patch samples: 81 for the SGM volume, 49 for the Refine volume)
            "similarity": {"sgm_voxelT_per_s": X * Y * Z * T / (stages["sgm_similarity"] * 1e-3),
                           "refine_voxelT_per_s": W * H * (2 * ref.halfNbDepths + 1) * T / (stages["refine_similarity"] * 1e-3),
                           "sgm_samples_per_s": X * Y * Z * T * (2 * sgm.wsh + 1) ** 2 / (stages["sgm_similarity"] * 1e-3),
                           "refine_samples_per_s": W * H * (2 * ref.halfNbDepths + 1) * T * (2 * ref.wsh + 1) ** 2 / (stages["refine_similarity"] * 1e-3)},
            "stages_ms": stages, "valid_fraction": valid,
        }
        if os.environ.get("AVDM_SIM_STATS") == "1":
            st = (ctypes.c_uint * 4)()
            abi.load().avdm_debug_similarity_stats(st)
            line["similarity_plane_workgroups"] = {"lds": int(st[0]), "generic_r_tile": int(st[1]), "generic_t_outside": int(st[2]),
                                                   "generic_t_too_large": int(st[3])}
        if world == 1 and not args.no_cpu_baseline:
            small = make_scene(3, 512, 384, seed=3, device="cpu")
            line["cpu_baseline"] = cpu_baseline(small, sgm, ref, Z, W * H, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
