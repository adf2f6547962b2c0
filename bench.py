#!/usr/bin/env python
"""bench.py — depth-maps/sec of the MI355X-native depth-map estimation hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads (BASELINE.json configs; config.workload names the one that ran):
  cfg3 (default at N = 1)  11 views at 12 MP (4000x3000), 1 reference camera + 10 neighbours, 256 depth planes — the metric's configuration;
  cfg4 (default at N > 1)  20 views at 12 MP sharded round-robin over the ranks, 10 neighbours each, 256 planes — the same work per depth map;
  cfg5                     100 views at 24 MP (6000x4000), each depth map computed as 4 x 4 tiles (tile buffer 1664 x 1152, padding 64);
  cfg2 / cfg1              the smaller parity configurations.
One step = one depth map of one reference camera: R-image pyramid build -> similarity volume x10 -> 4-path SGM aggregation -> WTA ->
thickness smoothing -> upscale -> Refine volume x10 -> sub-sample arg-min -> 100 optimisation iterations.  Every rank computes K depth
maps of reference cameras it owns (weak scaling); value = N*K / max-over-ranks time; `fixed_job` next to it prices the FIXED job BASELINE
quotes (all 20 cameras of cfg4 dealt to the N ranks, finished when the slowest rank is) from the per-rank seconds per depth map of this run.
Images are resident in HBM before the timed region.  Multi-GPU: a view's pyramid is built ONLY by the rank that owns the view; all pyramids
of a rank live in one arena whose rows of `world` views travel in ONE in-place all-gather each at set-up ("neighbour views broadcast once"),
and NO collective runs inside the timed region — the neighbour pyramids a rank sweeps against are the bytes it received
(alicevision_amd/sharding.py: ViewExchange, StepProtocol).  `--stream-views` is the streaming job instead: each step's freshly rebuilt R
pyramids travel by one all-gather on a side stream, beside the sweep, into a staging row that the next step commits.  `--force-dist` runs
all of it with one rank.
All compute goes through the C ABI of alicevision_amd/csrc/libavdm.so (hand-written HIP); the oracle is only used for the
`cpu_baseline` leg on rank 0 at N = 1.
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
from alicevision_amd.pipeline import DepthMapTile, DevicePyramid, optimize_tiles_batched
from alicevision_amd.sharding import measured_fixed_job  # noqa: E402
from alicevision_amd.sharding import StepProtocol, ViewExchange, cameras_of_rank, fixed_job, owner_of_view
from alicevision_amd.synthetic import make_scene, plane_depths

WORKLOADS = {
    # name: (views, width, height, planes, tcams, tiles per side)
    "cfg3": (11, 4000, 3000, 256, 10, 1),
    "cfg4": (20, 4000, 3000, 256, 10, 1),
    "cfg5": (100, 6000, 4000, 256, 10, 4),
    "cfg2": (5, 1920, 1080, 128, 4, 1),
    "cfg1": (3, 640, 480, 64, 2, 1),
}


TILE_BUFFER = (1664, 1152)  # cfg5: 4 x 4 tiles of a 6000 x 4000 image with the default padding of 64


def tile_rois(W, H, n_side, buffer_w=TILE_BUFFER[0], buffer_h=TILE_BUFFER[1], padding=64, max_downscale=4):
    """mvsUtils::getTileRoiList (mvsUtils/TileParams.cpp:15-61) for the cfg5 geometry, as (x0, x1, y0, y1): the number of tiles per side
    from the buffer size without its padding, tiles of equal effective size (a multiple of the largest downscale, SGM scale x step = 4)
    that start every effective width / height and extend by the padding at their END only, clipped to the image; column-major order.
    (tests/test_host_ref.py pins this layout to the reference's own function.)"""
    if n_side == 1:
        return [None]
    ceil_div = lambda a, b: (a + b - 1) // b
    nx, ny = ceil_div(W, buffer_w - 2 * padding), ceil_div(H, buffer_h - 2 * padding)
    ew = ceil_div(ceil_div(W, max_downscale), nx) * max_downscale
    eh = ceil_div(ceil_div(H, max_downscale), ny) * max_downscale
    return [(i * ew, min((i + 1) * ew + padding, W), j * eh, min((j + 1) * eh + padding, H)) for i in range(nx) for j in range(ny)]


def sgm_algorithmic_bytes(vols, Z, prepared=True):
    """Algorithmic bytes of ONE aggregation of the volumes `vols` = [(X, Y), ...] with Z planes: per voxel 4 reads of the
    input volume, 3 reads + 4 writes of the output volume = 11 B; per pixel the adaptive P2 — 64 B of R texels read by the map kernel, and
    one float per (path, pixel) = 16 B of P2 maps read by the four path walks.
      timed_call   what the timed `avdm_volume_optimize_tiles_prepared` call moves: 11 B/voxel + 16 B/pixel (the maps are evaluated beside the
                   similarity sweep since round 4); the one-call form (prepared = False) contains the map kernel: SURVEY's figure;
      survey       SURVEY's 11 B/voxel + 64 B/pixel, to be divided by the call PLUS the map kernel's time."""
    survey = sum(11.0 * x * y * Z + 64.0 * x * y for x, y in vols)
    timed = sum(11.0 * x * y * Z + 16.0 * x * y for x, y in vols) if prepared else survey
    return {"timed_call": timed, "survey": survey}


def cpu_baseline(sc_small, sgm, ref, n_planes, full_px, full_t):
    """The CPU side of the metric on a bounded sample of the same workload (SGM + Refine of one R camera against 2 T cameras), scaled
    linearly to one full depth map.
    kind = "reference": the REFERENCE'S OWN kernel-launch layer compiled for the CPU (oracle/_ref/libavdm_ref.so: its 19 cuda_* wrappers and
    every kernel over the stand-in CUDA runtime of oracle/ref/shim, each launch an OpenMP loop over the grid on all host cores) — used when
    the prebuilt library travelled with the snapshot; kind = "port": the oracle's restatement (OpenMP over rows) otherwise."""
    imgs = sc_small.images.cpu().numpy()
    depths = plane_depths(sc_small, n_planes)
    tcs = [1, 2]
    kind, o = "port", None
    try:
        from oracle import ref as refmod
        if os.path.exists(refmod.LIB_PATH):
            o = refmod.RefDepthMap(imgs, sc_small.K, sc_small.R, sc_small.C, sgm, ref)
            kind = "reference"
    except Exception:
        o = None
    if o is None:
        from oracle import oracle
        o = oracle.OracleDepthMap(imgs, sc_small.K, sc_small.R, sc_small.C, sgm, ref)
    t = {}
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
    what = "the reference's kernels on the CPU (oracle/_ref)" if kind == "reference" else "oracle"
    return {"value": 1.0 / total, "unit": "depth-maps/s", "cores": cores, "kind": kind,
            "sample": f"{what}: SGM+Refine on {sc_small.width}x{sc_small.height}, {n_planes} planes, {len(tcs)} T cams "
                      f"({t['sgm'] + t['refine']:.1f} s), scaled x{scale:.0f} (pixels x T cams) to one 12 MP / 10 T depth map"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, argv, port=None):
    """python -m torch.distributed.run ... bench.py <argv>: N ranks on this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
            str(port if port is not None else free_port()), os.path.abspath(__file__)] + list(argv)


def dry_run(args, rank, world):
    """The N-rank protocol of the bench without the GPU: process group (gloo), the views / reference cameras each rank owns, W warm-up + K timed
    "steps" bracketed by barriers, MAX over ranks of the elapsed time, ONE JSON line from rank 0."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    V, _, _, _, T, _ = WORKLOADS[args.workload or ("cfg3" if world == 1 else "cfg4")]
    cams_of = [cameras_of_rank(list(range(V)), r, world) for r in range(world)]
    my_cams = cams_of[rank]
    owned = [v for v in range(V) if owner_of_view(v, world) == rank]
    # the rank-side protocol of a step with stand-ins for the kernels: "pyramids" of 64 bytes whose content names (view, version)
    ex = ViewExchange(V, 64, rank, world, dist if world > 1 else None)
    version = {v: 0 for v in range(V)}
    for v in owned:
        ex.buffer(v).fill_(1 + (v % 16))
    ex.setup()
    done, checked = [], [0]

    def build(rc):
        assert owner_of_view(rc, world) == rank, "a rank only ever builds a view it owns"
        if args.stream_views:
            version[rc] += 1
        ex.buffer(rc).fill_(1 + (rc % 16) + 16 * (version[rc] % 15))

    def sweep(rc, tcs):
        for v in tcs:  # whatever this rank sweeps against is a WHOLE pyramid of the right view: built here or received
            b = ex.buffer(v)
            assert int(b[0]) == int(b[-1]) and (int(b[0]) - 1) % 16 == v % 16, (v, int(b[0]), int(b[-1]))
            checked[0] += 1
        done.append(rc)

    proto = StepProtocol(ex, cams_of, V, min(T, V - 1), build, sweep, stream_views=args.stream_views)
    fixed = world > 1 and not args.weak and not args.stream_views  # main(): one step = the workload's cameras once, each on its rank
    per_step = len(my_cams) if fixed else 1
    for i in range(args.warmup):
        proto.step(i)
    done.clear()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    for i in range(args.steps):
        for k in range(per_step):
            proto.step((0 if fixed else args.warmup) + i * per_step + k)  # (the fixed job: every pass in the order round-robin dealt the cameras)
    proto.finish()
    if world > 1:
        dist.barrier()
    elapsed = time.time() - t0
    counts = torch.tensor([len(owned), len(my_cams), checked[0]], dtype=torch.int64)
    done_of = [list(done)]
    my_elapsed = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.all_reduce(counts)
        done_of = [None] * world
        dist.all_gather_object(done_of, list(done))  # what every rank really computed in the timed region, in order
    if rank == 0:
        # the fixed-job accounting with one second per depth map on every rank: cameras per rank, makespan, the ceiling of the speed-up
        n_job = WORKLOADS["cfg4"][0] if (args.workload or "cfg3") in ("cfg3", "cfg4") else V
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "views": V, "views_owned_total": int(counts[0]),
                          "reference_cameras_total": int(counts[1]), "rank0_cameras": done, "cameras_done_of_rank": done_of, "elapsed_s": elapsed,
                          "rank0_elapsed_s": my_elapsed, "scaling": "strong" if fixed else "weak",
                          "value": (V if fixed else world) * args.steps / elapsed if elapsed > 0 else None,
                          "measured_fixed_job": measured_fixed_job(V, world, cams_of, [1.0] * world, elapsed / args.steps) if fixed else None,
                          "stream_views": bool(args.stream_views), "reference_arithmetic": args.reference_arithmetic or None, "exchange_collectives": ex.collectives, "tcam_pyramids_checked": int(counts[2]),
                          "fixed_job": fixed_job(n_job, world, [1.0] * world)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cli_end_to_end(sc, V, W, H, Z, T, n_cams):
    """The same scene through the PROGRAM (alicevision_amd/bin/aliceVision_depthMapEstimation, the C++ host above the C ABI): the views written
    as linear float EXR + an .sfm file, then n_cams reference cameras with the default tiling (tile buffer 1024, padding 64: 20 tiles per
    12 MP camera, batched SGM aggregation), Z planes, T neighbours — wall time of the process, i.e. start-up, EXR decode, upload, every tile,
    tile merge and EXR output included."""
    import subprocess
    import tempfile
    from alicevision_amd import exr_io, scene_io
    cli = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
    if not os.path.exists(cli):
        return {"error": "host program not built"}
    d = tempfile.mkdtemp(prefix="avdm_bench_cli_")
    os.makedirs(os.path.join(d, "images"))
    t0 = time.time()
    lms = scene_io.sample_landmarks(sc, 3000)
    with open(os.path.join(d, "scene.sfm"), "w") as f:
        json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), f)
    for i in range(V):
        im = sc.images[i].cpu().numpy()
        exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]},
                         compression=0)
    t_write = time.time() - t0
    cmd = [cli, "-i", os.path.join(d, "scene.sfm"), "--imagesFolder", os.path.join(d, "images"), "-o", os.path.join(d, "out"), "--downscale", "1", "--rangeStart", "0",
           "--rangeSize", str(n_cams), "--sgmMaxDepths", str(Z), "--maxTCams", str(T), "--sgmMaxTCamsPerTile", str(T), "--refineMaxTCamsPerTile", str(T), "-v",
           "info"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    if r.returncode != 0:
        return {"error": (r.stdout + r.stderr)[-400:]}
    # the program's own clock (host/DepthMapEstimator.cpp: per batch — images decoded / uploaded / converted, tiles computed, maps merged and written)
    import re
    log = r.stdout + r.stderr
    num = r"([0-9.eE+-]+)"
    dec = {int(b): float(x) for b, x in re.findall(r"Batch (\d+)/\d+: images decoded, uploaded and converted to pyramids in " + num + " s", log)}
    comp = {}
    for b, x in re.findall(r"Batch (\d+)/\d+: \d+ tile\(s\) computed, " + num + " s since the batch started", log):
        comp[int(b)] = max(comp.get(int(b), 0.0), float(x))
    wr = {int(b): float(x) for b, x in re.findall(r"Batch (\d+)/\d+: depth / similarity maps merged and written,\s*" + num + " s since the batch started", log)}
    task = [float(x) for x in re.findall(r"Task done in \(s\): " + num, log)]
    # what the program actually swept (host/DepthMapEstimator.cpp: per batch, in voxel x T camera; the depth lists are capped per tile and every
    # T camera has its own plane range)
    work = re.findall(r"Batch \d+/\d+: swept (\d+) tile\(s\): (\d+) SGM voxel-T, (\d+) Refine voxel-T; per tile on average " + num + " planes, " + num +
                      " SGM T cameras, " + num + " Refine T cameras", log)
    swept = None
    if work:
        nt = sum(int(w[0]) for w in work)
        swept = {"tiles": nt, "sgm_voxelT": sum(int(w[1]) for w in work), "refine_voxelT": sum(int(w[2]) for w in work),
                 "planes_per_tile": sum(int(w[0]) * float(w[3]) for w in work) / nt, "sgm_tcams_per_tile": sum(int(w[0]) * float(w[4]) for w in work) / nt,
                 "refine_tcams_per_tile": sum(int(w[0]) * float(w[5]) for w in work) / nt}
    split = None
    if dec and set(dec) == set(comp):
        # per batch: decode + upload + pyramids, then the tiles (every stage of every tile, results copied back), then merge + EXR output — the
        # output of a batch runs in the background beside the next batch's tiles, so only its tail after the last tile is on the critical path
        setup = [float(x) for x in re.findall(r"set-up \(streams, per-stream device buffers[^)]*\) in " + num + " s", log)]
        tail = [float(x) for x in re.findall(r"waited " + num + " s for the last batch's maps to be merged and written", log)]
        stalls = [float(x) for x in re.findall(r"waited " + num + " s for the previous batch's maps to be written", log)]
        split = {"batches": len(dec), "decode_upload_pyramids_s": sum(dec.values()), "tiles_s": sum(comp[b] - dec[b] for b in dec),
                 # merge + EXR output of a batch runs beside the next batch's tiles: on the critical path are only the waits for it
                 "merge_write_beside_the_tiles_s": sum(max(wr[b] - comp[b], 0.0) for b in wr if b in comp),
                 "merge_write_waited_s": (sum(stalls) + sum(tail)) if tail else None, "setup_s": setup[0] if setup else None,
                 "task_s": task[-1] if task else None, "process_start_and_scene_s": (wall - task[-1]) if task else None}
    return {"value": n_cams / wall, "unit": "depth-maps/s", "cameras": n_cams, "wall_s": wall, "scene_write_s": t_write, "split": split, "swept": swept,
            "includes": "process start, EXR decode of the views, upload, pyramids, default 1024 tiling (tiles batched per SGM launch), tile merge, EXR output"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cli-e2e", type=int, default=-1, metavar="N",
                    help="after the timed region (1 GPU): run the C++ program aliceVision_depthMapEstimation on N reference cameras of the same scene "
                         "(EXR files, default 1024 tiling, I/O included) and add its end-to-end rate to the line as `cli_end_to_end`; "
                         "default: all cameras of cfg3 (about 10 s), 0 = off")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group (nccl = RCCL) and run the pyramid exchange's collectives even with ONE rank: the multi-GPU code "
                         "path exercised on a single GPU (tests/test_gpu_parity.py::test_bench_rccl_path_on_one_gpu)")
    ap.add_argument("--stream-views", action="store_true",
                    help="the streaming job: every step's freshly rebuilt R pyramids travel to every other rank by one all-gather on a side stream "
                         "(stages pyramid_exchange / pyramid_commit).  Default: the pyramids are handed over once, before the timed region")
    ap.add_argument("--reference-arithmetic", default="", choices=["", "sgm", "all"],
                    help="run the similarity sweeps in the product's reference-arithmetic mode (avdm_sgm_params_t / avdm_refine_params_t::referenceArithmetic: "
                         "the reference's operations as written, volumes equal to its own code compiled for the CPU bit for bit): sgm = the SGM sweep, "
                         "all = both sweeps.  The parity mode's measured cost; NOT the headline configuration (the line says so in config.reference_arithmetic)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: the weak-scaling form of rounds 1-5 (every rank times K depth maps of its own; value = N K / elapsed).  Default at "
                         "N > 1 is the FIXED job BASELINE quotes (a step = all reference cameras of the workload dealt round-robin to the ranks, "
                         "timed to the slowest rank; value = cameras / makespan)")
    ap.add_argument("--no-parity-mode-cost", action="store_true",
                    help="skip the leg after the timed region that measures what the product's reference-arithmetic mode costs (N = 1: two depth maps with "
                         "the SGM sweep in that mode, ~4 s)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / timing protocol only (gloo, no GPU work): what tests/test_sharding.py runs on the CPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks on this node (one process per GPU) — the command the driver uses
        os.execvp(sys.executable, launcher_command(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only without a launcher (--force-dist at one rank)
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload is None:
        args.workload = "cfg3" if world == 1 else "cfg4"
    V, W, H, Z, T, n_side = WORKLOADS[args.workload]
    sgm = abi.SgmParams.default(referenceArithmetic=1 if args.reference_arithmetic in ("sgm", "all") else 0)
    ref = abi.RefineParams.default(referenceArithmetic=1 if args.reference_arithmetic == "all" else 0)

    # ---- scene: cameras everywhere; a rank renders and converts ONLY the views it owns ----
    sc = make_scene(V, W, H, seed=3, device=dev, render=[v for v in range(V) if owner_of_view(v, world) == rank])
    images = sc.images  # {view: (H, W, 4) fp32 in HBM} for the owned views
    min_ds, max_ds = min(sgm.scale, ref.scale), max(sgm.scale, ref.scale) * 64
    # all pyramids of a rank live in the exchange arena ([row][owner rank][bytes]): a row of views travels in ONE in-place all-gather
    if not cameras_of_rank(list(range(V)), rank, world):
        raise SystemExit(f"rank {rank} of {world} has no reference camera of the {V}-view workload")
    exchange = ViewExchange(V, DevicePyramid.pyramid_bytes(W, H, min_ds, max_ds, abi.FILTER_CUDA_FIXED8), rank, world, dist, device=dev)
    pyr = []
    for v in range(V):
        if v in images:
            pyr.append(DevicePyramid(images[v], min_ds, max_ds, abi.FILTER_CUDA_FIXED8, device=dev, storage=exchange.buffer(v)))
        else:  # received, never built here
            pyr.append(DevicePyramid.allocate(W, H, min_ds, max_ds, abi.FILTER_CUDA_FIXED8, device=dev, storage=exchange.buffer(v)))
    torch.cuda.synchronize()
    t_ex = 0.0
    if dist is not None:
        dist.barrier()
        t0 = time.time()
        exchange.setup()  # every view's pyramid handed to every rank once over xGMI: one collective per row of `world` views
        torch.cuda.synchronize()
        t_ex = time.time() - t0
    depths = plane_depths(sc, Z)
    rois = tile_rois(W, H, n_side)
    # several tiles per depth map: volumes laid out for and aggregated over the tile BUFFER like the reference (pipeline.DepthMapTile)
    tile_buffer = TILE_BUFFER if n_side > 1 else None
    tiles = [DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=r, device=dev, tile_buffer=tile_buffer) for r in rois]
    for t in tiles:
        t.timers = tiles[0].timers
    tile = tiles[0]
    tile.enable_timers(True)
    lib = abi.load()
    lib.avdm_debug_sgm_kernel_timing.argtypes = [ctypes.c_int]
    lib.avdm_debug_sgm_kernel_timing_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.c_int]
    cams_of = [cameras_of_rank(list(range(V)), r, world) for r in range(world)]  # reference cameras per rank (round-robin = the views it owns)
    my_cams = cams_of[rank]

    def build(rc):
        pyr[rc].fill(images[rc])                  # image -> Lab pyramid (DeviceCache::addMipmapImage): only ever for a view I own

    def sweep(rc, tcs):
        out = None
        if len(tiles) == 1:
            tiles[0].run_sgm(rc, tcs, depths)
            return tiles[0].run_refine(rc, tcs)
        # several tiles per depth map (cfg5): sweep every tile, aggregate ALL their volumes with one launch per axis
        # (avdm_volume_optimize_tiles, what host/DepthMapEstimator.cpp does for every group of tiles), then refine every tile
        for t in tiles:
            t.run_sgm(rc, tcs, depths, optimize="defer")
        optimize_tiles_batched(tiles, rc, timers=tile.timers)
        for t in tiles:
            out = t.run_refine(rc, tcs)
        return out

    # the rank-side order of a step (commit / build / publish / sweep: sharding.StepProtocol — the same object tests/test_sharding.py runs with
    # two gloo ranks and stand-in kernels)
    proto = StepProtocol(exchange, cams_of, V, T, build, sweep, stream_views=args.stream_views, on_stage=tile.timers.range)
    step = proto.step

    # N > 1 (default): the FIXED job — one step = every reference camera of the workload once, each on the rank round-robin deals it to
    # (computeOnMultiGPUs.cpp:42-66; cfg4: 20 cameras -> 3, 3, 3, 3, 2, 2, 2, 2 at 8 ranks), timed to the slowest rank.  N = 1 (and --weak,
    # --stream-views): one step = one depth map on every rank.
    fixed = world > 1 and not args.weak and not args.stream_views
    per_step = len(my_cams) if fixed else 1  # depth maps of THIS rank per step
    for i in range(args.warmup):  # (warm-up: W depth maps per rank in either form)
        step(i)
    torch.cuda.synchronize()
    tile.reset_timers()
    kernel_events = os.environ.get("AVDM_BENCH_KERNEL_EVENTS", "1") != "0"  # 0 (diagnosis): no per-launch events inside the timed call
    lib.avdm_debug_sgm_kernel_timing(1 if kernel_events else 0)  # HIP events on the launch stream around every path-aggregation kernel launch
    exchange.commit()
    torch.cuda.synchronize()
    exchange.events = []
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_events[0].record()
    stats_each = []
    lean_each = []
    # The two figures of the SGM aggregation perturb each other (session r06_a: the start / stop events bound to the two launches of a call cost the
    # call ~17 us — 0.443 ms with them, 0.424 without, the sum of the kernels' own durations being 0.424): they are taken on ALTERNATING depth maps
    # of the timed region — even ones with the per-launch events (kernel durations, rocprofv3's figure), odd ones without (the whole call).
    n_maps_total = args.steps * per_step
    instrumented = []  # per depth map of the timed region: were the per-launch events on?
    for i in range(args.steps):
        for k in range(per_step):
            m = i * per_step + k
            instrumented.append(kernel_events and (m % 2 == 0 or n_maps_total < 2))
            lib.avdm_debug_sgm_kernel_timing(1 if instrumented[-1] else 0)
            out = step((0 if fixed else args.warmup) + m)  # (the fixed job: every pass in the order round-robin dealt the cameras)
        step_events[i + 1].record()
        if os.environ.get("AVDM_SIM_STATS") == "1":  # diagnosis only (synchronises every step): tap sources of the similarity kernels per step
            torch.cuda.synchronize()
            st = (ctypes.c_uint * 4)()
            lib.avdm_debug_similarity_stats(st)
            stats_each.append([int(v) for v in st])
        if os.environ.get("AVDM_LEAN_STATS") == "1" and hasattr(lib, "avdm_debug_lean_stats"):  # diagnosis only, with a -DAVDM_LEAN_STATS=1 variant (AVDM_LIB)
            st16 = (ctypes.c_uint * 32)()
            lib.avdm_debug_lean_stats(st16)
            lean_each.append([int(v) for v in st16])
    proto.finish()  # (streaming job: the last round's pyramids are part of it)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.time() - t0
    # seconds per depth map on every rank (GPU time between the step events): what the fixed-job figure is priced with
    my_step_s = 1e-3 * step_events[0].elapsed_time(step_events[-1]) / (args.steps * max(per_step, 1))
    step_s_of_rank = [my_step_s]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        allt = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, torch.tensor([my_step_s], dtype=torch.float64, device=dev))
        step_s_of_rank = [float(x) for x in allt.cpu()]

    n_maps0 = args.steps * per_step  # depth maps this rank computed in the timed region
    stages = tile.timers.mean_ms(per=n_maps0)  # ms per DEPTH MAP and stage (HIP events on the launch stream; summed over the tiles of a step)
    if dist is not None and args.stream_views:
        # the all-gather of a step's R pyramids runs on a side stream beside the sweep: its own events, not part of the critical path
        stages["pyramid_exchange"] = exchange.exchange_ms() / n_maps0
    k_ms, k_n = ctypes.c_double(0.0), ctypes.c_long(0)
    abi.check(lib.avdm_debug_sgm_kernel_timing_read(ctypes.byref(k_ms), ctypes.byref(k_n), 1), "avdm_debug_sgm_kernel_timing_read")
    path_ms, path_n = (ctypes.c_double * 4)(), (ctypes.c_long * 4)()
    if hasattr(lib, "avdm_debug_sgm_kernel_timing_read_paths"):
        lib.avdm_debug_sgm_kernel_timing_read_paths(path_ms, path_n)
    span_ms, span_n = ctypes.c_double(0.0), ctypes.c_long(0)
    if hasattr(lib, "avdm_debug_sgm_kernel_timing_read_spans"):
        lib.avdm_debug_sgm_kernel_timing_read_spans.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        lib.avdm_debug_sgm_kernel_timing_read_spans(ctypes.byref(span_ms), ctypes.byref(span_n))
    lib.avdm_debug_sgm_kernel_timing(0)
    valid = float((out[..., 0] > 0).float().mean().item())

    line = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # fixed job: the workload's cameras per step, all ranks together, over the time of the slowest rank; otherwise one depth map per rank and step
        value = (V if fixed else world) * args.steps / elapsed
        # roofline of the SGM path-aggregation kernel (BASELINE.json: "SGM HBM GB/s vs roofline")
        ds = sgm.scale * sgm.stepXY
        # (X, Y) the path aggregation walks for every tile of a depth map: the tile's buffer extent (= its ROI for a whole-image tile)
        vols = [t.sgm_extent() for t in tiles]
        swept = []  # (X, Y) of the ROI the similarity kernel sweeps
        for r in rois:
            x0, x1, y0, y1 = r if r is not None else (0, W, 0, H)
            swept.append(((x1 + ds - 1) // ds - x0 // ds, (y1 + ds - 1) // ds - y0 // ds))
        # 11 B/voxel + 64 B/pixel for the four paths = two launches of sgm_pair_kernel (forward + reverse path of one
        # axis per launch); AVDM_SGM_PAIR=0 runs the four sequential sgm_path_kernel launches instead
        n_launches = 4 if os.environ.get("AVDM_SGM_PAIR") == "0" else 2
        # several tiles per depth map are aggregated by ONE call (all their volumes per launch): the "volume" of the accounting is then the batch
        batched = len(vols) > 1
        n_calls = 1 if batched else len(vols)
        # SURVEY 8(d)'s figure for the whole aggregation, adaptive-P2 maps included: 11 B/voxel + 64 B/pixel of R texels
        nbytes = sgm_algorithmic_bytes(vols, Z, prepared=os.environ.get("AVDM_SGM_PREPARE") != "0")
        survey_bytes_per_volume = nbytes["survey"] / n_calls
        # ... of which the TIMED call (avdm_volume_optimize_tiles_prepared: the path launches alone) moves the 11 B/voxel and READS the P2 maps — one
        # float per (path, pixel) = 16 B/pixel; the 64 B/pixel of R texels belong to the map kernel, which runs beside the similarity sweep
        # (stage sgm_p2_map) since round 4.  One-call form (AVDM_SGM_PREPARE=0): the map kernel is inside the call, SURVEY's figure is the numerator.
        prepared = os.environ.get("AVDM_SGM_PREPARE") != "0"
        alg_bytes_per_volume = nbytes["timed_call"] / n_calls
        alg_bytes_per_launch = alg_bytes_per_volume / n_launches
        # average duration of one path-aggregation kernel launch (HIP events around the launches alone, on their stream);
        # stages["sgm_optimize"] is the whole avdm_volume_optimize call, i.e. these launches + the adaptive-P2 map kernel
        n_instr = sum(1 for f in instrumented if f)
        if not kernel_events:
            k_ms.value, k_n.value = float("nan"), 0
        if k_n.value != n_launches * n_instr * n_calls:
            raise SystemExit(f"expected {n_launches * n_instr * n_calls} path-kernel launches with events in the timed region, the library timed {k_n.value}")
        sgm_ms_per_launch = k_ms.value / k_n.value if k_n.value else float("nan")
        achieved = alg_bytes_per_launch / (sgm_ms_per_launch * 1e-3) / 1e9
        # the whole call: HIP events around it on the launch stream, over the depth maps WITHOUT per-launch events (all of them when there is one)
        call_events = tile.timers.events.get("sgm_optimize", [])
        assert len(call_events) == len(instrumented) * n_calls, (len(call_events), len(instrumented), n_calls)
        plain = [j for j, f in enumerate(instrumented) if not f] or list(range(len(instrumented)))
        whole_call_ms = sum(call_events[j * n_calls + c][0].elapsed_time(call_events[j * n_calls + c][1]) for j in plain for c in range(n_calls)) / (len(plain) * n_calls)
        whole_call_instr_ms = (sum(call_events[j * n_calls + c][0].elapsed_time(call_events[j * n_calls + c][1]) for j, f in enumerate(instrumented) if f for c in range(n_calls))
                               / max(n_instr * n_calls, 1)) if n_instr else None
        # Headline = the WHOLE avdm_volume_optimize_tiles_prepared call (HIP events around it on the launch stream: the path launches of a volume and
        # the gaps between them; the adaptive-P2 maps are evaluated beside the similarity sweep, stage sgm_p2_map).  The kernels alone (events
        # bound to each launch, what rocprofv3's kernel trace averages) are reported next to it as achieved_kernels_only / frac_kernels_only.
        achieved_call = alg_bytes_per_volume / (whole_call_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "sgm_pair_kernel" if n_launches == 2 else "sgm_path_kernel", "achieved": achieved_call, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved_call / 8000.0, "achieved_kernels_only": achieved, "traffic": None, "alg_bytes_per_launch": alg_bytes_per_launch,
                "ms_per_launch": sgm_ms_per_launch, "launches_per_volume": n_launches, "volumes_per_launch": len(vols) if batched else 1,
                "alg_bytes_per_volume": alg_bytes_per_volume,
                "alg_bytes": "11 B/voxel + 16 B/pixel (the P2 maps the path launches read)" if prepared else "SURVEY 8(d): 11 B/voxel + 64 B/pixel",
                "ms_whole_call_per_volume": whole_call_ms, "frac_whole_call": achieved_call / 8000.0,
                # how the two figures were taken: depth maps of the timed region with / without the per-launch events, and the whole call ON the
                # instrumented ones (what rounds 3-5 quoted as the headline: the events' own cost is in it)
                "depth_maps_with_per_launch_events": n_instr, "depth_maps_without": len(instrumented) - n_instr,
                "ms_whole_call_with_per_launch_events": whole_call_instr_ms,
                "frac_kernels_only": achieved / 8000.0,
                # SURVEY 8(d)'s bytes (with the 64 B/pixel of R texels) over the timed call PLUS the adaptive-P2 map kernel that reads them (stage
                # sgm_p2_map, beside the similarity sweep): the other self-consistent form of the same figure
                "frac_with_p2_map": survey_bytes_per_volume / ((whole_call_ms + stages.get("sgm_p2_map", 0.0) / n_calls) * 1e-3) / 1e9 / 8000.0,
                # per launch of a volume: [first filtering axis (paths 0 + 1), second axis (paths 2 + 3)]
                "ms_per_launch_by_axis": [path_ms[k] / path_n[k] if path_n[k] else None for k in (0, 2)]}
        if span_n.value > 0:
            # the call on the device's own clock: start of its first path kernel -> end of its last one (both launches and the gap between them), from
            # the events bound to the launches — without the two command-processor hops the bracketing hipEventRecord pair adds around the call
            roof["ms_call_span_per_volume"] = span_ms.value / span_n.value / (1 if batched else 1)
            roof["frac_call_span"] = alg_bytes_per_volume / (roof["ms_call_span_per_volume"] * 1e-3) / 1e9 / 8000.0
        # what this box's HBM delivers to a plain device-to-device copy (1 GiB read + 1 GiB written, measured here, after the timed region):
        # the practical ceiling next to the 8 TB/s nominal peak the fraction is quoted against (profiles/README.md: box-to-box variance)
        try:
            src_t = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0)
            dst_t = torch.empty_like(src_t)
            dst_t.copy_(src_t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst_t.copy_(src_t)
            e1.record()
            torch.cuda.synchronize()
            copy_gbps = 5 * 2.0 * src_t.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            roof["box_copy_GBps"] = copy_gbps
            roof["achieved_over_box_copy"] = achieved / copy_gbps
            del src_t, dst_t
        except Exception:
            pass
        # HBM bytes per launch from the PMC counters: NOT measured by this run (counters need their own rocprofv3 --pmc passes); the value
        # is the committed summary of the latest counter session of the SAME kernel and volume size, named here with its provenance
        import glob
        import hashlib
        sha = hashlib.sha256(open(os.path.join(ROOT, "alicevision_amd", "csrc", "avdm_sgm.hip"), "rb").read()).hexdigest()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sgm_pmc.json")), reverse=True):
            if not (n_side == 1 and (W, H, Z) == (4000, 3000, 256)):
                break
            try:
                rec = json.load(open(path))
            except Exception:
                continue
            name = os.path.basename(path)
            if rec.get("kernel_source_sha256") == sha:
                roof["traffic"] = rec.get("hbm_bytes_per_launch")
                roof["traffic_source"] = (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/sgm_microbench.py (2 x FETCH + WRITE, "
                                          "KiB), taken with this very csrc/avdm_sgm.hip (sha256 matches); not this run")
            else:
                roof["traffic_source"] = (f"profiles/{name} was measured with another csrc/avdm_sgm.hip (sha256 differs): stale, not quoted — re-run "
                                          "scripts/gpu_pmc_sgm.sh + scripts/collect_profiles.py")
            break
        # the similarity kernels are VALU / LDS-gather bound, HBM fraction is not their figure.  Work units: a voxel-T is
        # (2 wsh + 1)^2 patch samples (81 SGM, 49 Refine); flops per voxel-T from SURVEY §8(a) (8.1 k / 4.9 k); LDS bytes per sample as the
        # kernels read them (SGM: half-paired 8-byte records, 2 x 8 + 2 x 4 B per image; with two planes per pass the R taps are read once
        # for both planes: 24 + 12 B per plane-sample; Refine: paired 16-byte records, 4 x 16 B)
        px_sgm = sum(x * y for x, y in swept)
        px_ref = sum(((r[1] - r[0]) * (r[3] - r[2])) if r is not None else W * H for r in rois)
        nz_ref = 2 * ref.halfNbDepths + 1
        t_sgm, t_ref = stages["sgm_similarity"] * 1e-3, stages["refine_similarity"] * 1e-3
        vt_sgm, vt_ref = px_sgm * Z * T / t_sgm, px_ref * nz_ref * T / t_ref
        s_sgm, s_ref = vt_sgm * (2 * sgm.wsh + 1) ** 2, vt_ref * (2 * ref.wsh + 1) ** 2
        # The unit that bounds the similarity kernels is VALU ISSUE: wave-instructions x 4 cycles over the SIMD-cycles of
        # the launch.  Instruction counts and the busy / LDS-conflict fractions come from the committed PMC session of THIS kernel source
        # (profiles/r*_sim_pmc.json, sha256-stamped like roofline.traffic); the launch durations are this run's.
        sim_pmc = {"valu_issue_frac": None, "source": None}
        sha_sim = hashlib.sha256(open(os.path.join(ROOT, "alicevision_amd", "csrc", "avdm_similarity.hip"), "rb").read()).hexdigest()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sim_pmc.json")), reverse=True):
            if not (n_side == 1 and (W, H, Z, T) == (4000, 3000, 256, 10)):
                break
            try:
                rec = json.load(open(path))
            except Exception:
                continue
            name = os.path.basename(path)
            # the source the counters were taken with, or one whose kernels compile to the same instructions (scripts/isa_identity.py --certify)
            same_isa = [e for e in rec.get("isa_identical_sources", []) if e.get("sha256") == sha_sim]
            if rec.get("kernel_source_sha256") != sha_sim and not same_isa:
                sim_pmc["source"] = f"profiles/{name} was measured with another csrc/avdm_similarity.hip (sha256 differs): stale, not quoted — re-run scripts/pmc_similarity.sh + scripts/collect_sim_pmc.py"
                break
            stamp = ("taken with this very csrc/avdm_similarity.hip (sha256 matches)" if not same_isa else
                     "taken with a csrc/avdm_similarity.hip whose kernels compile to the same gfx950 instructions as this one's (" + str(same_isa[0].get("evidence")) + ")")
            simd_hz = 1024 * 2.4e9  # 256 CUs x 4 SIMDs at the 2.4 GHz peak clock (MI355X_MICROARCH.md); one wave64 VALU instruction = 4 cycles of a SIMD
            out_pmc = {}
            for key, t_stage in (("sgm", t_sgm), ("refine", t_ref)):
                k = rec["per_kernel"][key]
                t_launch = t_stage / T
                out_pmc[key] = {"valu_insts_per_launch": k["SQ_INSTS_VALU"], "ms_per_launch": 1e3 * t_launch,
                                "valu_issue_frac": k["SQ_INSTS_VALU"] * 4.0 / (simd_hz * t_launch), "lds_conflict_frac": k["lds_conflict_frac"],
                                # from the counter session itself (its own launch duration and GRBM_GUI_ACTIVE): the clock the kernel ran at, the
                                # issue fraction and the share of SIMD cycles the VALU pipe was held at THAT clock, wave cycles parked on s_waitcnt
                                "clock_GHz_pmc": k.get("clock_GHz"), "valu_issue_frac_at_measured_clock": k.get("valu_issue_frac_at_measured_clock"),
                                "valu_busy_frac_at_measured_clock": k.get("valu_busy_frac_at_measured_clock"), "waves_parked_frac": k.get("waves_parked_frac"),
                                "lds_busy_frac": k.get("lds_busy_frac")}
            sim_pmc = {"valu_issue_frac": {k: v["valu_issue_frac"] for k, v in out_pmc.items()}, "per_kernel": out_pmc,
                       "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction",
                       "source": f"profiles/{name}: rocprofv3 --pmc passes (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE, SQ_WAIT_ANY, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) of "
                                 "one T-camera launch of each kernel, " + stamp + "; durations: this run"}
            break
        line = {
            "metric": "depth-maps/sec (12 MP, 256 depth hyp, 10 neighbours)", "value": value, "unit": "depth-maps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            # N > 1: the FIXED job (total work does not grow with N: "strong"); N = 1 / --weak: one depth map per rank and step
            "scaling": "strong" if fixed else "weak",
            "fixed_job": measured_fixed_job(V, world, cams_of, step_s_of_rank, elapsed / args.steps) if fixed else
            # (N = 1, --weak: PRICED, not measured — this run's per-rank seconds per depth map x the cameras round-robin deals each rank)
            fixed_job(WORKLOADS["cfg4"][0] if args.workload in ("cfg3", "cfg4") else V, world, step_s_of_rank),
            # the weak-scaling rate next to it: depth maps per second summed over the ranks, each at its own pace
            "weak_scaling_rate": sum(1.0 / t for t in step_s_of_rank if t > 0),
            "vs_baseline": None, "dtype": "f32 (fp16 texels, u8 cost volume)", "data": "synthetic",
            "config": {"workload": args.workload, "views": V, "width": W, "height": H, "depth_planes": Z, "t_cams": T, "tiles_per_depth_map": len(rois),
                       "sgm": "scale 2 stepXY 2 wsh 4, 4 paths", "refine": "scale 1 stepXY 1 wsh 3, 31 planes, 100 opt iters",
                       "sharding": f"round-robin reference cameras over {world} rank(s); a view's pyramid is built by its owner only",
                       "stream_views": bool(args.stream_views),
                       "process_group": None if dist is None else f"nccl (RCCL), {world} rank(s)" + (" [--force-dist]" if world == 1 else ""),
                       "pyramid_setup_broadcast_s": t_ex, "pyramid_bytes_received_per_rank": exchange.bytes_received,
                       "pyramid_exchange_collectives": exchange.collectives},
            "roofline": roof,
            "similarity": {"sgm_voxelT_per_s": vt_sgm, "refine_voxelT_per_s": vt_ref, "sgm_samples_per_s": s_sgm, "refine_samples_per_s": s_ref,
                           "sgm_lds_GBps": s_sgm * (48.0 if os.environ.get("AVDM_SIM_PLANE_PAIRS") == "0" else 36.0) / 1e9, "refine_lds_GBps": s_ref * 64.0 / 1e9, "lds_peak_GBps": 150000.0,
                           # (the estimated-flop fractions of rounds 1-4 are gone: the kernels are bound by VALU ISSUE of mostly non-FMA instructions)
                           "valu_issue_frac": sim_pmc["valu_issue_frac"], "valu_issue": sim_pmc},
            "stages_ms": stages, "valid_fraction": valid,
            # GPU time of every timed step (HIP events between the steps): shows the clock settling under sustained load
            "ms_per_step_each": [round(step_events[i].elapsed_time(step_events[i + 1]), 2) for i in range(args.steps)],
            "similarity_ms_each": {k: [round(a.elapsed_time(b), 2) for a, b in tile.timers.events.get(k, [])] for k in ("sgm_similarity", "refine_similarity")}
            if len(tiles) == 1 else None,
        }
        if lean_each:
            # per step (diagnosis, variant build): SGM sweep {8-plane, 4-plane, 1-plane LDS, 1-plane global passes of a wave; workgroups without a chunk
            # window, workgroups, retried without outliers, -}, Refine sweep {8, 4, 1 LDS, 1 global; anchored windows, workgroups, without window, -}
            line["lean_pass_counters_each"] = lean_each
        if os.environ.get("AVDM_SIM_STATS") == "1":
            # [LDS path, generic: R tile unusable / nothing valid, generic: T taps leave the image, generic: T window exceeds the LDS budget] per step
            line["similarity_plane_workgroups_each"] = stats_each
        if os.environ.get("AVDM_REFINE_OUTLIER_STATS") == "1":
            # diagnosis: (pixel, chunk) units of the Refine outlier list over the timed steps {worked off, refused by a full list}, and per T-camera launch
            u = (ctypes.c_uint * 2)()
            lib.avdm_debug_refine_outlier_units.argtypes = [ctypes.POINTER(ctypes.c_uint)]
            lib.avdm_debug_refine_outlier_units(u)
            line["refine_outlier_units"] = {"worked_off": int(u[0]), "refused": int(u[1]), "per_launch": int(u[0]) / max(args.steps * T * len(tiles), 1),
                                            "pixel_chunks_per_launch": px_ref * ((nz_ref + 7) // 8) / max(len(tiles), 1), "includes_warmup": True}
        n_cli = args.cli_e2e if args.cli_e2e >= 0 else (V if args.workload == "cfg3" else 0)
        if world == 1 and n_cli > 0:
            e2e = cli_end_to_end(sc, V, W, H, Z, T, min(n_cli, V))
            sw = e2e.get("swept")
            if sw and e2e.get("split"):
                # the same work at THIS run's kernel rates (seconds per voxel-T of the two similarity stages; every other stage per camera):
                # what the program's tiles would take if it ran exactly at the bench's rate — the program sweeps other amounts than the
                # bench's 256 planes x 10 T cameras (padded tiles: more pixels; per-tile depth lists and per-T plane ranges: fewer voxel-T)
                other_ms = ms_per_step - stages["sgm_similarity"] - stages["refine_similarity"]
                pred = (sw["sgm_voxelT"] / vt_sgm) + (sw["refine_voxelT"] / vt_ref) + e2e["cameras"] * other_ms * 1e-3
                e2e["kernel_only_s_for_the_same_work"] = pred
                e2e["kernel_only_rate_for_the_same_work"] = e2e["cameras"] / pred
                e2e["value_over_kernel_only_rate"] = e2e["value"] / (e2e["cameras"] / pred)
                e2e["tiles_s_over_kernel_only_s"] = e2e["split"]["tiles_s"] / pred
                e2e["bench_voxelT_per_depth_map"] = {"sgm": px_sgm * Z * T, "refine": px_ref * nz_ref * T}
            line["cli_end_to_end"] = e2e
        if world == 1 and len(tiles) == 1 and not args.reference_arithmetic and not args.no_parity_mode_cost:
            # THE PARITY MODE'S COST, measured here ("with its measured cost in the bench line"): the same depth map with the SGM sweep in
            # the reference's arithmetic (avdm_sgm_params_t::referenceArithmetic — the mode in which every parity case meets BASELINE's bar, volumes
            # identical to the reference's own code), the default Refine kernels; one warm-up + two timed depth maps
            sgm_p = abi.SgmParams.default(referenceArithmetic=1)
            tile_p = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm_p, ref, roi=rois[0], device=dev, tile_buffer=tile_buffer)
            tile_p.enable_timers(True)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for j in range(3):
                rc = my_cams[j % len(my_cams)]
                if j == 1:
                    tile_p.reset_timers()
                if j >= 1:
                    ev[j - 1].record()
                tile_p.run_sgm(rc, proto.tcams_of(rc), depths)
                tile_p.run_refine(rc, proto.tcams_of(rc))
            ev[2].record()
            torch.cuda.synchronize()
            ms_p = ev[0].elapsed_time(ev[2]) / 2.0
            st_p = tile_p.timers.mean_ms(per=2)
            line["reference_arithmetic"] = {"mode": "sgm sweep (avdm_sgm_params_t::referenceArithmetic = 1), default Refine kernels",
                                            "value": 1e3 / ms_p, "unit": "depth-maps/s", "ms_per_step": ms_p, "steps": 2,
                                            "sgm_similarity_ms": st_p.get("sgm_similarity"), "refine_similarity_ms": st_p.get("refine_similarity"),
                                            "slowdown_vs_default": ms_p / ms_per_step,
                                            "parity": "similarity / aggregated volumes and WTA depths identical to the reference's own code compiled for the CPU; "
                                                      "final depth RMSE < 1e-3 untrimmed on every parity case (tests/test_gpu_parity.py::_assert_reference_arithmetic)"}
            del tile_p
        if world == 1 and not args.no_cpu_baseline:
            small = make_scene(3, 512, 384, seed=3, device="cpu")
            line["cpu_baseline"] = cpu_baseline(small, sgm, ref, Z, W * H, T)
    # ONE JSON line from rank 0, and it is the LAST thing on the job's stdout: RCCL prints a version banner through C stdio, which a pipe only
    # sees when the buffer is flushed — at process exit, i.e. AFTER a line Python printed earlier (session r05_dist).  So: every rank flushes its
    # C stdio, the ranks meet, the group is torn down, C stdio is flushed once more, and only then rank 0 prints.
    def flush_c_stdio():
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    if dist is not None:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
        if rank == 0 and world > 1:
            time.sleep(1.0)  # the other ranks' last flushes
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
