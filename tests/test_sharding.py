"""Multi-GPU plumbing on CPU: camera sharding and the pyramid exchange over torch.distributed (gloo, world_size 2)."""
import os
import sys

import numpy as np
import pytest

from alicevision_amd.sharding import cameras_of_rank, owner_of_view

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(20, 8), (100, 8), (7, 2), (3, 4), (1, 8)])
def test_shards_partition_the_camera_list(n, world):
    cams = list(range(100, 100 + n))
    for contiguous in (False, True):
        parts = [cameras_of_rank(cams, r, world, contiguous) for r in range(world)]
        assert sorted(sum(parts, [])) == cams
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # contiguous mode is computeOnMultiGPUs.cpp:49-63: [r*n/w, (r+1)*n/w)
    for r in range(world):
        assert cameras_of_rank(cams, r, world, True) == cams[(r * n) // world:((r + 1) * n) // world]
    assert {owner_of_view(v, world) for v in range(max(n, world))} <= set(range(world))


def _worker(rank, world, port, q):
    """rank-local part of the world-2 test.  The compute stand-in on CPU is the oracle (tests may use it); what is under test is the
    sharding: a view is built ONLY by its owner, every other rank computes from the bytes it RECEIVED."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from alicevision_amd import abi
    from alicevision_amd.sharding import ViewExchange, cameras_of_rank, exchange_pyramid, owner_of_view
    from alicevision_amd.synthetic import make_scene, plane_depths
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        V, W, H = 3, 96, 64
        sc = make_scene(V, W, H, seed=4)
        imgs = sc.images.numpy()
        sgm, rp = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=2)
        mode = abi.FILTER_CUDA_FIXED8

        def build(img):
            return oracle.HostPyramid(img, 1, 128, mode)

        # every rank builds ONLY the views it owns, into its slots of the exchange arena; the others start as zeros
        nbytes = build(imgs[0]).desc.bytes
        ex = ViewExchange(V, nbytes, rank, world, dist)
        bufs = [ex.buffer(v) for v in range(V)]
        for v in range(V):
            if owner_of_view(v, world) == rank:
                bufs[v].copy_(torch.from_numpy(build(imgs[v]).buf.copy()))
        ex.setup()
        ok &= ex.collectives == (V + world - 1) // world  # one collective per ROW of views, not one per view
        ok &= ex.bytes_received == nbytes * sum(1 for v in range(V) if owner_of_view(v, world) != rank)

        def depth_map(buffers, rc):
            pyr = [oracle.HostPyramid.from_bytes(W, H, 1, 128, mode, b.numpy()) for b in buffers]
            o = oracle.OracleDepthMap(None, sc.K, sc.R, sc.C, sgm, rp, filter_mode=mode, pyramids=pyr)
            o.run_sgm(rc, [v for v in range(V) if v != rc], plane_depths(sc, 8))
            return o.run_refine(rc, [v for v in range(V) if v != rc])

        # a depth map of one of MY reference cameras from received neighbour pyramids == the same from locally built ones
        rc = cameras_of_rank(list(range(V)), rank, world)[0]
        got = depth_map(bufs, rc)
        want = depth_map([torch.from_numpy(build(imgs[v]).buf.copy()) for v in range(V)], rc)
        ok &= bool(np.array_equal(got, want)) and bool((want[..., 0] > 0).mean() > 0.3)

        # steady state: every rank rebuilds the pyramid of a view it owns (a changed image); one all-gather brings the others' new pyramids
        # into the STAGING row — the arena keeps the old bytes until commit()
        mine = cameras_of_rank(list(range(V)), rank, world)[0]
        bufs[mine].copy_(torch.from_numpy(build(imgs[mine] * np.float32(0.8)).buf))
        views = [cameras_of_rank(list(range(V)), r, world)[0] for r in range(world)]
        ex.publish_async(views)
        for r, v in enumerate(views):
            if r != rank:
                ok &= bool(torch.equal(bufs[v], torch.from_numpy(build(imgs[v]).buf)))  # still the set-up bytes
        ex.commit()
        for r, v in enumerate(views):
            ok &= bool(torch.equal(bufs[v], torch.from_numpy(build(imgs[v] * np.float32(0.8)).buf)))
        # the one-shot helpers
        one = torch.full((64,), rank + 7, dtype=torch.uint8)
        got = exchange_pyramid(one, src=rank, dist=dist, all_ranks=True)
        ok &= len(got) == world and all(bool((g == r + 7).all()) for r, g in enumerate(got))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_depth_map_from_received_pyramids_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_exchange_is_identity_without_a_process_group():
    import torch
    from alicevision_amd.sharding import exchange_pyramid
    b = torch.arange(10, dtype=torch.uint8)
    assert exchange_pyramid(b, 0, None)[0] is b


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (one process per rank, rendezvous on
    127.0.0.1): the launch, the process group, the round-robin ownership and the barrier / MAX-over-ranks timing protocol run here on the
    CPU with --dry-run (gloo, no GPU work) and print ONE JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "cfg1", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["steps"] == 4
    assert out["views_owned_total"] == out["views"] == 3 and out["reference_cameras_total"] == 3  # every view / camera owned exactly once
    # rank 0 of 2 owns views 0 and 2; N > 1 runs the FIXED job: a step = every camera of the workload once, each on its rank (4 steps x 2 cameras)
    assert len(out["rank0_cameras"]) == 8 and set(out["rank0_cameras"]) == {0, 2} and out["scaling"] == "strong"
    # and the command it becomes is the driver's own
    sys.path.insert(0, root)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], port=29511)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[9].endswith("bench.py") and cmd[10:] == ["--gpus", "4", "--steps", "2"]


def _bench_dry_run(n, extra=()):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--steps", "3", "--warmup", "1"] + list(extra),
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n,per_rank,ceiling", [(1, [20], 1.0), (2, [10, 10], 2.0), (4, [5] * 4, 4.0), (8, [3, 3, 3, 3, 2, 2, 2, 2], 20.0 / 3.0)])
def test_fixed_job_accounting(n, per_rank, ceiling):
    """`bench.py --gpus N` prices, next to its weak-scaling value, the FIXED job BASELINE.json quotes its multi-GPU target on — the 20 reference
    cameras of cfg4 dealt round-robin to N ranks, finished when the slowest rank is (computeOnMultiGPUs.cpp:15-69 is that job).  The accounting
    (cameras per rank, makespan, ceiling of the speed-up: 6.67 at 8 ranks, not 8) through the real launch with --dry-run: N processes, gloo, one
    second per depth map on every rank.  The default timed region runs NO collective: the pyramids were handed over once at set-up."""
    out = _bench_dry_run(n)
    fj = out["fixed_job"]
    assert out["n_gpus"] == n and fj["cameras"] == 20 and fj["cameras_per_rank"] == per_rank
    assert fj["makespan_s"] == max(per_rank) * 1.0 and abs(fj["depth_maps_per_s"] - 20.0 / max(per_rank)) < 1e-12
    assert abs(fj["speedup_ceiling"] - ceiling) < 1e-12
    V = out["views"]
    assert out["views_owned_total"] == V and out["reference_cameras_total"] == V
    # set-up: one collective per row of N views; nothing after it
    assert not out["stream_views"] and out["exchange_collectives"] == (0 if n == 1 else (V + n - 1) // n)
    if n == 1:
        # one rank: a step = one depth map (the cfg3 shape of the headline), the fixed job stays a priced figure
        assert out["scaling"] == "weak" and out["measured_fixed_job"] is None and out["tcam_pyramids_checked"] == 4 * 10
        return
    # N > 1 (round 6): the bench RUNS the fixed job — a step is the 20 cameras once, each on the rank round-robin deals it to,
    # K = 3 steps timed barrier to barrier to the slowest rank; `value` = 20 K / that time
    assert out["scaling"] == "strong"
    for r in range(n):
        mine = list(range(r, 20, n))
        assert out["cameras_done_of_rank"][r] == mine * 3, (r, out["cameras_done_of_rank"][r])
    mj = out["measured_fixed_job"]
    assert mj["measured"] and mj["cameras_per_rank"] == per_rank and mj["cameras_of_rank"] == [list(range(r, 20, n)) for r in range(n)]
    assert abs(mj["makespan_s"] - out["elapsed_s"] / 3) < 1e-12 and abs(out["value"] - 20 * 3 / out["elapsed_s"]) < 1e-9
    assert abs(mj["depth_maps_per_s"] - out["value"]) < 1e-9 and out["elapsed_s"] >= out["rank0_elapsed_s"]  # MAX over ranks, not rank 0's clock
    # every T-camera pyramid a rank swept against was whole and of the right view: (1 warm-up + 3 x its cameras) depth maps per rank, 10 T each
    assert out["tcam_pyramids_checked"] == (n + 3 * 20) * 10


def test_streaming_job_runs_one_collective_per_step():
    """--stream-views: every step's rebuilt R pyramids travel by ONE all-gather (beside the sweep) and are committed at the next step boundary"""
    out = _bench_dry_run(2, ["--stream-views"])
    assert out["stream_views"] and out["exchange_collectives"] == 10 + 4  # 10 rows of 2 views at set-up + (1 warm-up + 3 timed) steps
    assert out["tcam_pyramids_checked"] == 2 * 4 * 10


def _protocol_worker(rank, world, port, q, stream_views):
    """the WHOLE rank-side step — commit / build / publish / sweep (sharding.StepProtocol, the object bench.py steps) — with two gloo ranks and
    stand-in kernels: `build` writes a pyramid whose bytes name (view, version), `sweep` records what the rank would have swept against"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from alicevision_amd.sharding import StepProtocol, ViewExchange, cameras_of_rank, owner_of_view
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V, T, nbytes = 5, 3, 4096 + 17
        ex = ViewExchange(V, nbytes, rank, world, dist)
        cams_of = [cameras_of_rank(list(range(V)), r, world) for r in range(world)]
        version = {v: 0 for v in range(V)}
        content = lambda v, k: (7 * v + 3 * k + 1) % 251
        for v in range(V):
            if owner_of_view(v, world) == rank:
                ex.buffer(v).fill_(content(v, 0))
        ex.setup()
        seen, built, stages = [], [], []

        def build(rc):
            built.append(rc)
            if stream_views:
                version[rc] += 1
            ex.buffer(rc).fill_(content(rc, version[rc]))

        def sweep(rc, tcs):
            seen.append((rc, [(v, int(ex.buffer(v)[0]), int(ex.buffer(v)[-1])) for v in tcs]))

        class stage:
            def __init__(self, name):
                stages.append(name)

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        proto = StepProtocol(ex, cams_of, V, T, build, sweep, stream_views=stream_views, on_stage=stage)
        n_steps = 5
        for i in range(n_steps):
            proto.step(i)
        proto.finish()
        ok = all(owner_of_view(rc, world) == rank for rc in built) and built == [cams_of[rank][i % len(cams_of[rank])] for i in range(n_steps)]
        # what step i sweeps against: view v at the version its owner had built BEFORE step i's round was committed, i.e. after i rounds for a
        # streamed job (a round is committed at the next step boundary: the pyramids of round i are visible from step i + 1), version 0 otherwise
        for i, (rc, taps) in enumerate(seen):
            ok &= [v for v, _, _ in taps] == [(rc + 1 + k) % V for k in range(T)]
            for v, first, last in taps:
                o = owner_of_view(v, world)
                rounds = i if o != rank else i + 1  # my own views: rebuilt in place by build() up to and including this step
                want_k = sum(1 for j in range(rounds) if cams_of[o][j % len(cams_of[o])] == v) if stream_views else 0
                ok &= first == last == content(v, want_k)
        ok &= ex.collectives == (V + world - 1) // world + (n_steps if stream_views else 0)
        ok &= (stages.count("pyramid_commit") == (n_steps + 1 if stream_views else 0)) and stages.count("image_pyramid") == n_steps
        ok &= ex._pending is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("stream_views", [False, True])
def test_whole_step_protocol_gloo_world2(stream_views):
    """not only ViewExchange — the whole step() of a rank (commit / fill / publish / sweep stub) with world_size 2 on gloo, in
    both modes: the default (pyramids handed over once, no collective per step) and the streaming job (one all-gather per step, visible to the
    sweeps from the NEXT step on, never half-written)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (1 if stream_views else 0)
    procs = [ctx.Process(target=_protocol_worker, args=(r, 2, port, q, stream_views)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_roofline_numerator_is_what_the_timed_call_moves():
    """VERDICT r4 / ADVICE r4: the headline fraction divides by the time of the path launches alone, so its numerator must not carry the 64 B /
    pixel of R texels that only the adaptive-P2 map kernel reads (it runs beside the similarity sweep).  cfg3: 1000 x 750 x 256 voxels —
    2.112 GB of volume traffic + 12 MB of P2 maps = 2.124 GB in the timed call (the counters say 2.121 GB, profiles/r04_sgm_pmc.json); SURVEY
    section 8(d)'s figure with the texels: 2.160 GB, the numerator of `frac_with_p2_map`, whose denominator includes the map kernel."""
    sys.path.insert(0, ROOT)
    import bench
    b = bench.sgm_algorithmic_bytes([(1000, 750)], 256)
    assert b["timed_call"] == 11.0 * 1000 * 750 * 256 + 16.0 * 1000 * 750 == 2.124e9
    assert b["survey"] == 11.0 * 1000 * 750 * 256 + 64.0 * 1000 * 750 == 2.16e9
    assert bench.sgm_algorithmic_bytes([(1000, 750)], 256, prepared=False)["timed_call"] == b["survey"]
    # configuration 5: 16 tile volumes of 416 x 288 in one call
    b5 = bench.sgm_algorithmic_bytes([(416, 288)] * 16, 256)
    assert b5["timed_call"] == 16 * (11.0 * 416 * 288 * 256 + 16.0 * 416 * 288)
    # and the committed summary of the counters agrees with the timed call's bytes to 2 % (2 launches per volume)
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_sgm_pmc.json")))
    assert abs(2 * rec["hbm_bytes_per_launch"] / b["timed_call"] - 1.0) < 0.02
