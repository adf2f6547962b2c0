"""GPU tests of the C++ host + CLI (aliceVision_depthMapEstimation): .sfm + EXR in, EXR out, on an MI355X.

  * single tile: the CLI's maps must equal, bit for bit, what the ctypes harness (alicevision_amd/pipeline.py, itself parity
    tested stage by stage against the oracle in test_gpu_parity.py) computes from the same planes / T cameras — this pins the
    C++ sequencing of Sgm / Refine / DepthMapEstimator and the EXR writer;
  * against the CPU oracle end to end (depth RMSE < 1e-3, BASELINE.json) and against the analytic ground truth;
  * tiled run (2 x 2 tiles, batched SGM aggregation, weighted merge): valid everywhere and as close to the ground truth as
    the single-tile run.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from alicevision_amd import abi, exr_io, scene_io
from alicevision_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
W, H, NVIEWS = 640, 480, 5
OPT_ITERS = 20


def run_cli(args, check=True):
    r = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, text=True, timeout=300)  # a run takes seconds
    if check:
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    assert os.path.exists(CLI), "host CLI not built (python -c 'import __graft_entry__ as g; g.build()')"
    d = str(tmp_path_factory.mktemp("scene"))
    sc = make_scene(NVIEWS, W, H, seed=5, baseline=0.9, amp=0.6)
    lms = scene_io.sample_landmarks(sc, 500, amp=0.6)
    os.makedirs(os.path.join(d, "images"), exist_ok=True)
    sfm = os.path.join(d, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), f)
    for i in range(NVIEWS):
        im = sc.images[i].numpy()
        exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]},
                         compression=0)
    return sc, sfm, os.path.join(d, "images"), d


def common_args(sfm, img, out):
    return ["-i", sfm, "--imagesFolder", img, "-o", out, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 96,
            "--colorOptimizationNbIterations", OPT_ITERS, "-v", "warning"]


def read_maps(out, vid=None):
    vid = scene_io.view_id(0) if vid is None else vid
    dm, dinfo = exr_io.read_exr(os.path.join(out, "%d_depthMap.exr" % vid))
    sm, sinfo = exr_io.read_exr(os.path.join(out, "%d_simMap.exr" % vid))
    return dm["Y"], sm["Y"], dinfo, sinfo


def test_single_tile_equals_harness_and_oracle(dataset):
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc, sfm, img, d = dataset
    out = os.path.join(d, "out_single")
    args = common_args(sfm, img, out)
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    assert t0["rc"] == 0 and t0["nbTiles"] == 1 and len(t0["depths"]) > 12, t0
    assert args[-2:] == ["-v", "warning"]
    log = run_cli(args[:-2] + ["-v", "info"])
    # the Refine sweep's outlier lists were never full (avdm_refine_outlier_refused, logged per worker: VERDICT r5 weak #8)
    assert "Refine outlier lists: no unit refused" in (log.stdout + log.stderr)
    depth, sim, dinfo, sinfo = read_maps(out)
    assert depth.shape == (H, W)
    assert dinfo["channel_types"]["Y"] == 2 and sinfo["channel_types"]["Y"] == 1  # float depth, half sim (mapIO.cpp:517-526)
    assert exr_io.attr_value(dinfo, "AliceVision:downscale") == 1
    assert exr_io.attr_value(dinfo, "AliceVision:nbDepthValues") == int((depth > 0).sum())
    P = exr_io.attr_value(dinfo, "AliceVision:P").reshape(4, 4)[:3]
    Pref = sc.K @ np.concatenate([sc.R[0], (-sc.R[0] @ sc.C[0])[:, None]], axis=1)
    assert np.allclose(P, Pref, atol=1e-6), (P, Pref)

    # the harness on the same plan
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS)
    depths = np.asarray(t0["depths"], np.float32)
    ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges)
    got = h.run_refine(0, t0["refineTCams"]).cpu().numpy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())
    assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim)

    # the oracle end to end on the same plan (BASELINE.json: depth RMSE < 1e-3)
    o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref)
    with oracle.well_posed():
        o.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges)
        want = o.run_refine(0, t0["refineTCams"])
    both = (want[..., 0] > 0) & (depth > 0)
    assert ((want[..., 0] > 0) != (depth > 0)).mean() < 0.005
    err = np.sort((depth - want[..., 0])[both] ** 2)
    rmse = float(np.sqrt(err[: int(0.995 * err.size)].mean()))
    assert rmse < 1e-3, rmse

    # analytic ground truth (distance along the ray)
    gt = sc.gt_depth.numpy()
    inner = np.zeros_like(both)
    inner[16:-16, 16:-16] = True
    m = both & inner
    assert m.mean() > 0.6
    rel = np.abs(depth - gt)[m] / gt[m]
    assert np.median(rel) < 2e-3, float(np.median(rel))


def test_reference_arithmetic_flags_equal_the_oracle(dataset):
    """`--sgmReferenceArithmetic 1 --refineReferenceArithmetic 1` (not flags of the reference: the product's parity mode): the PROGRAM's maps — files
    in, EXR files out — equal the harness's in the same mode bit for bit (the flags reach both sweeps), and against the literal oracle (= the
    reference's own code compiled for the CPU) on the program's own plan, NO trimming, the Refine stage's map is IDENTICAL; what separates the
    final depth maps is the colour optimisation's tolerance class alone.  The run logs that no Refine outlier-list unit was refused."""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc, sfm, img, d = dataset
    out = os.path.join(d, "out_refarith")
    args = common_args(sfm, img, out) + ["--sgmReferenceArithmetic", 1, "--refineReferenceArithmetic", 1]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    assert args[-6:-4] == ["-v", "warning"]
    log = run_cli(args[:-6] + ["-v", "info"] + args[-4:])
    assert "no unit refused" in (log.stdout + log.stderr)
    depth, sim, _, _ = read_maps(out)
    depths = np.asarray(t0["depths"], np.float32)
    ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
    # the harness in the same mode, on the same plan
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"], referenceArithmetic=1)
    ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS, referenceArithmetic=1)
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges)
    got = h.run_refine(0, t0["refineTCams"]).cpu().numpy()
    refined_h = h.refined.cpu().numpy().copy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())
    assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim)
    # the literal oracle (the default parameter structs: the oracle ignores the mode field)
    o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref)
    o.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges)
    want = o.run_refine(0, t0["refineTCams"])
    assert np.array_equal(o.sgm_depth_sim[..., 0], h.sgm_depth_sim.cpu().numpy()[..., 0])       # the winner-take-all depths of the SGM stage
    assert np.array_equal(o.refined.view(np.uint32), refined_h.view(np.uint32))                 # the Refine stage's (depth, sim) map, every bit
    assert np.array_equal(want[..., 0] > 0, depth > 0)
    both = depth > 0
    err = (depth - want[..., 0])[both].astype(np.float64)
    rmse = float(np.sqrt((err ** 2).mean()))
    assert rmse < 1e-3, rmse  # BASELINE's bar, untrimmed; measured: see DESIGN.md section 2 (the colour optimisation's tolerance class)


def test_tiled_run_merges(dataset):
    sc, sfm, img, d = dataset
    out1, out4 = os.path.join(d, "out_single_b"), os.path.join(d, "out_tiled")
    base = common_args(sfm, img, out1) + ["--autoAdjustSmallImage", 0]
    run_cli(base + ["--tileBufferWidth", 640, "--tileBufferHeight", 640])
    args4 = common_args(sfm, img, out4) + ["--autoAdjustSmallImage", 0, "--tileBufferWidth", 416, "--tileBufferHeight", 352, "--tilePadding", 32]
    plan = json.loads(run_cli(args4 + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    assert len(plan["tiles"]) == 4, [t["roi"] for t in plan["tiles"]]
    run_cli(args4 + ["--exportIntermediateDepthSimMaps", 1, "--exportIntermediateNormalMaps", 1, "--exportIntermediateVolume9pCsv", 1])
    # 9-point CSV dumps of the volumes (volumeIO.cpp:28-146), one file per tile and stage: name line + p1..p9 with one value per plane
    csvs = sorted(f for f in os.listdir(out4) if f.endswith(".csv"))
    assert len([f for f in csvs if "_9p_sgm_" in f]) == 4 and len([f for f in csvs if "_9p_refine_" in f]) == 4, csvs
    for f in csvs:
        lines = open(os.path.join(out4, f)).read().strip().split("\n")
        blocks = [ln for ln in lines if not ln.startswith("p")]
        assert blocks == (["beforeFiltering", "afterFiltering"] if "_sgm_" in f else ["afterRefine"]), (f, blocks)
        pts = [ln for ln in lines if ln.startswith("p")]
        assert len(pts) == 9 * len(blocks)
        vals = [float(v) for v in pts[0].split(";")[1:] if v]
        assert len(vals) >= 12 and (all(0 <= v <= 255 for v in vals) if "_sgm_" in f else len(vals) == 31)
    # normal maps (SGM resolution, refined, final): merged from the tiles, unit vectors facing the camera where a depth exists
    vid = scene_io.view_id(0)
    for name, shape in (("normalMap_sgm", (H // 4, W // 4)), ("normalMap_refinedFused", (H, W)), ("normalMap", (H, W))):
        nm, ninfo = exr_io.read_exr(os.path.join(out4, "%d_%s.exr" % (vid, name)))
        assert set(nm) == {"R", "G", "B"} and nm["R"].shape == shape, (name, nm["R"].shape)
        n = np.stack([nm["R"], nm["G"], nm["B"]], -1)
        core = n[shape[0] // 4: -shape[0] // 4, shape[1] // 4: -shape[1] // 4]
        good = np.abs(np.linalg.norm(core, axis=-1) - 1.0) < 2e-2  # half storage, blended at the tile seams
        assert good.mean() > 0.7, (name, float(good.mean()))
        # the surface faces the camera: the normal points against the viewing direction (camera z axis = row 2 of R)
        assert (core[good] @ sc.R[0][2] < 0).mean() > 0.95, name
    d1, s1, _, _ = read_maps(out1)
    d4, s4, info4, _ = read_maps(out4)
    assert d4.shape == d1.shape == (H, W)
    # the merged file is a whole image again and the tile files of the intermediate maps were merged and removed
    assert info4["data_window"] == (0, 0, W - 1, H - 1)
    leftovers = [f for f in os.listdir(out4) if f.count("_") >= 3 and f.endswith(".exr") and f.split("_")[-1][0].isdigit()]  # (the CSVs stay per tile)
    assert not leftovers, leftovers
    assert os.path.exists(os.path.join(out4, "%d_depthMap_sgm.exr" % scene_io.view_id(0)))
    assert os.path.exists(os.path.join(out4, "%d_depthMap_refinedFused.exr" % scene_io.view_id(0)))
    gt = sc.gt_depth.numpy()
    inner = np.zeros(gt.shape, bool)
    inner[16:-16, 16:-16] = True
    for dm in (d1, d4):
        m = (dm > 0) & inner
        assert m.mean() > 0.6, m.mean()
        assert np.median(np.abs(dm - gt)[m] / gt[m]) < 2e-3
    # tiles see a different neighbourhood near their borders only: most pixels agree closely with the single-tile run
    m = (d1 > 0) & (d4 > 0) & inner
    assert np.median(np.abs(d1 - d4)[m]) < 1e-3


def _merge_tiles(tile_maps, rois, pad, ss):
    """mapIO.cpp's weighted tile merge (addSingleTileMapWeighted) of per-tile maps at 1 / ss resolution; also returns how many tiles cover a pixel"""
    from oracle import host_oracle as ho
    out = np.zeros((-(-H // ss), -(-W // ss)), np.float32)
    cover = np.zeros(out.shape, np.int32)
    for m, roi in zip(tile_maps, rois):
        wmap, (bx, ex, by, ey) = ho.tile_weight_map(tuple(roi), W, H, pad, ss)
        out[by:ey, bx:ex] += (m.astype(np.float32) * wmap).astype(np.float32)
        cover[by:ey, bx:ex] += 1
    return out, cover


def test_tiled_run_equals_harness_and_oracle_per_tile(dataset):
    """The default workflow is TILED, and tiles that do not start at the image origin are where the extent of the SGM aggregation matters:
    the reference's volumes are allocated for the tile buffer and its path aggregation walks that extent (deviceSimilarityVolume.cu:278-283).
    The program's 2 x 2-tile run, tile by tile on the program's own plan, against (1) the harness with the same buffer (bit for bit: pins
    host/Sgm.cpp's layout and descriptors, the batched aggregation and the merge) and (2) OracleDepthMap(tile_buffer=...), which
    tests/test_oracle_ref.py holds to the reference's own Sgm.cpp / Refine.cpp bit for bit (depth RMSE < 1e-3, BASELINE.json)."""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc, sfm, img, d = dataset
    out4 = os.path.join(d, "out_tiled_pin")
    buf, pad = (416, 352), 32
    args4 = common_args(sfm, img, out4) + ["--autoAdjustSmallImage", 0, "--tileBufferWidth", buf[0], "--tileBufferHeight", buf[1], "--tilePadding", pad,
                                           "--exportIntermediateDepthSimMaps", 1]
    plan = json.loads(run_cli(args4 + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    tiles = plan["tiles"]
    assert len(tiles) == 4 and sum(1 for t in tiles if t["roi"][0] > 0 or t["roi"][2] > 0) == 3
    log = run_cli(args4 + ["-v", "info"])  # (the later -v wins)
    assert "Refine outlier lists: no unit refused" in (log.stdout + log.stderr)  # the tiled run too
    depth, sim, _, _ = read_maps(out4)
    vid = scene_io.view_id(0)
    sgm_depth = exr_io.read_exr(os.path.join(out4, "%d_depthMap_sgm.exr" % vid))[0]["Y"]

    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS)
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    o_pyr = None
    got_final, got_sgm, want_final, rois = [], [], [], []
    for t in tiles:
        roi = tuple(t["roi"])
        depths = np.asarray(t["depths"], np.float32)
        ranges = [(a, a + n) for a, n in t["depthsTcLimits"]]
        h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=roi, tile_buffer=buf)
        _, dsm = h.run_sgm(0, t["sgmTCams"], depths, tc_ranges=ranges)
        got_sgm.append(dsm[..., 0].cpu().numpy().copy())
        got_final.append(h.run_refine(0, t["refineTCams"]).cpu().numpy().copy())
        o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref, roi=roi, pyramids=o_pyr)
        o_pyr = o.pyr
        with oracle.well_posed():
            o.run_sgm(0, t["sgmTCams"], depths, tc_ranges=ranges, tile_buffer=buf)
            want_final.append(o.run_refine(0, t["refineTCams"], tile_buffer=buf).copy())
        rois.append(roi)

    # (1) the program == the harness: exactly where one tile alone contributes with weight 1, to float rounding of the weighted sum elsewhere
    hd, cover = _merge_tiles([g[..., 0] for g in got_final], rois, pad, 1)
    single = cover == 1
    assert single.mean() > 0.5
    assert np.array_equal(depth[single], hd[single]), float(np.abs(depth - hd)[single].max())
    assert np.allclose(depth, hd, rtol=1e-6, atol=1e-6)
    hs, cover4 = _merge_tiles(got_sgm, rois, pad, 4)
    assert np.array_equal(sgm_depth[cover4 == 1], hs[cover4 == 1])
    # and the extent is what the program computes with: over the tiles' ROIs alone (rounds 1-2) the offset tiles come out differently
    h0 = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=rois[3])
    t3 = tiles[3]
    _, dsm0 = h0.run_sgm(0, t3["sgmTCams"], np.asarray(t3["depths"], np.float32), tc_ranges=[(a, a + n) for a, n in t3["depthsTcLimits"]])
    assert (dsm0[..., 0].cpu().numpy() != got_sgm[3]).mean() > 1e-3

    # (2) the program against the oracle, tile by tile over the same buffer extent, merged like the program merges
    wd, _ = _merge_tiles([w[..., 0] for w in want_final], rois, pad, 1)
    both = (wd > 0) & (depth > 0)
    assert ((wd > 0) != (depth > 0)).mean() < 0.005
    err = np.sort((depth - wd)[both] ** 2)
    rmse, untrimmed = float(np.sqrt(err[: int(0.995 * err.size)].mean())), float(np.sqrt(err.mean()))
    print("merged map: rmse over the best 99.5 %%: %.3e, untrimmed: %.3e" % (rmse, untrimmed))
    assert rmse < 1e-3, (rmse, untrimmed)
    # untrimmed as well: an ~800-pixel image (one plane step is ~1e-2 depth units); at 12 MP the default tiles meet 1e-3 untrimmed
    # (tests/test_gpu_parity.py::test_parity_of_default_tiles_at_12mp)
    assert untrimmed < 3e-3, (rmse, untrimmed)
    # per tile, offset tiles on their own (no merge in between)
    for g, w, roi in zip(got_final, want_final, rois):
        m = (g[..., 0] > 0) & (w[..., 0] > 0)
        e = np.sort((g[..., 0] - w[..., 0])[m] ** 2)
        r, ru = float(np.sqrt(e[: int(0.995 * e.size)].mean())), float(np.sqrt(e.mean()))
        print("tile", roi, "rmse over the best 99.5 %%: %.3e, untrimmed: %.3e" % (r, ru))
        assert r < 1e-3 and ru < 3e-3, (roi, r, ru)


def test_cli_fails_loudly_without_inputs(dataset):
    sc, sfm, img, d = dataset
    r = run_cli(["-i", os.path.join(d, "missing.sfm"), "--imagesFolder", img, "-o", os.path.join(d, "o")], check=False)
    assert r.returncode == 1 and "cannot be read" in (r.stdout + r.stderr)
    r = run_cli(["-i", sfm, "--imagesFolder", img], check=False)
    assert r.returncode == 1 and "required" in r.stderr


def test_sweep_only_configuration(dataset):
    """BASELINE cfg2 shape ("plane sweep only, no SGM"): --sgmOptimizeVolume 0 --useRefine 0 makes the SGM-resolution WTA map the
    final output (DepthMapEstimator.cpp:285,314-315,441-442); it must equal the harness' run_sgm(optimize=False)."""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sfm, img, d = dataset
    out = os.path.join(d, "out_sweep")
    args = common_args(sfm, img, out) + ["--sgmOptimizeVolume", 0, "--useRefine", 0, "--autoAdjustSmallImage", 0, "--tileBufferWidth", 640,
                                         "--tileBufferHeight", 640, "--maxTCams", 4]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    assert plan["sgmScale"] == 2 and plan["sgmStepXY"] == 2
    run_cli(args)
    depth, sim, dinfo, _ = read_maps(out)
    assert depth.shape == (H // 4, W // 4) and exr_io.attr_value(dinfo, "AliceVision:downscale") == 4
    sgm = abi.SgmParams.default()
    ref = abi.RefineParams.default()
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
    _, ds = h.run_sgm(0, t0["sgmTCams"], np.asarray(t0["depths"], np.float32), tc_ranges=ranges, optimize=False)
    ds = ds.cpu().numpy()
    assert np.array_equal(ds[..., 0], depth)
    assert np.array_equal(ds[..., 1].astype(np.float16).astype(np.float32), sim)
    # the raw sweep already finds the surface on most pixels
    gt = sc.gt_depth.numpy()[::4, ::4]
    m = depth > 0
    m[:4] = m[-4:] = False
    m[:, :4] = m[:, -4:] = False
    assert m.mean() > 0.5 and np.median(np.abs(depth - gt)[m] / gt[m]) < 2e-2


def test_non_default_refine_switches(dataset):
    """non-default Refine switches through the program: --refineInterpolateMiddleDepth 1 (bilinear upscale of the SGM depth,
    deviceDepthSimilarityMapKernels.cuh:276-383), --colorOptimizationEnabled 0 (Refine.cpp:156-160: the refined + fused map is the
    output), --refineEnabled 0 (Refine.cpp:143-151: the upscaled SGM depth goes straight to the optimisation) and
    --sgmUseConsistentScale / --refineUseConsistentScale 1 (Patch.cuh:250-308), a custom patch pattern (Patch.cuh:598-773) — each
    must equal the harness with the same switches"""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sfm, img, d = dataset
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    results = {}
    for name, extra, kw, run_kw in (("interp", ["--refineInterpolateMiddleDepth", 1], dict(interpolateMiddleDepth=1), {}),
                                    ("noopt", ["--colorOptimizationEnabled", 0], {}, dict(optimize_enabled=False)),
                                    ("norefine", ["--refineEnabled", 0], {}, dict(refine_enabled=False)),
                                    ("cscale", ["--sgmUseConsistentScale", 1, "--refineUseConsistentScale", 1], dict(useConsistentScale=1), {}),
                                    ("pattern", ["--sgmUseCustomPatchPattern", 1, "--refineUseCustomPatchPattern", 1, "--customPatchPatternSubparts",
                                                 "circle:4:16:0:0.5", "full:2:0:1:0.5"], dict(useCustomPatchPattern=1), {})):
        out = os.path.join(d, "out_" + name)
        args = common_args(sfm, img, out) + extra
        plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
        t0 = plan["tiles"][0]
        run_cli(args)
        depth, sim, _, _ = read_maps(out)
        sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"], useConsistentScale=kw.get("useConsistentScale", 0),
                                    useCustomPatchPattern=kw.get("useCustomPatchPattern", 0))
        if kw.get("useCustomPatchPattern"):
            abi.build_custom_patch_pattern([("circle", 4, 16, 0, 0.5), ("full", 2, 0, 1, 0.5)], False)
        ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS, **kw)
        h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
        ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
        h.run_sgm(0, t0["sgmTCams"], np.asarray(t0["depths"], np.float32), tc_ranges=ranges)
        got = h.run_refine(0, t0["refineTCams"], **run_kw).cpu().numpy()
        assert np.array_equal(got[..., 0], depth), (name, float(np.abs(got[..., 0] - depth).max()))
        assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim), name
        results[name] = depth
    base = read_maps(os.path.join(d, "out_single"))[0] if os.path.exists(os.path.join(d, "out_single")) else None
    if base is not None:  # the switches change the result, and not by much
        for name, depth in results.items():
            both = (base > 0) & (depth > 0)
            assert both.mean() > 0.5 and not np.array_equal(base, depth)
            assert np.median(np.abs(base - depth)[both] / base[both]) < 3e-2, name


def test_default_process_downscale(tmp_path):
    """the program's default --downscale 2 (main_depthMapEstimation.cpp:75): images are halved on load (mvsUtils/fileIO.cpp:389-443), the
    maps come out at half the image size with AliceVision:downscale = 2 and still find the surface"""
    w, h = 1280, 960
    sc = make_scene(4, w, h, seed=5, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    sfm, img = scene_io.write_scene(sc, d, n_landmarks=10, compression=0)
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 500, amp=0.6), img), f)
    out = os.path.join(d, "out")
    run_cli(["-i", sfm, "--imagesFolder", img, "-o", out, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 96, "--colorOptimizationNbIterations",
             OPT_ITERS, "-v", "warning"])
    depth, sim, dinfo, _ = read_maps(out)
    assert depth.shape == (h // 2, w // 2)
    assert exr_io.attr_value(dinfo, "AliceVision:downscale") == 2
    gt = sc.gt_depth.numpy()[::2, ::2]
    m = depth > 0
    m[:16] = m[-16:] = False
    m[:, :16] = m[:, -16:] = False
    assert m.mean() > 0.6
    assert np.median(np.abs(depth - gt)[m] / gt[m]) < 3e-3

    # the same plan through the harness on images resized by the ORACLE's restatement of OpenImageIO's default filter (fileIO.cpp:432-441):
    # the program resizes on the device (avdm_image_resize, bit-exact against that restatement), so the maps must agree bit for bit
    import ctypes as C
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    args = ["-i", sfm, "--imagesFolder", img, "-o", out, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 96, "--colorOptimizationNbIterations",
            OPT_ITERS, "-v", "warning"]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    assert t0["nbTiles"] == 1
    olib = oracle.load()
    small = []
    for i in range(4):
        src = np.ascontiguousarray(sc.images[i].numpy())
        dst = np.zeros((h // 2, w // 2, 4), np.float32)
        assert olib.avo_image_resize(oracle.ptr(dst), (w // 2) * 16, w // 2, h // 2, oracle.ptr(src), w * 16, w, h, 4) == 0
        small.append(torch.from_numpy(dst))
    K2 = sc.K.copy()
    K2[:2] /= 2.0  # MultiViewParams.cpp:288-291: P rows 0-1 divided by the process downscale
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS)
    depths = np.asarray(t0["depths"], np.float32)
    ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(small[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(4)]
    hn = DepthMapTile(pyr, K2, sc.R, sc.C, sgm, ref)
    hn.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges)
    got = hn.run_refine(0, t0["refineTCams"]).cpu().numpy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())


def test_multi_worker_exchange_equals_single_worker(dataset):
    """the in-process multi-GPU path (computeOnMultiGPUs.cpp: one thread per device, R cameras dealt round-robin, every view decoded and
    converted by ONE worker and copied device-to-device to the others through PyramidExchange): with AVDM_FAKE_DEVICES=2 the two workers
    share the box's one GPU, so the whole path runs here, peer copies included.  The maps of every camera must be byte-identical to the
    single-worker run — each worker computed from pyramids it RECEIVED for the views it does not own."""
    sc, sfm, img, d = dataset
    base = ["-i", sfm, "--imagesFolder", img, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 4, "--sgmMaxDepths", 64,
            "--colorOptimizationNbIterations", 5, "--tileBufferWidth", 400, "--tileBufferHeight", 300, "--tilePadding", 32, "-v", "info"]
    out1, out2 = os.path.join(d, "out_w1"), os.path.join(d, "out_w2")
    run_cli(base + ["-o", out1, "--nbGPUs", 1])
    env = dict(os.environ, AVDM_FAKE_DEVICES="2")
    import re
    # repeated: two host threads on one device is where a runtime deadlock showed up once in ~25 runs (stream-ordered allocations in one
    # thread, event calls in the other; profiles/r02_multiworker_hang.md) — a run takes about a second, a hang is cut after 90
    for rep in range(10):
        r = subprocess.run([CLI] + [str(a) for a in base + ["-o", out2, "--nbGPUs", 2]], capture_output=True, text=True, timeout=90, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        log = r.stdout + r.stderr
        assert "workers: 2" in log and "Pyramid exchange:" in log, log[-2000:]
        m = re.search(r"Pyramid exchange: (\d+) views converted once, (\d+) peer copies", log)
        assert m and int(m.group(1)) == NVIEWS and int(m.group(2)) >= 2, log[-2000:]  # every view decoded once; the others travelled
        for i in range(4):
            vid = scene_io.view_id(i)
            for name in ("%d_depthMap.exr" % vid, "%d_simMap.exr" % vid):
                a = open(os.path.join(out1, name), "rb").read()
                b = open(os.path.join(out2, name), "rb").read()
                assert a == b, (rep, name)


def test_png_input_is_decoded_on_the_device(dataset, tmp_path):
    """<viewId>.png in --imagesFolder (the reference reads any format OpenImageIO decodes, mvsUtils/fileIO.cpp:386-446; here PNG next to
    OpenEXR): the host inflates and un-filters, the INTEGER samples go to the device and become linear float RGBA there
    (avdm_image_decode_integer).  The program's maps equal, bit for bit, the harness run on the images the ORACLE decodes from the same
    samples; 16-bit RGB files written with the sRGB encoding of the scene's linear images."""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc, sfm, img, d = dataset
    png_dir = str(tmp_path / "png")
    os.makedirs(png_dir)
    tool = os.path.join(ROOT, "alicevision_amd", "bin", "avdm_host_tool")
    olib = oracle.load()
    decoded = []
    for i in range(NVIEWS):
        lin = sc.images[i].numpy()[..., :3].astype(np.float64)
        enc = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055)
        q = np.clip(np.rint(enc * 65535.0), 0, 65535).astype(np.uint16)
        raw = str(tmp_path / "v.raw")
        q.tofile(raw)
        r = subprocess.run([tool, "png-write", raw, os.path.join(png_dir, "%d.png" % scene_io.view_id(i)), str(W), str(H), "3", "16"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = np.zeros((H, W, 4), np.float32)
        assert olib.avo_image_decode_integer(oracle.ptr(out), W * 16, oracle.ptr(q), W * 6, W, H, 3, 16, 1) == 0
        decoded.append(out)
        assert np.abs(out[..., :3] - lin).max() < 2e-5  # 16-bit quantisation of the encoded value
    out_dir = os.path.join(d, "out_png")
    args = ["-i", sfm, "--imagesFolder", png_dir, "-o", out_dir, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 64,
            "--colorOptimizationNbIterations", 5, "-v", "warning"]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    run_cli(args)
    depth, sim, _, _ = read_maps(out_dir)
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=5)
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(torch.from_numpy(decoded[i]).cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], np.asarray(t0["depths"], np.float32), tc_ranges=[(a, a + n) for a, n in t0["depthsTcLimits"]])
    got = h.run_refine(0, t0["refineTCams"]).cpu().numpy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())
    assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim)
    # two files for one view are refused like in the reference (MultiViewParams.cpp:96-100)
    import shutil
    shutil.copy(os.path.join(img, "%d.exr" % scene_io.view_id(0)), png_dir)
    r = run_cli(args + ["--dryRun", 1], check=False)
    assert r.returncode == 1 and "Ambiguous" in (r.stdout + r.stderr)


def test_multi_worker_exchange_budget(dataset):
    """the exchange's residency is bounded: with a budget that holds ONE 640 x 480 pyramid per worker
    (AVDM_EXCHANGE_BUDGET_MB=4: a pyramid is 3.3 MB) the owners decline their other views, every worker that needs one of those decodes
    and converts it itself like the reference does — and every camera's maps are still byte-identical to the single-worker run."""
    import re
    sc, sfm, img, d = dataset
    base = ["-i", sfm, "--imagesFolder", img, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 4, "--sgmMaxDepths", 64,
            "--colorOptimizationNbIterations", 5, "--tileBufferWidth", 400, "--tileBufferHeight", 300, "--tilePadding", 32, "-v", "info"]
    out1, out2 = os.path.join(d, "out_w1"), os.path.join(d, "out_w2_budget")
    if not os.path.exists(os.path.join(out1, "%d_depthMap.exr" % scene_io.view_id(0))):
        run_cli(base + ["-o", out1, "--nbGPUs", 1])
    env = dict(os.environ, AVDM_FAKE_DEVICES="2", AVDM_EXCHANGE_BUDGET_MB="4")
    r = subprocess.run([CLI] + [str(a) for a in base + ["-o", out2, "--nbGPUs", 2]], capture_output=True, text=True, timeout=90, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    log = r.stdout + r.stderr
    assert "residency budget 4 MB per worker" in log, log[-2000:]
    m = re.search(r"Pyramid exchange: (\d+) views converted once, (\d+) peer copies .* (\d+) views over the residency budget", log)
    assert m, log[-2000:]
    assert int(m.group(1)) == 2 and int(m.group(3)) == NVIEWS - 2, m.groups()  # one resident pyramid per worker, the rest declined
    for i in range(4):
        vid = scene_io.view_id(i)
        for name in ("%d_depthMap.exr" % vid, "%d_simMap.exr" % vid):
            assert open(os.path.join(out1, name), "rb").read() == open(os.path.join(out2, name), "rb").read(), name


def test_prepare_dense_scene_undistorts_and_feeds_the_estimation(tmp_path):
    """aliceVision_prepareDenseScene (the program before this stage): views of a camera with radial distortion come out undistorted —
    bit for bit the oracle's camera::UndistortImage — with the camera in the image metadata, and aliceVision_depthMapEstimation plans
    from those files exactly as from the SfMData."""
    import ctypes as C
    from oracle import oracle
    exe = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_prepareDenseScene")
    w, h = 320, 240
    sc = make_scene(4, w, h, seed=9, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    sfm, img = scene_io.write_scene(sc, d, n_landmarks=10, compression=0)
    sd = scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 300, amp=0.6), img)
    k = (0.09, -0.04, 0.012)
    sd["intrinsics"][0]["distortionType"] = "radialk3"
    sd["intrinsics"][0]["distortionParams"] = ["%.17g" % v for v in k]
    with open(sfm, "w") as f:
        json.dump(sd, f)
    out = os.path.join(d, "prepared")
    r = subprocess.run([exe, "-i", sfm, "-o", out, "--saveMatricesTxtFiles", "1", "-v", "info"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    olib = oracle.load()
    fx = float(sc.K[0, 0])
    cam = abi.Intrinsic(width=w, height=h, scale_x=fx, scale_y=fx, offset_x=float(sc.K[0, 2]) - w / 2.0, offset_y=float(sc.K[1, 2]) - h / 2.0,
                        distortion_model=abi.DISTORTION_RADIALK3, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0, 0, 0, 0)
    for i in range(4):
        vid = scene_io.view_id(i)
        ch, info = exr_io.read_exr(os.path.join(out, "%d.exr" % vid))
        got = np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1)
        src = np.ascontiguousarray(sc.images[i].numpy())
        want = np.zeros_like(src)
        assert olib.avo_image_undistort(oracle.ptr(want), w * 16, oracle.ptr(src), w * 16, C.byref(cam), C.byref(fill)) == 0
        assert np.array_equal(got, want), (i, float(np.abs(got - want).max()))
        assert not np.array_equal(got, src)
        P = exr_io.attr_value(info, "AliceVision:P").reshape(4, 4)[:3]
        Pref = sc.K @ np.concatenate([sc.R[i], (-sc.R[i] @ sc.C[i])[:, None]], axis=1)
        assert np.allclose(P, Pref, atol=1e-6)
        assert exr_io.attr_value(info, "AliceVision:downscale") == 1
        Ptxt = np.loadtxt(os.path.join(out, "%d_P.txt" % vid))
        assert np.allclose(Ptxt, Pref, rtol=1e-8, atol=1e-8)
    # the estimation program on the prepared folder: same plan as from the SfMData with the original images
    a = ["-i", sfm, "--imagesFolder", out, "-o", os.path.join(d, "o1"), "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 48, "--dryRun", 1]
    b = ["-i", sfm, "--imagesFolder", img, "-o", os.path.join(d, "o2"), "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 48, "--dryRun", 1]
    pa = json.loads(run_cli(a).stdout.strip().splitlines()[-1])
    pb = json.loads(run_cli(b).stdout.strip().splitlines()[-1])
    assert pa["tiles"][0]["sgmTCams"] == pb["tiles"][0]["sgmTCams"]
    assert np.allclose(pa["tiles"][0]["depths"], pb["tiles"][0]["depths"], rtol=1e-5)


def test_alembic_scene_gives_the_same_maps_as_the_sfm_scene(dataset, tmp_path):
    """`-i sfm.abc` — what Meshroom's StructureFromMotion node hands to the DepthMap node (sfmDataIO::load dispatches on the extension,
    sfmDataIO.cpp:106-131) — read by host/alembic.cpp: the maps of a reference camera equal those of the run on the .sfm file the
    archive was written from to float precision (cameras identical to the last ulp of the pose inverse; the archive holds the landmarks,
    which seed the depth-plane list, as float32 like the reference's exporter)."""
    sc, sfm, img, d = dataset
    tool = os.path.join(ROOT, "alicevision_amd", "bin", "avdm_host_tool")
    abc_path = str(tmp_path / "sfm.abc")
    r = subprocess.run([tool, "sfm-to-abc", sfm, abc_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    maps = []
    for k, scene in enumerate((sfm, abc_path)):
        out_dir = str(tmp_path / ("out%d" % k))
        run_cli(["-i", scene, "--imagesFolder", img, "-o", out_dir, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 64,
                 "--colorOptimizationNbIterations", 5, "-v", "warning"])
        maps.append(read_maps(out_dir)[:2])
    valid = maps[0][0] > 0
    assert valid.mean() > 0.5
    # the depth-plane list is seeded from the landmarks (float32 in the archive, double in the .sfm): the same planes up to 1e-7 relative
    assert np.array_equal(valid, maps[1][0] > 0)
    np.testing.assert_allclose(maps[1][0][valid], maps[0][0][valid], rtol=2e-5)
    assert float(np.abs(maps[1][1] - maps[0][1])[valid].mean()) < 1e-3


def test_volume_exports_as_alembic_point_clouds(dataset, tmp_path):
    """--exportIntermediateVolumes / --exportIntermediateCrossVolumes / --exportIntermediateTopographicCutVolumes (volumeIO.cpp:148-441):
    the point clouds the program saves (Alembic archives like the reference's sfmDataIO::save(.., STRUCTURE), read back here by the
    program's own reader) equal a numpy restatement of the reference's loops over the harness's volumes of the same plan."""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sfm, img, d = dataset
    tool = os.path.join(ROOT, "alicevision_amd", "bin", "avdm_host_tool")
    out = str(tmp_path / "out")
    # The reference samples the ALLOCATED volume (tile buffer / (scale * step) cells a side, initialised to 255 = "skipped"), whose centre
    # row and column are the cross; with the default 1024 buffer the centre row of this 640 x 480 image's volume lies below the image.
    # A 640 x 640 buffer (still one tile: hasOnlyOneTile compares the buffer HEIGHT with both image sides, TileParams.hpp:35-38) puts
    # both inside.
    BUF = 640
    args = common_args(sfm, img, out) + ["--exportIntermediateVolumes", 1, "--exportIntermediateCrossVolumes", 1,
                                         "--exportIntermediateTopographicCutVolumes", 1, "--tileBufferWidth", BUF, "--tileBufferHeight", BUF]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    run_cli(args)
    vid = scene_io.view_id(0)

    def cloud(name):
        r = subprocess.run([tool, "sfm-dump", os.path.join(out, "%d_%s.abc" % (vid, name))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        doc = json.loads(r.stdout)
        assert doc["views"] == [] and doc["poses"] == []
        lm = doc["landmarks"]
        assert [l["id"] for l in lm] == list(range(len(lm))) and all(l["obs"] == [] for l in lm)
        return np.array([l["X"] for l in lm]).reshape(-1, 3), np.array([l["rgb"] for l in lm]).reshape(-1, 3)

    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=OPT_ITERS)
    depths = np.asarray(t0["depths"], np.float32)
    ranges = [(a, a + n) for a, n in t0["depthsTcLimits"]]
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], depths, tc_ranges=ranges, keep_raw=True)
    ss = plan["sgmScale"] * plan["sgmStepXY"]
    K, R, C0 = sc.K.astype(np.float64), sc.R[0].astype(np.float64), sc.C[0].astype(np.float64)
    iCam = np.linalg.inv(K @ R)
    n = R.T @ np.array([0.0, 0.0, 1.0])
    n /= np.linalg.norm(n)

    def plane_point(x, y, depth):
        v = iCam @ np.array([x, y, 1.0])
        v /= np.linalg.norm(v)
        planep = C0 + n * depth
        return C0 + v * ((planep @ n - n @ C0) / (n @ v))

    def jet(values):
        table = np.stack([np.clip(1.5 - np.abs(4.0 * (np.arange(64) + 1) / 64.0 - c), 0.0, 1.0) for c in (3.0, 2.0, 1.0)], axis=1).astype(np.float32)
        out_ = []
        for v in np.asarray(values, np.float32):
            if v <= 0:
                out_.append([0, 0, 0])
            elif v >= 1:
                out_.append([255, 255, 255])
            else:
                f = v * np.float32(63.0)
                i = int(np.floor(f))
                b = np.float32(f - np.float32(i))
                a = np.float32(1.0) - b
                out_.append([int(np.float32(np.float32(table[i, k] * a + table[i + 1, k] * b) * np.float32(255.0))) for k in range(3)])
        return np.array(out_).reshape(-1, 3)

    for name, vol_t in (("beforeFiltering", h.second), ("afterFiltering", h.best)):
        tile_vol = vol_t.cpu().numpy()
        Y = X = BUF // ss
        vol = np.full((Y, X, tile_vol.shape[2]), 255, tile_vol.dtype)
        vol[:tile_vol.shape[0], :tile_vol.shape[1]] = tile_vol
        Z = len(depths)
        # the whole volume, every 10th column (volumeIO.cpp:148-194)
        pts, sims = [], []
        for vy in range(0, Y, 10):
            for vx in range(0, X, 10):
                for vz in range(Z):
                    s = float(vol[vy, vx, vz])
                    if s > 80.0:
                        continue
                    pts.append(plane_point(vx * ss, vy * ss, float(depths[vz])))
                    sims.append(s / 80.0)
        Xg, Cg = cloud("volume_" + name)
        assert len(pts) == len(Xg) > 100, (len(pts), len(Xg))
        np.testing.assert_allclose(Xg, np.array(pts), rtol=3e-7, atol=1e-6)  # float32 in the archive
        assert np.array_equal(Cg, jet(np.float32(sims)))
        # the cross (volumeIO.cpp:196-246): the centre row in full, the centre column elsewhere
        pts, sims = [], []
        for vz in range(Z):
            for vy in range(Y):
                centre = (vy >= Y // 2) and ((vy - 1) < Y // 2)
                for vx in (range(X) if centre else range(X // 2, X // 2 + 1)):
                    s = float(vol[vy, vx, vz])
                    if s > 80.0:
                        continue
                    pts.append(plane_point(vx * ss, vy * ss, float(depths[vz])))
                    sims.append(s / 80.0)
        Xg, Cg = cloud("volumeCross_" + name)
        assert len(pts) == len(Xg) > 100
        np.testing.assert_allclose(Xg, np.array(pts), rtol=3e-7, atol=1e-6)
        assert np.array_equal(Cg, jet(np.float32(sims)))
        # the topographic cut (volumeIO.cpp:306-376): the row below the centre, lifted by the normalised similarity
        vy = (Y + 1) // 2
        row = vol[vy, :, :Z].astype(np.float32)
        valid = row <= 254.0
        lo, hi = row[valid].min(), row[valid].max()
        norm = np.float32(0.0) if hi == lo else np.float32(1.0) / (hi - lo)
        pts, sims = [], []
        for vx in range(X):
            for vz in range(Z):
                if not valid[vx, vz]:
                    continue
                sn = np.float32((row[vx, vz] - lo) * norm)
                pts.append(plane_point(vx * ss, vy * ss + float(sn) * 15.0, float(depths[vz])))
                sims.append(sn)
        Xg, Cg = cloud("volumeTopographicCut_" + name)
        assert len(sims) == len(Xg) > 100
        np.testing.assert_allclose(Xg, np.array(pts), rtol=3e-7, atol=1e-6)
        assert np.array_equal(Cg, jet(np.float32(sims)))
    # Refine: the cross and the cut exist, hold one point per (pixel of the cross, plane) with a valid middle depth
    Xg, Cg = cloud("volumeCross_afterRefine")
    Zr = 2 * ref.halfNbDepths + 1
    assert len(Xg) > 0 and len(Xg) <= (W + H - 1) * Zr
    Xg, Cg = cloud("volumeTopographicCut_afterRefine")
    assert len(Xg) > 0 and len(Xg) % Zr == 0 and len(Xg) <= W * Zr


def _oracle_jpeg_to_linear(path):
    """the oracle's decode of a JPEG file to linear float RGBA: host entropy decoder -> avo_image_decode_jpeg -> avo_image_decode_integer"""
    import ctypes as C
    from alicevision_amd import jpeg_io
    from oracle import oracle
    olib = oracle.load()
    j = jpeg_io.read_coefficients(path)
    comps = j.descriptors([c["coef"].ctypes.data for c in j.components])
    rgb = np.zeros((j.height, j.width, 3), np.uint8)
    olib.avo_image_decode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(abi.JpegComponent), C.c_int, C.c_int, C.c_int, C.c_int]
    assert olib.avo_image_decode_jpeg(rgb.ctypes.data, 3 * j.width, j.width, j.height, comps, len(j.components), j.hmax, j.vmax, 0 if j.stored_as_rgb else 1) == 0
    lin = np.zeros((j.height, j.width, 4), np.float32)
    assert olib.avo_image_decode_integer(oracle.ptr(lin), j.width * 16, oracle.ptr(rgb), j.width * 3, j.width, j.height, 3, 8, 1) == 0
    return lin


def _write_jpegs(sc, folder, n, per_view=lambda i: {}, **kw):
    """the scene's linear images, sRGB-encoded to 8 bits, as <viewId>.jpg"""
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    paths = []
    for i in range(n):
        lin = sc.images[i].numpy()[..., :3].astype(np.float64)
        enc = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055)
        p = os.path.join(folder, "%d.jpg" % scene_io.view_id(i))
        Image.fromarray(np.clip(np.rint(enc * 255.0), 0, 255).astype(np.uint8)).save(p, **dict(kw, **per_view(i)))
        paths.append(p)
    return paths


def test_jpeg_input_is_decoded_on_the_device(dataset, tmp_path):
    """<viewId>.jpg in --imagesFolder: markers and Huffman decoding on the host, the rest of the decode on the device (avdm_image_decode_jpeg,
    then the sRGB decoding of avdm_image_decode_integer).  The program's maps equal, bit for bit, the harness run on the images the ORACLE
    decodes from the same files (4:2:0, progressive for the odd views)."""
    pytest.importorskip("PIL.Image")
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sfm, img, d = dataset
    jpg_dir = str(tmp_path / "jpg")
    paths = _write_jpegs(sc, jpg_dir, NVIEWS, per_view=lambda i: {"progressive": bool(i % 2)}, quality=95, subsampling=2)
    decoded = [_oracle_jpeg_to_linear(p) for p in paths]
    assert np.abs(decoded[0][..., :3] - sc.images[0].numpy()[..., :3]).mean() < 0.02  # JPEG at quality 95
    out_dir = str(tmp_path / "out_jpg")
    args = ["-i", sfm, "--imagesFolder", jpg_dir, "-o", out_dir, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 64,
            "--colorOptimizationNbIterations", 5, "-v", "warning"]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    run_cli(args)
    depth, sim, _, _ = read_maps(out_dir)
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=5)
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(torch.from_numpy(decoded[i]).cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], np.asarray(t0["depths"], np.float32), tc_ranges=[(a, a + n) for a, n in t0["depthsTcLimits"]])
    got = h.run_refine(0, t0["refineTCams"]).cpu().numpy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())
    assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim)
    assert (depth > 0).mean() > 0.5


def test_prepare_dense_scene_reads_jpeg_sources(tmp_path):
    """aliceVision_prepareDenseScene on JPEG photographs (what a real dataset is): decoded on the device, undistorted there, written as
    OpenEXR — bit for bit the oracle's undistortion of the oracle's decode of the same file; 4:2:2 here"""
    pytest.importorskip("PIL.Image")
    import ctypes as C
    from oracle import oracle
    exe = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_prepareDenseScene")
    w, h = 320, 240
    sc = make_scene(3, w, h, seed=11, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    src_dir = os.path.join(d, "photos")
    paths = _write_jpegs(sc, src_dir, 3, quality=90, subsampling=1)
    sd = scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 50, amp=0.6), src_dir)
    for v, p in zip(sd["views"], paths):
        v["path"] = p
    k = (0.07, -0.03, 0.01)
    sd["intrinsics"][0]["distortionType"] = "radialk3"
    sd["intrinsics"][0]["distortionParams"] = ["%.17g" % v for v in k]
    sfm = os.path.join(d, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(sd, f)
    out = os.path.join(d, "prepared")
    r = subprocess.run([exe, "-i", sfm, "-o", out, "-v", "info"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    olib = oracle.load()
    fx = float(sc.K[0, 0])
    cam = abi.Intrinsic(width=w, height=h, scale_x=fx, scale_y=fx, offset_x=float(sc.K[0, 2]) - w / 2.0, offset_y=float(sc.K[1, 2]) - h / 2.0,
                        distortion_model=abi.DISTORTION_RADIALK3, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0, 0, 0, 0)
    for i in range(3):
        ch, info = exr_io.read_exr(os.path.join(out, "%d.exr" % scene_io.view_id(i)))
        got = np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1)
        src = _oracle_jpeg_to_linear(paths[i])
        want = np.zeros_like(src)
        assert olib.avo_image_undistort(oracle.ptr(want), w * 16, oracle.ptr(src), w * 16, C.byref(cam), C.byref(fill)) == 0
        assert np.array_equal(got, want), (i, float(np.abs(got - want).max()))


def test_tiff_input_is_decoded_on_the_device(dataset, tmp_path):
    """<viewId>.tif in --imagesFolder (16-bit RGB, tiled and big-endian for the odd views): strips / tiles on the host, the integer samples
    to linear float RGBA on the device (avdm_image_decode_integer) — the program's maps equal, bit for bit, the harness run on the images
    the ORACLE decodes from the same samples"""
    import torch
    from test_host_cpu import _write_tiff
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc, sfm, img, d = dataset
    tif_dir = str(tmp_path / "tif")
    os.makedirs(tif_dir)
    olib = oracle.load()
    decoded = []
    for i in range(NVIEWS):
        lin = sc.images[i].numpy()[..., :3].astype(np.float64)
        enc = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055)
        q = np.clip(np.rint(enc * 65535.0), 0, 65535).astype(np.uint16)
        _write_tiff(os.path.join(tif_dir, "%d.tif" % scene_io.view_id(i)), q, big_endian=bool(i % 2), tile=(128, 96) if i % 2 else None)
        out = np.zeros((H, W, 4), np.float32)
        assert olib.avo_image_decode_integer(oracle.ptr(out), W * 16, oracle.ptr(q), W * 6, W, H, 3, 16, 1) == 0
        decoded.append(out)
    out_dir = str(tmp_path / "out_tif")
    args = ["-i", sfm, "--imagesFolder", tif_dir, "-o", out_dir, "--downscale", 1, "--rangeStart", 0, "--rangeSize", 1, "--sgmMaxDepths", 64,
            "--colorOptimizationNbIterations", 5, "-v", "warning"]
    plan = json.loads(run_cli(args + ["--dryRun", 1]).stdout.strip().splitlines()[-1])
    t0 = plan["tiles"][0]
    run_cli(args)
    depth, sim, _, _ = read_maps(out_dir)
    sgm = abi.SgmParams.default(scale=plan["sgmScale"], stepXY=plan["sgmStepXY"])
    ref = abi.RefineParams.default(optimizationNbIterations=5)
    torch.cuda.set_device(0)
    pyr = [DevicePyramid(torch.from_numpy(decoded[i]).cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(NVIEWS)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, t0["sgmTCams"], np.asarray(t0["depths"], np.float32), tc_ranges=[(a, a + n) for a, n in t0["depthsTcLimits"]])
    got = h.run_refine(0, t0["refineTCams"]).cpu().numpy()
    assert np.array_equal(got[..., 0], depth), float(np.abs(got[..., 0] - depth).max())
    assert np.array_equal(got[..., 1].astype(np.float16).astype(np.float32), sim)


def test_prepare_dense_scene_applies_masks(tmp_path):
    """--masksFolders (main_prepareDenseScene.cpp:255-273, image::tryLoadMask): <viewId>.png or <image name>.png, one 8-bit channel; alpha = 0
    where the mask is 0, 1 elsewhere, applied BEFORE the undistortion (so the alpha is resampled with the colours); a mask of another size
    is ignored with a warning, a multi-channel mask is an error like in the reference"""
    import ctypes as C
    from png_util import write_png
    from oracle import oracle
    exe = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_prepareDenseScene")
    w, h = 160, 120
    sc = make_scene(3, w, h, seed=12, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    sfm, img = scene_io.write_scene(sc, d, n_landmarks=10, compression=0)
    sd = scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 50, amp=0.6), img)
    k = (0.08, -0.03, 0.01)
    sd["intrinsics"][0]["distortionType"] = "radialk3"
    sd["intrinsics"][0]["distortionParams"] = ["%.17g" % v for v in k]
    with open(sfm, "w") as f:
        json.dump(sd, f)
    masks = os.path.join(d, "masks")
    os.makedirs(masks)
    yy, xx = np.mgrid[0:h, 0:w]
    m0 = (((xx - 80) ** 2 + (yy - 60) ** 2) < 50 ** 2).astype(np.uint8) * 255   # view 0: a disc, by view id
    m1 = ((xx // 16 + yy // 16) % 2).astype(np.uint8) * 200                       # view 1: a checker board, by image name
    write_png(os.path.join(masks, "%d.png" % scene_io.view_id(0)), m0)
    write_png(os.path.join(masks, "%d.png" % scene_io.view_id(1)), m1)
    write_png(os.path.join(masks, "%d.png" % scene_io.view_id(2)), np.zeros((8, 8), np.uint8))  # wrong size: ignored
    out = os.path.join(d, "prepared")
    r = subprocess.run([exe, "-i", sfm, "-o", out, "--masksFolders", masks, "-v", "info"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mask is ignored" in r.stdout + r.stderr
    olib = oracle.load()
    fx = float(sc.K[0, 0])
    cam = abi.Intrinsic(width=w, height=h, scale_x=fx, scale_y=fx, offset_x=float(sc.K[0, 2]) - w / 2.0, offset_y=float(sc.K[1, 2]) - h / 2.0,
                        distortion_model=abi.DISTORTION_RADIALK3, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0, 0, 0, 0)
    for i, m in enumerate((m0, m1, None)):
        ch, info = exr_io.read_exr(os.path.join(out, "%d.exr" % scene_io.view_id(i)))
        got = np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1)
        src = np.ascontiguousarray(sc.images[i].numpy()).copy()
        if m is not None:
            src[..., 3] = np.where(m == 0, 0.0, 1.0)
        want = np.zeros_like(src)
        assert olib.avo_image_undistort(oracle.ptr(want), w * 16, oracle.ptr(src), w * 16, C.byref(cam), C.byref(fill)) == 0
        assert np.array_equal(got, want), (i, float(np.abs(got - want).max()))
        if m is not None:
            assert 0.05 < float((got[..., 3] == 0).mean()) < 0.95
    write_png(os.path.join(masks, "%d.png" % scene_io.view_id(2)), np.zeros((h, w, 3), np.uint8))  # three channels
    r = subprocess.run([exe, "-i", sfm, "-o", out, "--masksFolders", masks], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Can't load channels" in r.stdout + r.stderr


def test_prepare_dense_scene_exposure_metadata_and_correction(tmp_path):
    """AliceVision:EV / AliceVision:EVComp (main_prepareDenseScene.cpp:241-247) from the views' EXIF metadata, and --evCorrection: the
    colours scaled by the compensation towards the scene's median exposure (alpha untouched), before the undistortion"""
    exe = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_prepareDenseScene")
    w, h = 160, 120
    sc = make_scene(3, w, h, seed=13, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    sfm, img = scene_io.write_scene(sc, d, n_landmarks=10, compression=0)
    sd = scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 50, amp=0.6), img)
    metas = [{"ExposureTime": "1/100", "FNumber": "2", "ISO": "100"}, {"ExposureTime": "1/200", "FNumber": "2", "ISO": "100"},
             {"ExposureTime": "1/400", "FNumber": "2", "ISO": "100"}]
    for v, m in zip(sd["views"], metas):
        v["metadata"] = m
    with open(sfm, "w") as f:
        json.dump(sd, f)
    exposures = [0.01 / 4, 0.005 / 4, 0.0025 / 4]  # shutter * (1 / fnumber)^2 at ISO 100
    median = exposures[1]
    for correct in (0, 1):
        out = os.path.join(d, "prepared%d" % correct)
        r = subprocess.run([exe, "-i", sfm, "-o", out, "--evCorrection", str(correct)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        for i in range(3):
            ch, info = exr_io.read_exr(os.path.join(out, "%d.exr" % scene_io.view_id(i)))
            comp = np.float32(median / exposures[i])
            assert exr_io.attr_value(info, "AliceVision:EV") == np.float32(np.log2(1.0 / exposures[i]))
            assert exr_io.attr_value(info, "AliceVision:EVComp") == comp
            src = sc.images[i].numpy()
            got = np.stack([ch["R"], ch["G"], ch["B"], ch["A"]], -1)
            want = src.copy()
            if correct:
                want[..., :3] = src[..., :3] * comp
            assert np.array_equal(got, want), (correct, i)
    # the exposure values do not depend on --saveMetadata (main_prepareDenseScene.cpp:241-246 push them before the switch is looked at)
    out = os.path.join(d, "prepared_nometa")
    r = subprocess.run([exe, "-i", sfm, "-o", out, "--saveMetadata", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ch, info = exr_io.read_exr(os.path.join(out, "%d.exr" % scene_io.view_id(0)))
    assert exr_io.attr_value(info, "AliceVision:EVComp") == np.float32(median / exposures[0])
    assert "AliceVision:EV" in info["attributes"] and "AliceVision:P" not in info["attributes"]
