"""aliceVision_depthMapFiltering end to end on the GPU: the files it writes (modal-count PNG, filtered depth / similarity EXR, normal
maps) against the CPU restatement of fuseCut::Fuser fed with the SAME cameras (dumped with round-trip precision by
`avdm_host_tool fuse-cameras`) — bit-exact — and the chain depth-map estimation -> filtering."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from alicevision_amd import abi, exr_io, scene_io
from alicevision_amd.synthetic import make_scene
from fuse_scene import make_fuse_scene, write_depth_maps

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "alicevision_amd", "bin")
FILTER_CLI = os.path.join(BIN, "aliceVision_depthMapFiltering")
ESTIMATION_CLI = os.path.join(BIN, "aliceVision_depthMapEstimation")
TOOL = os.path.join(BIN, "avdm_host_tool")


def run(cmd, check=True):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=300)
    if check:
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def _half(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def test_filtering_cli_matches_the_restatement(oracle_lib, tmp_path):
    from oracle import fuse_oracle as fo
    from oracle import oracle
    from png_util import read_png_gray8
    n, w, h, nn = 5, 320, 240, 3
    fs = make_fuse_scene(n, w, h, seed=13, noise=2e-4, outliers=0.08, weak=0.25, masked=0.03)
    lms = scene_io.sample_landmarks(fs, 500)
    d = str(tmp_path)
    sfm = os.path.join(d, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(fs, lms, os.path.join(d, "images")), f)
    dm, flt = os.path.join(d, "depthMaps"), os.path.join(d, "filtered")
    write_depth_maps(dm, fs, fs.depth, fs.sim)
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt, "--nNearestCams", nn, "--computeNormalMaps", 1, "-v", "warning"])

    info = json.loads(run([TOOL, "fuse-cameras", sfm, dm, flt, nn]).stdout)["cams"]
    cams = [fo.fuse_cam(np.array(c["P"]), np.array(c["iP"]), np.array(c["C"]), c["width"], c["height"]) for c in info]
    # what the program read: float depth, half similarity
    depth = [exr_io.read_exr(os.path.join(dm, "%d_depthMap.exr" % scene_io.view_id(i)))[0]["Y"] for i in range(n)]
    sim = [exr_io.read_exr(os.path.join(dm, "%d_simMap.exr" % scene_io.view_id(i)))[0]["Y"].astype(np.float32) for i in range(n)]
    for i in range(n):
        assert np.array_equal(depth[i], fs.depth[i]) and np.array_equal(sim[i], _half(fs.sim[i]))
    kept = 0
    for rc in range(n):
        vid = scene_io.view_id(rc)
        tc = info[rc]["tcams"]
        assert len(tc) == nn and rc not in tc
        want_nmod = fo.filter_groups_rc(depth[rc], sim[rc], cams[rc], [cams[t] for t in tc], [depth[t] for t in tc])
        got_nmod = read_png_gray8(os.path.join(flt, "%d_nmodMap.png" % vid))
        assert want_nmod.max() == nn
        assert np.array_equal(got_nmod, want_nmod), f"rc {rc}: {(got_nmod != want_nmod).sum()} pixels differ"
        want_d, want_s = fo.filter_depth_maps_rc(depth[rc], sim[rc], want_nmod)
        got_d, dinfo = exr_io.read_exr(os.path.join(flt, "%d_depthMap.exr" % vid))
        got_s, _ = exr_io.read_exr(os.path.join(flt, "%d_simMap.exr" % vid))
        assert np.array_equal(got_d["Y"], want_d)
        assert np.array_equal(got_s["Y"].astype(np.float32), _half(want_s))
        kept += int((want_d > 0).sum())
        # metadata of the filtered depth map (mapIO.cpp:440-511)
        assert exr_io.attr_value(dinfo, "AliceVision:downscale") == 1
        assert exr_io.attr_value(dinfo, "AliceVision:nbDepthValues") == int((want_d > 0).sum())
        assert "AliceVision:P" in dinfo["attributes"]
        # normal map of the filtered depth map against the oracle's closed-form eigenvector (tolerance class, see test_gpu_parity)
        nm, _ = exr_io.read_exr(os.path.join(flt, "%d_normalMap.exr" % vid))
        got_n = np.stack([nm["R"], nm["G"], nm["B"]], axis=-1).astype(np.float32)
        cam = abi.camera_fill(fs.K, fs.R[rc], fs.C[rc], 1)
        dsm = np.ascontiguousarray(np.stack([want_d, np.ones_like(want_d)], axis=-1))
        want_n = np.zeros((h, w, 3), np.float32)
        oracle.load().avo_depth_sim_map_compute_normal(oracle.ptr(want_n), w * 12, oracle.ptr(dsm), w * 8, C.byref(cam), 1, abi.ROI.make(0, w, 0, h))
        inv_w, inv_g = np.all(want_n == -1.0, axis=-1), np.all(got_n == -1.0, axis=-1)
        assert (inv_w == inv_g).mean() > 0.999
        ok = ~inv_w & ~inv_g
        g = got_n[ok].astype(np.float64)
        assert np.allclose(np.linalg.norm(g, axis=-1), 1.0, atol=2e-3)  # stored as half: 1e-3 per component
        cosang = np.einsum("ij,ij->i", want_n[ok].astype(np.float64), g / np.linalg.norm(g, axis=-1, keepdims=True))
        assert ok.mean() > 0.3 and (cosang > 1.0 - 1e-5).mean() > 0.97
    assert kept > 0.5 * n * w * h

    # second run: the modal-count maps exist and are reused (Fuser.cpp:146-149), outputs unchanged
    before = open(os.path.join(flt, "%d_depthMap.exr" % scene_io.view_id(0)), "rb").read()
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt, "--nNearestCams", nn, "-v", "warning"])
    assert open(os.path.join(flt, "%d_depthMap.exr" % scene_io.view_id(0)), "rb").read() == before
    # the reference's order (all first passes, then all second passes; hidden switch) writes the same files as the single pass
    flt3 = os.path.join(d, "filtered3")
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt3, "--nNearestCams", nn, "--twoPasses", 1, "-v", "warning"])
    for i in range(n):
        for suffix in ("_nmodMap.png", "_depthMap.exr", "_simMap.exr"):
            name = "%d%s" % (scene_io.view_id(i), suffix)
            assert open(os.path.join(flt3, name), "rb").read() == open(os.path.join(flt, name), "rb").read(), name
    # a sub-range only touches its cameras
    flt2 = os.path.join(d, "filtered2")
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt2, "--nNearestCams", nn, "--rangeStart", 1, "--rangeSize", 2, "--pixSizeBall", 1,
         "--pixSizeBallWithLowSimilarity", 1, "--minNumOfConsistentCams", 2, "-v", "warning"])
    assert sorted(f for f in os.listdir(flt2) if f.endswith("nmodMap.png")) == ["%d_nmodMap.png" % scene_io.view_id(i) for i in (1, 2)]
    rc = 1
    tc = info[rc]["tcams"]
    want_nmod = fo.filter_groups_rc(depth[rc], sim[rc], cams[rc], [cams[t] for t in tc], [depth[t] for t in tc], 2.0, 1, 1)
    assert np.array_equal(read_png_gray8(os.path.join(flt2, "%d_nmodMap.png" % scene_io.view_id(rc))), want_nmod)
    want_d, _ = fo.filter_depth_maps_rc(depth[rc], sim[rc], want_nmod, 2, 4)
    assert np.array_equal(exr_io.read_exr(os.path.join(flt2, "%d_depthMap.exr" % scene_io.view_id(rc)))[0]["Y"], want_d)


def test_estimation_then_filtering(tmp_path):
    """the filtered maps of a real run: every view's depth map is estimated, then filtered against the others"""
    n, w, h = 4, 480, 360
    sc = make_scene(n, w, h, seed=5, baseline=0.9, amp=0.6)
    d = str(tmp_path)
    sfm, img = scene_io.write_scene(sc, d, n_landmarks=500, compression=0)
    # landmarks of write_scene use the default surface amplitude: rewrite the file with this scene's
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(sc, scene_io.sample_landmarks(sc, 500, amp=0.6), img), f)
    dm, flt = os.path.join(d, "depthMaps"), os.path.join(d, "filtered")
    run([ESTIMATION_CLI, "-i", sfm, "--imagesFolder", img, "-o", dm, "--downscale", 1, "--sgmMaxDepths", 96, "--colorOptimizationNbIterations", 10,
         "--maxTCams", 3, "-v", "warning"])
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt, "--nNearestCams", 3, "--minNumOfConsistentCams", 2, "-v", "warning"])
    for i in range(n):
        vid = scene_io.view_id(i)
        before = exr_io.read_exr(os.path.join(dm, "%d_depthMap.exr" % vid))[0]["Y"]
        after = exr_io.read_exr(os.path.join(flt, "%d_depthMap.exr" % vid))[0]["Y"]
        assert after.shape == before.shape == (h, w)
        valid_b, valid_a = before > 0, after > 0
        assert valid_b.mean() > 0.5
        assert not (valid_a & ~valid_b).any() and np.array_equal(after[valid_a], before[valid_a])  # filtering only removes
        inner = np.zeros_like(valid_b)
        inner[60:-60, 60:-60] = True
        assert (valid_a & inner).sum() > 0.6 * (valid_b & inner).sum(), (i, (valid_a & inner).sum(), (valid_b & inner).sum())


def test_filtering_with_a_missing_neighbour_depth_map(oracle_lib, tmp_path):
    """a neighbour camera without depth map files: readMap gives an all-zero map (mapIO.cpp:343-357), the camera contributes no point
    but still takes its place in the ranking (and in the carried-over counters)"""
    from oracle import fuse_oracle as fo
    from png_util import read_png_gray8
    n, w, h, nn = 4, 200, 150, 3
    fs = make_fuse_scene(n, w, h, seed=17, noise=2e-4, outliers=0.05)
    d = str(tmp_path)
    sfm = os.path.join(d, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(fs, scene_io.sample_landmarks(fs, 400), os.path.join(d, "images")), f)
    dm, flt = os.path.join(d, "depthMaps"), os.path.join(d, "filtered")
    write_depth_maps(dm, fs, fs.depth, fs.sim)
    info = json.loads(run([TOOL, "fuse-cameras", sfm, dm, flt, nn]).stdout)["cams"]  # cameras while every map is still there
    missing = info[0]["tcams"][1]
    for suffix in ("_depthMap.exr", "_simMap.exr"):
        os.remove(os.path.join(dm, "%d%s" % (scene_io.view_id(missing), suffix)))
    run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", dm, "-o", flt, "--nNearestCams", nn, "--rangeStart", 0, "--rangeSize", 1, "-v", "error"])
    cams = [fo.fuse_cam(np.array(c["P"]), np.array(c["iP"]), np.array(c["C"]), c["width"], c["height"]) for c in info]
    tc = info[0]["tcams"]
    sim0 = fs.sim[0].astype(np.float16).astype(np.float32)
    depths = [np.zeros_like(fs.depth[t]) if t == missing else fs.depth[t] for t in tc]
    want = fo.filter_groups_rc(fs.depth[0], sim0, cams[0], [cams[t] for t in tc], depths)
    got = read_png_gray8(os.path.join(flt, "%d_nmodMap.png" % scene_io.view_id(0)))
    assert want.max() == nn and np.array_equal(got, want)
