"""The depth-plane list pinned to the REFERENCE'S OWN code: oracle/host_oracle.depth_list — which the C++ host equals
(tests/test_host_cpu.py) — against depthMap/SgmDepthList.cpp compiled whole and unchanged for this CPU (oracle/_ref/libavdm_host_ref.so,
oracle/ref/host_driver.cpp): same cameras, landmarks, tile and parameters into both.

The two sides derive K, R, C from the projection matrix with different RQ codes (numpy there, the reference's Matrix3x3::RQ here) and
the oracle evaluates the geometry in numpy double expressions: the planes agree to fp32 rounding (the bar of test_host_cpu.py, 2e-6
relative), their NUMBER and the per-T-camera limits agree exactly."""
import os

import numpy as np
import pytest

from alicevision_amd import scene_io
from alicevision_amd.synthetic import make_scene
from oracle import host_oracle as ho
from oracle import host_ref as hr

pytestmark = pytest.mark.skipif(not hr.available(), reason="oracle/_ref/libavdm_host_ref.so not built (no reference tree)")


@pytest.fixture(scope="module")
def scene():
    sc = make_scene(6, 640, 480, seed=9, baseline=0.9, amp=0.6, render=[])
    return sc, scene_io.sample_landmarks(sc, 600, amp=0.6)


def _both(sc, lms, rc, tcams, roi, ds=1, **kw):
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height, process_downscale=ds)
    a = ho.depth_list(cams, lms, rc, tcams, roi, **kw)
    b = hr.depth_list(sc.K, sc.R, sc.C, sc.width, sc.height, lms, rc, tcams, roi, process_downscale=ds, **kw)
    return a, b


def _same(a, b):
    (da, la), (db, lb) = a, b
    assert len(da) == len(db), (len(da), len(db))
    if len(da):
        assert np.allclose(np.asarray(da, np.float64), np.asarray(db, np.float64), rtol=2e-6, atol=0.0), np.abs(np.asarray(da) - np.asarray(db)).max()
        assert [tuple(l) for l in la] == [tuple(l) for l in lb], (la, lb)


@pytest.mark.parametrize("max_depths", [1500, 96, 12])
def test_depth_list_full_image_equals_reference(scene, max_depths):
    """whole-image tile of every camera against its landmark-ranked neighbours; uncapped, capped (second pass with a scale) and heavily capped"""
    sc, lms = scene
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height)
    n_planes = []
    for rc in range(6):
        tc = ho.nearest_cams_from_landmarks(cams, lms, rc, 10)
        a, b = _both(sc, lms, rc, tc, (0, 640, 0, 480), sgm_scale=2, max_depths=max_depths)
        _same(a, b)
        n_planes.append(len(a[0]))
    assert min(n_planes) > 8
    if max_depths < 1500:
        assert max(n_planes) <= max_depths


def test_depth_list_per_tile_and_downscale_equals_reference(scene):
    """tiles of the image with depthListPerTile (landmark selection by ROI, reference pixel = ROI centre), at process downscale 1 and 2,
    stepZ given, SfM seeds off"""
    sc, lms = scene
    for ds in (1, 2):
        cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height, process_downscale=ds)
        W, H = 640 // ds, 480 // ds
        rois = ho.tile_roi_list(416 // ds, 352 // ds, 32 // ds, W, H, 4)
        assert len(rois) == 4
        for rc in (0, 3):
            tc = ho.nearest_cams_from_landmarks(cams, lms, rc, 4)
            for roi in rois:
                tt = ho.tile_nearest_cams(cams, lms, rc, 3, tc, roi)
                if not tt:
                    continue
                for kw in (dict(depth_list_per_tile=True, max_depths=64), dict(depth_list_per_tile=True, max_depths=1500, step_z=2),
                           dict(depth_list_per_tile=False, max_depths=200, use_sfm_seeds=False), dict(seeds_range_inflate=0.5, max_depths=1500)):
                    a, b = _both(sc, lms, rc, tt, roi, ds=ds, sgm_scale=2, **kw)
                    _same(a, b)


def test_no_landmarks_no_list(scene):
    """a tile whose ROI holds no landmark of the R camera gets no depth list from either (SgmDepthList.cpp:60-66)"""
    sc, lms = scene
    few = [(X, obs) for X, obs in lms if 0 in obs and obs[0][0] > 400]
    a, b = _both(sc, few, 0, [1, 2], (0, 200, 0, 200), sgm_scale=2, max_depths=64, depth_list_per_tile=True)
    assert len(a[0]) == 0 and len(b[0]) == 0


@pytest.mark.parametrize("case", [(4000, 3000, 1024, 1024, 64, 4), (6000, 4000, 1664, 1152, 64, 4), (1920, 1080, 1024, 1024, 64, 4),
                                  (640, 480, 1024, 1024, 64, 2), (1001, 777, 300, 260, 32, 4), (5000, 900, 1024, 1024, 128, 8),
                                  (640, 480, 416, 352, 32, 4), (320, 240, 208, 176, 16, 4)])
def test_tile_roi_list_equals_reference(case):
    """mvsUtils::getTileRoiList (TileParams.cpp:15-61, compiled whole): the tile grid of BASELINE's configurations and of odd sizes"""
    W, H, bw, bh, pad, md = case
    assert ho.tile_roi_list(bw, bh, pad, W, H, md) == hr.tile_roi_list(bw, bh, pad, W, H, md)


def test_bench_cfg5_tiles_are_the_reference_grid():
    """bench.py's multi-tile workload (cfg5: 24 MP, tile buffer 1664 x 1152, padding 64) sweeps the tiles the reference would make"""
    import ast
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "tile_rois"][0]
    consts = [n for n in ast.parse(src).body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "TILE_BUFFER"]
    ns = {}
    exec(compile(ast.Module(consts + [fn], []), "bench.py", "exec"), ns)
    assert ns["TILE_BUFFER"] == (1664, 1152)
    got = ns["tile_rois"](6000, 4000, 4)
    assert got == hr.tile_roi_list(1664, 1152, 64, 6000, 4000, 4) and len(got) == 16
    assert ns["tile_rois"](4000, 3000, 1) == [None]


@pytest.mark.parametrize("case", [(640, 480, 416, 352, 32, 4, 1), (640, 480, 416, 352, 32, 4, 2), (1001, 777, 300, 260, 32, 4, 1),
                                  (1001, 777, 300, 260, 32, 4, 2), (4000, 3000, 1024, 1024, 64, 4, 4)])
def test_tile_weights_equal_reference(case):
    """weightTileBorder / addSingleTileMapWeighted (mapIO.cpp:170-311, from the reference's text): the weight map of every tile of the
    grid, bit for bit, and the weights of all tiles summing to one wherever tiles overlap as the reference's merge assumes"""
    W, H, bw, bh, pad, md, ds = case
    total = None
    for roi in ho.tile_roi_list(bw, bh, pad, W, H, md):
        a, (bx, ex, by, ey) = ho.tile_weight_map(roi, W, H, pad, ds)
        b, full = hr.tile_weight_map(roi, W, H, pad, ds)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (roi, float(np.abs(a - b).max()))
        total = full if total is None else total + full
        assert np.array_equal(full[by:ey, bx:ex], b)
    assert total.min() > 0.0


def test_nearest_cams_equal_reference(scene):
    """MultiViewParams::findNearestCamsFromLandmarks (MultiViewParams.cpp:519-575) and findTileNearestCams (:577-667), from the
    reference's text: the landmark ranking of every camera for several list lengths, and the per-tile selection for every tile of a
    2 x 2 grid, at process downscale 1 and 2 (camera::angleBetweenRays is a stand-in: the one unpinned function on this path)"""
    sc, lms = scene
    for ds in (1, 2):
        cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height, process_downscale=ds)
        rois = ho.tile_roi_list(416 // ds, 352 // ds, 32 // ds, 640 // ds, 480 // ds, 4)
        for rc in range(6):
            for nb in (10, 4, 2):
                a = ho.nearest_cams_from_landmarks(cams, lms, rc, nb)
                b = hr.nearest_cams(sc.K, sc.R, lms, rc, nb, process_downscale=ds)
                assert a == b, (ds, rc, nb, a, b)
            tc = ho.nearest_cams_from_landmarks(cams, lms, rc, 10)
            for roi in rois + [(0, 640 // ds, 0, 480 // ds)]:
                for nb in (3, 2, 10):
                    a = ho.tile_nearest_cams(cams, lms, rc, nb, tc, roi)
                    b = hr.nearest_cams(sc.K, sc.R, lms, rc, nb, tcams=tc, roi=roi, process_downscale=ds)
                    assert a == b, (ds, rc, roi, nb, a, b)


def test_nearest_cams_ties_equal_reference():
    """equal scores: the reference sorts with qsort and a comparator that never returns 0 (mvsData/structures.cpp:37-47); what that does
    with ties is part of the behaviour"""
    sc = make_scene(7, 320, 240, seed=4, baseline=0.9, amp=0.3, render=[])
    lms = scene_io.sample_landmarks(sc, 250, amp=0.3)
    # every landmark seen by all views: equal counts for every T camera that passes the angle test
    full = [(X, obs) for X, obs in lms if len(obs) == 7]
    assert len(full) > 40
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height)
    for rc in range(7):
        for nb in (10, 3):
            a = ho.nearest_cams_from_landmarks(cams, full, rc, nb)
            b = hr.nearest_cams(sc.K, sc.R, full, rc, nb)
            assert a == b, (rc, nb, a, b)


def test_parameter_defaults_equal_reference_headers():
    """every default of SgmParams / RefineParams / DepthMapParams / TileParams as the C++ host holds it (avdm_host_tool params) against
    the reference's own headers compiled into the pin library, and the kernel-side defaults of the Python harness (abi.py)"""
    import os
    import subprocess
    from alicevision_amd import abi
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alicevision_amd", "bin", "avdm_host_tool")
    assert os.path.exists(tool), "host tool not built"
    out = subprocess.run([tool, "params"], capture_output=True, text=True, timeout=60, check=True).stdout
    host = dict(line.split("=", 1) for line in out.strip().splitlines())
    ref = hr.default_params()
    assert len(ref) > 50 and set(ref) == set(host)
    assert ref == host, {k: (ref[k], host[k]) for k in ref if ref[k] != host[k]}
    s, r = abi.SgmParams.default(), abi.RefineParams.default()
    for k in ("scale", "stepXY", "wsh", "gammaC", "gammaP", "p1", "p2Weighting", "maxSimilarity", "depthThicknessInflate", "useConsistentScale",
              "useCustomPatchPattern"):
        assert float(getattr(s, k)) == float(ref["sgm." + k]), k
    assert s.filteringAxes.decode() == ref["sgm.filteringAxes"]
    for k in ("scale", "stepXY", "wsh", "halfNbDepths", "nbSubsamples", "optimizationNbIterations", "sigma", "gammaC", "gammaP", "interpolateMiddleDepth",
              "useConsistentScale", "useCustomPatchPattern"):
        assert float(getattr(r, k)) == float(ref["refine." + k]), k


@pytest.mark.parametrize("model,k,size", [(1, (0.08, 0.0, 0.0), (101, 75)), (2, (0.1, -0.05, 0.01), (101, 75)), (2, (-0.25, 0.08, 0.0), (160, 120)),
                                          (3, (0.05, 0.02, -0.01), (101, 75)), (0, (0.0, 0.0, 0.0), (40, 30))])
def test_undistort_equals_reference(model, k, size):
    """the image-ingest slice: avo_image_undistort — which the HIP kernel equals bit for bit
    (tests/test_gpu_parity.py::test_image_undistort_bit_exact) — against the reference's own camera::UndistortImage with its distortion
    classes, pixel <-> camera transforms and bilinear sampler (out-of-range neighbours dropped and renormalised, nearest pixel below 0.2
    of the weight): identical, pixel for pixel, barrel and pincushion, fill colour where the distorted position leaves the image"""
    import ctypes as C
    from alicevision_amd import abi
    from oracle import oracle
    lib = oracle.load()
    W, H = size
    rng = np.random.default_rng(model * 7 + W)
    src = rng.random((H, W, 4), dtype=np.float32)
    cam = abi.Intrinsic(width=W, height=H, scale_x=0.9 * W, scale_y=0.88 * W, offset_x=2.25, offset_y=-1.5, distortion_model=model, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0.25, 0.5, 0.75, 0.0)
    got = np.zeros_like(src)
    assert lib.avo_image_undistort(oracle.ptr(got), W * 16, oracle.ptr(src), W * 16, C.byref(cam), C.byref(fill)) == 0
    want = hr.image_undistort(src, cam, list(fill))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (float(np.abs(got - want).max()), float((got != want).mean()))
    if model:
        assert not np.array_equal(got, src)


def test_exposure_setting_equals_reference():
    """ExposureSetting::getExposure / isPartiallyDefined (sfmData/ExposureSetting.hpp, compiled as it lies into oracle/_ref) against the
    host's restatement (sfmData.cpp) over a grid of shutters, apertures and sensitivities, the undefined ones (-1, 0, nan) included: the
    quantity behind AliceVision:EV / AliceVision:EVComp and --evCorrection of aliceVision_prepareDenseScene"""
    import ctypes as C
    import subprocess
    lib = C.CDLL(hr.LIB_PATH)
    lib.avr_exposure.restype = C.c_double
    lib.avr_exposure.argtypes = [C.c_double] * 3
    lib.avr_exposure_partially_defined.argtypes = [C.c_double] * 3
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alicevision_amd", "bin", "avdm_host_tool")
    __import__("common").build_host()
    triples = [(s, f, i) for s in (-1.0, 0.0, 1.0 / 4000, 1.0 / 200, 0.005, 1.0 / 3, 2.5, float("nan"))
               for f in (-1.0, 0.0, 1.4, 2.8, 5.6, 22.0, float("inf")) for i in (-1.0, 0.0, 1e-7, 50.0, 100.0, 800.0, 25600.0)]
    args = [repr(v) for t in triples for v in t]
    out = subprocess.run([tool, "exposure-of"] + args, capture_output=True, text=True, check=True).stdout.split()
    got = np.array(out, dtype=np.float64).reshape(-1, 2)
    for (s, f, i), (e, pd) in zip(triples, got):
        want = lib.avr_exposure(s, f, i)
        assert (np.isnan(want) and np.isnan(e)) or want == e, (s, f, i, want, e)
        assert int(pd) == lib.avr_exposure_partially_defined(s, f, i)
