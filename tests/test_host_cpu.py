"""CPU tests of the C++ host (alicevision_amd/host): no GPU needed.

The C++ code is exercised through the CLI's --dryRun (tiles, T cameras and depth-plane lists as JSON) and through
bin/avdm_host_tool, and compared with the independent Python restatement oracle/host_oracle.py and with the numpy EXR codec
alicevision_amd/exr_io.py.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import make_scene
from oracle import host_oracle as ho

from fuse_scene import write_depth_maps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
TOOL = os.path.join(ROOT, "alicevision_amd", "bin", "avdm_host_tool")


@pytest.fixture(scope="session", autouse=True)
def built():
    __import__("common").build_host()
    assert os.path.exists(CLI) and os.path.exists(TOOL)


def run(cmd, check=True):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("host_scene"))
    sc = make_scene(6, 640, 480, seed=9, baseline=0.9, amp=0.6)
    lms = scene_io.sample_landmarks(sc, 600, amp=0.6)
    os.makedirs(os.path.join(d, "images"))
    sfm = os.path.join(d, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), f)
    for i in range(6):
        im = sc.images[i].numpy()
        exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]},
                         compression=0)
    return sc, lms, sfm, os.path.join(d, "images"), d


def plan_of(sfm, img, out, extra=()):
    r = run([CLI, "-i", sfm, "--imagesFolder", img, "-o", out, "--downscale", 1, "--dryRun", 1, "-v", "error"] + list(extra))
    return json.loads(r.stdout.strip().splitlines()[-1])


# ------------------------------------------------------------------------------------------------------------- tiling
@pytest.mark.parametrize("case", [(4000, 3000, 1024, 1024, 64, 4), (6000, 4000, 1664, 1152, 64, 4), (1920, 1080, 1024, 1024, 64, 4),
                                  (640, 480, 1024, 1024, 64, 2), (1001, 777, 300, 260, 32, 4), (5000, 900, 1024, 1024, 128, 8)])
def test_tile_roi_list(case):
    W, H, bw, bh, pad, md = case
    got = [tuple(int(v) for v in l.split()) for l in run([TOOL, "tiles", W, H, bw, bh, pad, md]).stdout.strip().splitlines()]
    want = ho.tile_roi_list(bw, bh, pad, W, H, md)
    assert got == want
    # the tiles cover the image and each fits its buffer
    cover = np.zeros((H, W), bool)
    for x0, x1, y0, y1 in got:
        cover[y0:y1, x0:x1] = True
        if len(got) > 1:
            assert x1 - x0 <= bw and y1 - y0 <= bh
    assert cover.all()
    if case == (6000, 4000, 1664, 1152, 64, 4):
        assert len(got) == 16  # BASELINE cfg5: 4 x 4 tiles


# ------------------------------------------------------------------------------------- T cameras and depth-plane lists
def test_plan_matches_restatement_single_tile(scene):
    sc, lms, sfm, img, d = scene
    plan = plan_of(sfm, img, os.path.join(d, "o1"), ["--sgmMaxDepths", 96])
    assert plan["sgmStepXY"] == 1 and plan["sgmMaxTCamsPerTile"] == 10  # single-tile auto adjustment (main_depthMapEstimation.cpp:360-390)
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height)
    assert len(plan["tiles"]) == 6
    for t in plan["tiles"]:
        rc = t["rc"]
        assert t["viewId"] == scene_io.view_id(rc) and t["roi"] == [0, 640, 0, 480]
        tc = ho.nearest_cams_from_landmarks(cams, lms, rc, 10)
        sgm_t = ho.tile_nearest_cams(cams, lms, rc, 10, tc, tuple(t["roi"]))
        assert t["sgmTCams"] == sgm_t and t["refineTCams"] == sgm_t
        depths, limits = ho.depth_list(cams, lms, rc, sgm_t, tuple(t["roi"]), sgm_scale=plan["sgmScale"], max_depths=96)
        assert len(depths) == len(t["depths"]) > 8
        assert np.allclose(np.asarray(depths, np.float64), np.asarray(t["depths"]), rtol=2e-6), np.abs(np.asarray(depths) - np.asarray(t["depths"])).max()
        assert [list(l) for l in limits] == t["depthsTcLimits"]
        assert np.all(np.diff(t["depths"]) > 0)
        zmin, zmax = sc.z_range
        assert t["depths"][0] < zmin + 0.2 and t["depths"][-1] > zmax - 0.4


def test_plan_matches_restatement_tiled_and_capped(scene):
    sc, lms, sfm, img, d = scene
    extra = ["--autoAdjustSmallImage", 0, "--tileBufferWidth", 416, "--tileBufferHeight", 352, "--tilePadding", 32, "--sgmMaxDepths", 12,
             "--sgmDepthListPerTile", 1, "--maxTCams", 4, "--sgmMaxTCamsPerTile", 3, "--refineMaxTCamsPerTile", 2, "--rangeStart", 1, "--rangeSize", 2]
    plan = plan_of(sfm, img, os.path.join(d, "o2"), extra)
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height)
    rois = ho.tile_roi_list(416, 352, 32, 640, 480, 4)
    assert len(plan["tiles"]) == 2 * len(rois) == 8
    assert [t["rc"] for t in plan["tiles"]] == [1] * 4 + [2] * 4
    for t in plan["tiles"]:
        rc, roi = t["rc"], tuple(t["roi"])
        assert roi == rois[t["id"]]
        tc = ho.nearest_cams_from_landmarks(cams, lms, rc, 4)
        assert t["sgmTCams"] == ho.tile_nearest_cams(cams, lms, rc, 3, tc, roi)
        assert t["refineTCams"] == ho.tile_nearest_cams(cams, lms, rc, 2, tc, roi)
        depths, limits = ho.depth_list(cams, lms, rc, t["sgmTCams"], roi, sgm_scale=2, max_depths=12, depth_list_per_tile=True)
        assert len(t["depths"]) == len(depths) <= 12
        assert np.allclose(np.asarray(depths, np.float64), np.asarray(t["depths"]), rtol=2e-6)
        assert [list(l) for l in limits] == t["depthsTcLimits"]


def test_plan_from_image_metadata_equals_plan_from_sfm(scene, tmp_path):
    """MultiViewParams prefers the AliceVision:P matrix stored in the image (MultiViewParams.cpp:150-156)"""
    sc, lms, sfm, img, d = scene
    img2 = str(tmp_path / "images_p")
    os.makedirs(img2)
    for i in range(6):
        P = sc.K @ np.concatenate([sc.R[i], (-sc.R[i] @ sc.C[i])[:, None]], axis=1)
        tiny = np.zeros((sc.height, sc.width), np.float32)
        exr_io.write_exr(os.path.join(img2, "%d.exr" % scene_io.view_id(i)), {"R": tiny, "G": tiny, "B": tiny},
                         attributes={"AliceVision:P": exr_io.m44d(list(P.flatten()) + [0, 0, 0, 1]), "AliceVision:downscale": 1}, compression=3)
    a = plan_of(sfm, img, str(tmp_path / "oa"), ["--sgmMaxDepths", 64])
    b = plan_of(sfm, img2, str(tmp_path / "ob"), ["--sgmMaxDepths", 64])
    for ta, tb in zip(a["tiles"], b["tiles"]):
        assert ta["sgmTCams"] == tb["sgmTCams"] and ta["depthsTcLimits"] == tb["depthsTcLimits"]
        assert np.allclose(ta["depths"], tb["depths"], rtol=1e-6)


def test_process_downscale_plan(scene):
    sc, lms, sfm, img, d = scene
    r = run([CLI, "-i", sfm, "--imagesFolder", img, "-o", os.path.join(d, "o3"), "--downscale", 2, "--dryRun", 1, "-v", "error", "--sgmMaxDepths", 64])
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    cams = ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height, process_downscale=2)
    t = plan["tiles"][0]
    assert t["roi"] == [0, 320, 0, 240]
    tc = ho.nearest_cams_from_landmarks(cams, lms, 0, 10)
    sgm_t = ho.tile_nearest_cams(cams, lms, 0, 10, tc, tuple(t["roi"]))
    assert t["sgmTCams"] == sgm_t
    depths, limits = ho.depth_list(cams, lms, 0, sgm_t, tuple(t["roi"]), sgm_scale=plan["sgmScale"], max_depths=64)
    assert np.allclose(np.asarray(depths, np.float64), np.asarray(t["depths"]), rtol=2e-6)


# ---------------------------------------------------------------------------------------------------------------- EXR
@pytest.mark.parametrize("compression", [0, 2, 3])
@pytest.mark.parametrize("half", [False, True])
def test_exr_cpp_reads_python_writes_and_back(tmp_path, compression, half):
    rng = np.random.RandomState(3)
    h, w = 37, 53
    chans = {"R": rng.rand(h, w).astype(np.float32), "G": (rng.rand(h, w) * 1000).astype(np.float32), "B": -rng.rand(h, w).astype(np.float32),
             "A": np.ones((h, w), np.float32)}
    chans["G"][3, 5] = 0.0
    chans["R"][0, 0] = 6.1e-5  # near the half subnormal boundary
    a, b = str(tmp_path / "a.exr"), str(tmp_path / "b.exr")
    attrs = {"AliceVision:downscale": 2, "note": "hello", "AliceVision:P": exr_io.m44d(range(16)), "f": 1.5}
    exr_io.write_exr(a, chans, attributes=attrs, half=half, compression=compression, data_origin=(4, 6), display_size=(100, 90))
    run([TOOL, "exr-copy", a, b, 1 if half else 0])
    got, info = exr_io.read_exr(b)
    src, _ = exr_io.read_exr(a)
    assert info["data_window"] == (4, 6, 4 + w - 1, 6 + h - 1) and info["display_window"] == (0, 0, 99, 89)
    for n in chans:
        assert np.array_equal(got[n], src[n]), n
        if not half:
            assert np.array_equal(got[n], chans[n])
    assert exr_io.attr_value(info, "AliceVision:downscale") == 2 and exr_io.attr_value(info, "note") == "hello"
    assert np.array_equal(exr_io.attr_value(info, "AliceVision:P"), np.arange(16.0)) and exr_io.attr_value(info, "f") == 1.5
    lines = run([TOOL, "exr-info", b]).stdout
    assert "channels A B G R" in lines and "attr AliceVision:P m44d 128" in lines


@pytest.mark.parametrize("compression", [0, 2, 3])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("names", ["RGBA", "RGB", "Y"])
def test_exr_scan_lines_for_the_device(tmp_path, compression, half, names):
    """host/exr.cpp readExrLines: the container taken apart on the host (an uncompressed file mapped where it lies, ZIP / ZIPS blocks inflated), the
    scan lines handed over AS STORED with the channel offsets — de-interleaved by the oracle's restatement of avdm_image_decode_exr_lines they
    equal what the Python codec reads"""
    from oracle import oracle
    rng = np.random.RandomState(5 + compression)
    h, w = 37, 53  # (odd width: HALF lines end on a 2-byte boundary, the samples of the next line are not 4-aligned)
    chans = {n: (rng.rand(h, w) * (300.0 if n == "G" else 1.0)).astype(np.float32) for n in names}
    a, raw = str(tmp_path / "a.exr"), str(tmp_path / "a.raw")
    exr_io.write_exr(a, chans, half=half, compression=compression, data_origin=(3, 2), display_size=(80, 60))
    f = run([TOOL, "exr-lines-dump", a, raw]).stdout.split()
    assert f[0] != "refused"
    W, H, stride, nbytes, mapped = (int(v) for v in f[:5])
    off, typ = [int(v) for v in f[5:9]], [int(v) for v in f[9:13]]
    assert (W, H) == (w, h) and mapped == (1 if compression == 0 else 0)
    bpl = len(names) * w * (2 if half else 4)
    assert stride == bpl + (8 if compression == 0 else 0) and nbytes == (h - 1) * stride + bpl
    assert all(t == (1 if half else 2) for t, o in zip(typ, off) if o >= 0) and (off[3] >= 0) == ("A" in names)
    got = oracle.exr_lines_to_rgba(open(raw, "rb").read(), stride, W, H, off, typ)
    want, _ = exr_io.read_exr(a)
    for k, n in enumerate("RGBA"):
        src = want[n] if n in want else (want["Y"] if n != "A" else np.ones((h, w), np.float32))
        assert np.array_equal(got[..., k], src), n


def test_half_conversion_matches_numpy(tmp_path):
    """the C++ float -> half rounding (round to nearest even, subnormals, overflow) against numpy's"""
    vals = np.concatenate([np.linspace(-70000, 70000, 4001), np.logspace(-9, 5, 3000), -np.logspace(-9, 5, 500), [0.0, 65504.0, 65519.9, 65520.0, 5.96e-8,
                           2.98e-8, 2.99e-8, 1e-10]]).astype(np.float32)
    n = vals.size
    w = 64
    pad = (-n) % w
    arr = np.concatenate([vals, np.zeros(pad, np.float32)]).reshape(-1, w)
    a, b = str(tmp_path / "h_in.exr"), str(tmp_path / "h_out.exr")
    exr_io.write_exr(a, {"Y": arr}, half=False, compression=0)
    run([TOOL, "exr-copy", a, b, 1])
    got, _ = exr_io.read_exr(b)
    with np.errstate(over="ignore"):
        want = arr.astype(np.float16).astype(np.float32)
    assert np.array_equal(got["Y"], want)


# --------------------------------------------------------------------------------------------------- tile merge weights
def test_tile_merge_is_a_partition_of_unity(scene, tmp_path):
    sc, lms, sfm, img, d = scene
    out = str(tmp_path / "merge")
    os.makedirs(out)
    bw, bh, pad, ss = 416, 352, 32, 1
    run([TOOL, "merge-ones", sfm, img, out, 1, bw, bh, pad, ss])
    dm, dinfo = exr_io.read_exr(os.path.join(out, "%d_depthMap.exr" % scene_io.view_id(0)))
    sm, sinfo = exr_io.read_exr(os.path.join(out, "%d_simMap.exr" % scene_io.view_id(0)))
    want = np.zeros((480, 640), np.float32)
    for roi in ho.tile_roi_list(bw, bh, pad, 640, 480, ss):
        wmap, (bx, ex, by, ey) = ho.tile_weight_map(roi, 640, 480, pad, ss)
        want[by:ey, bx:ex] += wmap
    assert np.array_equal(dm["Y"], want)
    assert np.allclose(dm["Y"], 1.0, atol=1e-6)  # overlapping tiles blend with weights that sum to one
    assert np.allclose(sm["Y"], 0.5, atol=1e-3)
    # metadata of a merged full-size map (mapIO.cpp:440-512)
    assert exr_io.attr_value(dinfo, "AliceVision:roiEndX") == 640 and exr_io.attr_value(dinfo, "AliceVision:tileBufferWidth") == 1024
    assert exr_io.attr_value(dinfo, "AliceVision:nbDepthValues") == 640 * 480
    assert abs(exr_io.attr_value(dinfo, "AliceVision:minDepth") - 1.0) < 1e-6
    assert np.allclose(exr_io.attr_value(dinfo, "AliceVision:CArr"), sc.C[0], atol=1e-9)
    iCam = exr_io.attr_value(dinfo, "AliceVision:iCamArr").reshape(3, 3)
    assert np.allclose(iCam, np.linalg.inv(sc.R[0]) @ np.linalg.inv(sc.K), atol=1e-9)
    assert "AliceVision:SensorWidth" in dinfo["attributes"]  # the view's own metadata is carried over


def test_quarter_resolution_merge(scene, tmp_path):
    sc, lms, sfm, img, d = scene
    out = str(tmp_path / "merge4")
    os.makedirs(out)
    run([TOOL, "merge-ones", sfm, img, out, 1, 416, 352, 32, 4])
    dm, dinfo = exr_io.read_exr(os.path.join(out, "%d_depthMap.exr" % scene_io.view_id(0)))
    assert dm["Y"].shape == (120, 160)
    want = np.zeros((120, 160), np.float32)
    for roi in ho.tile_roi_list(416, 352, 32, 640, 480, 4):
        wmap, (bx, ex, by, ey) = ho.tile_weight_map(roi, 640, 480, 32, 4)
        want[by:ey, bx:ex] += wmap
    assert np.array_equal(dm["Y"], want)
    assert exr_io.attr_value(dinfo, "AliceVision:downscale") == 4


# ---------------------------------------------------------------------------------------------------------------- CLI
def test_cli_argument_errors(scene):
    sc, lms, sfm, img, d = scene
    base = [CLI, "-i", sfm, "--imagesFolder", img, "-o", os.path.join(d, "oe"), "--dryRun", 1]
    assert run(base + ["--downscale", 0], check=False).returncode == 1
    assert run(base + ["--sgmScale", 1, "--sgmStepXY", 1, "--refineScale", 2], check=False).returncode == 1  # SGM scale step < Refine scale step
    assert run(base + ["--sgmScale", 3, "--sgmStepXY", 1, "--refineScale", 2], check=False).returncode == 1  # not a multiple
    # fractional mip level (log2(3 / 1)): accepted like the reference accepts it (trilinear taps, round 3), with a warning about the slow path
    r = run(base + ["--sgmScale", 3, "--sgmStepXY", 1, "--refineScale", 1, "--refineStepXY", 1], check=False)
    assert r.returncode == 0 and "fractional mip level" in r.stdout + r.stderr
    assert run(base + ["--sgmScale", 4, "--sgmStepXY", 1, "--refineScale", 1, "--refineStepXY", 2, "--downscale", 1], check=False).returncode == 0
    # filtering axes: anything but one or two of 'X' / 'Y' is refused (the reference throws on unknown characters)
    assert run(base + ["--sgmFilteringAxes", "YZ"], check=False).returncode == 1
    assert run(base + ["--sgmFilteringAxes", "YXY"], check=False).returncode == 1
    assert run(base + ["--sgmFilteringAxes", "X", "--downscale", 1], check=False).returncode == 0
    assert run(base + ["--minViewAngle", 80, "--maxViewAngle", 70], check=False).returncode == 1
    assert run(base + ["--nosuchflag", 1], check=False).returncode == 1
    assert run(base + ["--sgmWSH", "abc"], check=False).returncode == 1
    assert run(base + ["--rangeStart", -1, "--rangeSize", 2], check=False).returncode == 1
    r = run(base + ["--rangeStart", 50, "--rangeSize", 2, "--downscale", 1], check=False)
    assert r.returncode == 0 and "No camera to process" in r.stdout
    assert run(base + ["--sgmUseCustomPatchPattern", 1], check=False).returncode == 1  # no subparts given
    assert run(base + ["--sgmUseCustomPatchPattern", 1, "--customPatchPatternSubparts", "circle:5:8:0"], check=False).returncode == 1  # bad token
    r = run(base + ["--refineUseCustomPatchPattern", 1, "--customPatchPatternSubparts", "circle:5:8:0:0.5", "full:2:0:1:0.5",
                    "--customPatchPatternGroupSubpartsPerLevel", 0, "--downscale", 1], check=False)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]  # multitoken option followed by another option
    assert run([CLI, "--help"], check=False).returncode == 0
    r = run([CLI, "-i", os.path.join(d, "nope.abc"), "--imagesFolder", img, "-o", d, "--dryRun", 1], check=False)
    assert r.returncode == 1 and "Alembic" in r.stdout + r.stderr


def test_sfm_reader_variants(scene, tmp_path):
    """bare (unquoted) numbers and an older file version with pxFocalLength are read to the same cameras"""
    sc, lms, sfm, img, d = scene
    doc = json.load(open(sfm))

    def unquote(o):
        if isinstance(o, dict):
            return {k: (v if k in ("path", "type", "serialNumber", "descType", "initializationMode", "distortionType", "undistortionType",
                                   "distortionInitializationMode") else unquote(v)) for k, v in o.items()}
        if isinstance(o, list):
            return [unquote(v) for v in o]
        try:
            return int(o)
        except (TypeError, ValueError):
            try:
                return float(o)
            except (TypeError, ValueError):
                return o

    bare = unquote(doc)
    bare["version"] = ["1", "2", "11"]
    p2 = str(tmp_path / "bare.sfm")
    json.dump(bare, open(p2, "w"))
    old = json.loads(json.dumps(doc))
    old["version"] = ["1", "2", "1"]
    for it in old["intrinsics"]:
        it["pxFocalLength"] = [str(sc.K[0, 0]), str(sc.K[1, 1])]
        it["type"] = "radial3"
        it["distortionParams"] = ["0", "0", "0"]
        del it["focalLength"], it["distortionType"]
    p3 = str(tmp_path / "old.sfm")
    json.dump(old, open(p3, "w"))
    ref = plan_of(sfm, img, str(tmp_path / "o_ref"), ["--sgmMaxDepths", 48])
    for p in (p2, p3):
        got = plan_of(p, img, str(tmp_path / "o_x"), ["--sgmMaxDepths", 48])
        for ta, tb in zip(ref["tiles"], got["tiles"]):
            assert ta["sgmTCams"] == tb["sgmTCams"] and np.allclose(ta["depths"], tb["depths"], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ depth-map filtering (host side)
FILTER_CLI = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_depthMapFiltering")


def test_png_codec_round_trips(tmp_path):
    """host PNG codec (nmodMap files, Fuser.cpp:220-223) against an independent reader / writer: every scan-line filter, several IDAT
    chunks, an RGB file (first channel is taken)"""
    from png_util import read_png_gray8, write_png
    rng = np.random.RandomState(2)
    img = (rng.randint(0, 11, (37, 53)) * (rng.uniform(size=(37, 53)) < 0.7)).astype(np.uint8)
    img[5:9, :] = 255
    for k, (filters, split) in enumerate([(None, None), ([1], 64), ([2, 4], None), ([0, 1, 2, 3, 4], 100)]):
        src, dst = str(tmp_path / ("in%d.png" % k)), str(tmp_path / ("out%d.png" % k))
        write_png(src, img, filters=filters, idat_split=split)
        assert np.array_equal(read_png_gray8(src), img)
        r = run([TOOL, "png-copy", src, dst])
        assert r.stdout.split() == ["53", "37"]
        assert np.array_equal(read_png_gray8(dst), img)
    rgb = np.stack([img, 255 - img, img // 2], axis=-1)
    write_png(str(tmp_path / "rgb.png"), rgb, filters=[4])
    run([TOOL, "png-copy", str(tmp_path / "rgb.png"), str(tmp_path / "rgb_out.png")])
    assert np.array_equal(read_png_gray8(str(tmp_path / "rgb_out.png")), img)
    bad = str(tmp_path / "bad.png")
    data = bytearray(open(str(tmp_path / "in0.png"), "rb").read())
    data[60] ^= 0x55
    open(bad, "wb").write(bytes(data))
    assert run([TOOL, "png-copy", bad, str(tmp_path / "x.png")], check=False).returncode == 1  # CRC mismatch


def test_filtering_sees_the_cameras_of_the_depth_maps(scene, tmp_path):
    """MultiViewParams built the way main_depthMapFiltering.cpp:103 does: sizes and P come from the depth maps' metadata (here maps at
    half the image resolution), the neighbour ranking from the landmarks"""
    sc, lms, sfm, img, d = scene
    n, ds = len(sc.R), 2
    w, h = sc.width // ds, sc.height // ds
    maps = [np.full((h, w), 4.0, np.float32) for _ in range(n)]
    folder = str(tmp_path / "dm")
    write_depth_maps(folder, sc, maps, [np.full((h, w), -0.5, np.float32)] * n, downscale=ds)
    info = json.loads(run([TOOL, "fuse-cameras", sfm, folder, str(tmp_path / "flt"), 4]).stdout)
    assert len(info["cams"]) == n
    Ks = np.diag([1.0 / ds, 1.0 / ds, 1.0]) @ sc.K
    want_rank = ho.nearest_cams_from_landmarks(ho.Cameras(sc.K, sc.R, sc.C, sc.width, sc.height), lms, 0, 4)
    for i, c in enumerate(info["cams"]):
        assert (c["viewId"], c["width"], c["height"]) == (scene_io.view_id(i), w, h)
        P = np.array(c["P"]).reshape(3, 4)
        want = Ks @ np.concatenate([sc.R[i], (-sc.R[i] @ sc.C[i])[:, None]], axis=1)
        assert np.allclose(P / P[2, 3] if abs(P[2, 3]) > 1e-9 else P, want / want[2, 3] if abs(want[2, 3]) > 1e-9 else want, rtol=1e-9, atol=1e-9)
        assert np.allclose(np.array(c["C"]), sc.C[i], atol=1e-9)
        iP = np.array(c["iP"]).reshape(3, 3)
        assert np.allclose(iP @ (Ks @ sc.R[i]), np.eye(3), atol=1e-9)
        assert len(c["tcams"]) == 4 and i not in c["tcams"]
    assert info["cams"][0]["tcams"] == list(want_rank)


def test_filtering_cli_argument_errors(scene, tmp_path):
    sc, lms, sfm, img, d = scene
    assert run([FILTER_CLI, "--help"], check=False).returncode == 0
    assert run([FILTER_CLI, "-i", sfm, "-o", str(tmp_path)], check=False).returncode == 1          # --depthMapsFolder is required
    assert run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", d, "-o", str(tmp_path), "--nNearestCams", "x"], check=False).returncode == 1
    r = run([FILTER_CLI, "-i", str(tmp_path / "nope.sfm"), "--depthMapsFolder", d, "-o", str(tmp_path)], check=False)
    assert r.returncode == 1 and "cannot be read" in r.stdout + r.stderr
    r = run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", d, "-o", str(tmp_path), "--rangeStart", 90, "--rangeSize", 2], check=False)
    assert r.returncode == 0 and "No camera to process" in r.stdout
    assert run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", d, "-o", str(tmp_path), "--rangeStart", -1, "--rangeSize", 2], check=False).returncode == 1
    assert run([FILTER_CLI, "-i", sfm, "--depthMapsFolder", d, "-o", str(tmp_path), "--pixSizeBall", -1], check=False).returncode == 1


def test_sgm_running_average_integer_forms():
    """csrc/avdm_sgm.hip replaces the reference's (uint8)((out * K + clamp(L)) / (K + 1)) (kernels.cuh:741-743) by integer forms on packed
    uint16 pairs: n >> 1, n >> 2 and, for K = 2, (n * 21856) >> 16 (two v_mul_u32_u24_sdwa + one v_perm_b32) == (n * 683) >> 11 — exact
    for every n = out * K + clamp(L) <= 1020; and the fp32 expression of the reference truncates to the same integer."""
    n = np.arange(0, 1021, dtype=np.int64)
    assert np.array_equal((n * 21856) >> 16, n // 3)
    assert np.array_equal((n * 683) >> 11, n // 3)
    for K in (1, 2, 3):
        o, c = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
        ref = ((o * np.float32(K) + c) / np.float32(K + 1)).astype(np.uint8)  # fp32 division, truncating cast
        assert np.array_equal(ref, ((o.astype(np.int64) * K + c.astype(np.int64)) // (K + 1)).astype(np.uint8)), K


def test_prepare_dense_scene_cli_argument_errors(scene, tmp_path):
    """aliceVision_prepareDenseScene: the reference's flags; what this implementation does not build is refused, not ignored"""
    sc, lms, sfm, img, d = scene
    exe = os.path.join(ROOT, "alicevision_amd", "bin", "aliceVision_prepareDenseScene")
    assert os.path.exists(exe)
    base = [exe, "-i", sfm, "-o", str(tmp_path / "prep")]
    assert run([exe, "--help"], check=False).returncode == 0
    assert run([exe, "-i", sfm], check=False).returncode == 1                                  # --output is required
    assert run(base + ["--outputFileType", "jpg"], check=False).returncode == 1               # exr only
    assert run(base + ["--evCorrection", 1], check=False).returncode == 1
    assert run(base + ["--rangeStart", 2, "--rangeSize", -1], check=False).returncode == 1   # Range is incorrect
    r = run([exe, "-i", str(tmp_path / "nope.sfm"), "-o", str(tmp_path / "prep")], check=False)
    assert r.returncode == 1 and "cannot be read" in r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------------------- PNG input
@pytest.mark.parametrize("mode,bits", [("L", 8), ("LA", 8), ("RGB", 8), ("RGBA", 8), ("I;16", 16)])
def test_png_reader_against_pillow(tmp_path, mode, bits):
    """host/png.cpp readPng (the decoder behind ImagesCache for <viewId>.png): files written by an independent encoder (Pillow, which picks
    its own scan-line filters) come back sample for sample"""
    from PIL import Image
    rng = np.random.default_rng(7)
    w, h = 67, 45
    ch = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4, "I;16": 1}[mode]
    if bits == 8:
        arr = rng.integers(0, 256, size=(h, w, ch), dtype=np.uint8)
        arr[:, :, 0] = (np.add.outer(np.arange(h), np.arange(w)) * 3 % 256).astype(np.uint8)  # smooth: makes the encoder use sub / up / paeth
        img = Image.fromarray(arr[:, :, 0] if ch == 1 else arr, mode)
    else:
        arr = (np.add.outer(np.arange(h), np.arange(w)) * 517 % 65536).astype(np.uint16).reshape(h, w, 1)
        img = Image.fromarray(arr[:, :, 0], "I;16")
    path, raw = str(tmp_path / "a.png"), str(tmp_path / "a.raw")
    img.save(path)
    out = run([TOOL, "png-dump", path, raw]).stdout.split()
    assert [int(v) for v in out] == [w, h, ch, bits]
    got = np.fromfile(raw, dtype=np.uint8 if bits == 8 else np.uint16).reshape(h, w, ch)
    assert np.array_equal(got, arr)


@pytest.mark.parametrize("ch,bits", [(1, 8), (2, 16), (3, 16), (4, 8), (4, 16)])
def test_png_writer_reader_roundtrip_all_filters(tmp_path, ch, bits):
    """writePng cycles through the five scan-line filter types row by row; readPng must undo each of them (8- and 16-bit, 1-4 channels:
    16-bit RGB / RGBA is what Pillow cannot write), and Pillow reads the 8-bit files back identically"""
    rng = np.random.default_rng(ch * 100 + bits)
    w, h = 53, 31
    dt = np.uint8 if bits == 8 else np.uint16
    arr = rng.integers(0, 1 << bits, size=(h, w, ch)).astype(dt)
    raw, path, raw2 = str(tmp_path / "in.raw"), str(tmp_path / "b.png"), str(tmp_path / "out.raw")
    arr.tofile(raw)
    run([TOOL, "png-write", raw, path, w, h, ch, bits])
    out = run([TOOL, "png-dump", path, raw2]).stdout.split()
    assert [int(v) for v in out] == [w, h, ch, bits]
    assert np.array_equal(np.fromfile(raw2, dtype=dt).reshape(h, w, ch), arr)
    if bits == 8:
        from PIL import Image
        back = np.asarray(Image.open(path))
        assert np.array_equal(back.reshape(h, w, ch), arr)


# ------------------------------------------------------------------------------------------------ Alembic (.abc) SfMData
ABC_GOLDEN = os.path.join(ROOT, "tests", "golden", "alembic")


def _gunzip(name, tmp_path):
    import gzip
    dst = str(tmp_path / name[:-3])
    with gzip.open(os.path.join(ABC_GOLDEN, name), "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


def _sfm_dump(path):
    return json.loads(run([TOOL, "sfm-dump", path]).stdout)


@pytest.mark.parametrize("version", ["1.2.0", "1.2.2", "1.2.3", "1.2.8", "1.2.11"])
def test_alembic_reader_on_the_reference_compatibility_scenes(version, tmp_path):
    """The reference's own scene_v<version>.abc (sfmDataIO/compatibilityData, written by Alembic 1.7.16 / 1.8.4 through its exporter of
    that version) read by alembic.cpp == the scene as its newest .json twin states it (tests/golden/make_alembic_fixtures.py).
    Follows sfmDataIOCompatibility_test.cpp, which loads each file and compares it with the generated sample scene."""
    exp = json.load(open(os.path.join(ABC_GOLDEN, "expected.json")))
    got = _sfm_dump(_gunzip("scene_v%s.abc.gz" % version, tmp_path))
    assert [{k: v[k] for k in ("viewId", "poseId", "intrinsicId", "path", "width", "height", "metadata")} for v in got["views"]] == \
        sorted(exp["views"], key=lambda v: v["viewId"])
    assert len(got["intrinsics"]) == len(exp["intrinsics"]) == 2
    for g, e in zip(got["intrinsics"], sorted(exp["intrinsics"], key=lambda i: i["intrinsicId"])):
        assert (g["intrinsicId"], g["type"], g["distortionType"], g["width"], g["height"], g["isPinhole"]) == \
            (e["intrinsicId"], e["type"], e["distortionType"], e["width"], e["height"], 1)
        assert g["sensorWidth"] == e["sensorWidth"] and g["sensorHeight"] == e["sensorHeight"] and g["distortionParams"] == e["distortionParams"]
        np.testing.assert_allclose(g["scale"], e["scale"], rtol=1e-12)
        np.testing.assert_allclose(g["offset"], e["offset"], rtol=1e-12)
    assert [p["poseId"] for p in got["poses"]] == sorted(p["poseId"] for p in exp["poses"])
    for g, e in zip(got["poses"], sorted(exp["poses"], key=lambda p: p["poseId"])):
        np.testing.assert_allclose(g["rotation"], e["rotation"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(g["center"], e["center"], rtol=0, atol=1e-14)
    assert len(got["landmarks"]) == exp["n_landmarks"] and all(l["obs"] == [] for l in got["landmarks"])
    for e in exp["landmarks"]:
        assert got["landmarks"][e["id"]]["id"] == e["id"] and got["landmarks"][e["id"]]["X"] == e["X"]


def test_alembic_writer_equals_the_reference_file_entry_by_entry(tmp_path):
    """scene_v1.2.11.abc -> SfMData -> saveSfMDataAlembic: every object, property header, metadata string and sample blob (digest
    included) of the archive written here equals the reference's own file — read back by a second, independent parser
    (tests/abc_explorer.py) — except the camera transforms' values, which differ by signed zeros / the last ulp of the 4 x 4 inverse."""
    import abc_explorer
    src = _gunzip("scene_v1.2.11.abc.gz", tmp_path)
    out = str(tmp_path / "written.abc")
    run([TOOL, "sfm-to-abc", src, out])
    ref, mine = abc_explorer.flatten(open(src, "rb").read()), abc_explorer.flatten(open(out, "rb").read())
    assert set(ref) == set(mine)
    differing = [k for k in ref if ref[k] != mine[k]]
    assert differing and all(k.endswith("/.xform/.vals") for k in differing), differing[:5]
    for k in differing:
        assert ref[k][:5] == mine[k][:5]
        a, b = (np.frombuffer(x[5][0][16:], dtype="<f8") for x in (ref[k], mine[k]))
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-14)


def test_alembic_round_trip_with_observations_and_cli_plan(scene, tmp_path):
    """a scene WITH observations and image paths: .sfm -> .abc -> the same SfMData (positions and observations to float precision — the
    archive holds them as float32 like the reference's exporter), and the program plans the same work from either file"""
    sc, lms, sfm, img, d = scene
    abc_path = str(tmp_path / "scene.abc")
    run([TOOL, "sfm-to-abc", sfm, abc_path])
    a, b = _sfm_dump(sfm), _sfm_dump(abc_path)
    strip = lambda vs: [{k: x for k, x in v.items() if not k.startswith("abs")} for v in vs]
    assert strip(a["views"]) == strip(b["views"]) and [p["poseId"] for p in a["poses"]] == [p["poseId"] for p in b["poses"]]
    for va, vb in zip(a["views"], b["views"]):
        np.testing.assert_allclose(va["absRotation"] + va["absCenter"], vb["absRotation"] + vb["absCenter"], rtol=0, atol=1e-13)
    for ia, ib in zip(a["intrinsics"], b["intrinsics"]):
        assert {k: v for k, v in ia.items() if k not in ("scale", "offset")} == {k: v for k, v in ib.items() if k not in ("scale", "offset")}
        np.testing.assert_allclose(ia["scale"] + ia["offset"], ib["scale"] + ib["offset"], rtol=1e-15)
    for pa, pb in zip(a["poses"], b["poses"]):
        np.testing.assert_allclose(pa["rotation"] + pa["center"], pb["rotation"] + pb["center"], rtol=0, atol=1e-13)
    assert len(a["landmarks"]) == len(b["landmarks"]) > 100 and sum(len(l["obs"]) for l in a["landmarks"]) > 1000
    for la, lb in zip(a["landmarks"], b["landmarks"]):
        np.testing.assert_allclose(la["X"], lb["X"], rtol=2e-7)
        assert [o[0] for o in la["obs"]] == [o[0] for o in lb["obs"]]
        np.testing.assert_allclose([o[1:] for o in la["obs"]], [o[1:] for o in lb["obs"]], rtol=2e-7)
    ref = plan_of(sfm, img, str(tmp_path / "o_sfm"), ["--sgmMaxDepths", 48])
    got = plan_of(abc_path, img, str(tmp_path / "o_abc"), ["--sgmMaxDepths", 48])
    assert len(ref["tiles"]) == len(got["tiles"]) > 0
    for ta, tb in zip(ref["tiles"], got["tiles"]):
        assert ta["sgmTCams"] == tb["sgmTCams"] and np.allclose(ta["depths"], tb["depths"], rtol=1e-5)


def test_alembic_reader_rejects_what_it_cannot_read(tmp_path):
    bad = str(tmp_path / "bad.abc")
    open(bad, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    r = run([TOOL, "sfm-dump", bad], check=False)
    assert r.returncode == 1 and "HDF5" in r.stderr
    src = _gunzip("scene_v1.2.11.abc.gz", tmp_path)
    data = open(src, "rb").read()
    open(bad, "wb").write(data[:len(data) // 2])  # truncated: the root group position points past the end
    assert run([TOOL, "sfm-dump", bad], check=False).returncode == 1
    open(bad, "wb").write(data[:5] + b"\x00" + data[6:])  # not frozen
    r = run([TOOL, "sfm-dump", bad], check=False)
    assert r.returncode == 1 and "frozen" in r.stderr


def test_jet_colour_map_of_the_volume_exports():
    """image/jetColorMap.cpp holds MATLAB's jet(64) as three tables; jet's closed form is entry i = clamp(1.5 - |4 (i + 1) / 64 - c|, 0, 1)
    with c = 3, 2, 1 for red, green, blue.  The reference interpolates the table linearly at value * 63 and truncates to 8 bits."""
    table = np.stack([np.clip(1.5 - np.abs(4.0 * (np.arange(64) + 1) / 64.0 - c), 0.0, 1.0) for c in (3.0, 2.0, 1.0)], axis=1).astype(np.float32)
    values = np.concatenate([[-0.5, 0.0, 1.0, 1.5], np.linspace(0.001, 0.999, 97)]).astype(np.float32)
    got = np.array([[int(t) for t in l.split()] for l in run([TOOL, "jet"] + ["%.9g" % v for v in values]).stdout.strip().splitlines()])
    for v, g in zip(values, got):
        if v <= 0:
            want = [0, 0, 0]
        elif v >= 1:
            want = [255, 255, 255]
        else:
            f = np.float32(v) * np.float32(63.0)
            i = int(np.floor(f))
            b = np.float32(f - np.float32(i))
            a = np.float32(1.0) - b
            want = [int(np.float32(np.float32(table[i, k] * a + table[i + 1, k] * b) * np.float32(255.0))) for k in range(3)]
        assert list(g) == want, (v, g, want)


def test_jpeg_images_folder_is_planned_like_the_exr_folder(scene, tmp_path):
    """<viewId>.jpg in --imagesFolder: the program finds the files, reads their size from the frame header and plans the same work; a
    second file for a view is refused like in the reference (MultiViewParams.cpp:96-100)"""
    Image = pytest.importorskip("PIL.Image")
    sc, lms, sfm, img, d = scene
    jpg = str(tmp_path / "jpg")
    os.makedirs(jpg)
    for i in range(6):
        a = np.clip(sc.images[i].numpy()[..., :3] * 255.0, 0, 255).astype(np.uint8)
        Image.fromarray(a).save(os.path.join(jpg, "%d.jpg" % scene_io.view_id(i)), quality=90, progressive=bool(i % 2))
    ref = plan_of(sfm, img, str(tmp_path / "o_exr"), ["--sgmMaxDepths", 48])
    got = plan_of(sfm, jpg, str(tmp_path / "o_jpg"), ["--sgmMaxDepths", 48])
    assert len(ref["tiles"]) == len(got["tiles"]) > 0
    for ta, tb in zip(ref["tiles"], got["tiles"]):
        assert ta["sgmTCams"] == tb["sgmTCams"] and ta["depths"] == tb["depths"]
    import shutil
    shutil.copy(os.path.join(img, "%d.exr" % scene_io.view_id(0)), jpg)
    r = run([CLI, "-i", sfm, "--imagesFolder", jpg, "-o", str(tmp_path / "o_x"), "--downscale", 1, "--dryRun", 1, "-v", "error"], check=False)
    assert r.returncode == 1 and "Ambiguous" in (r.stdout + r.stderr)


def test_rig_scene_equals_the_scene_with_absolute_poses(scene, tmp_path):
    """a scene whose cameras hang on a rig (sfmData/Rig.hpp: pose of a view = sub-pose composed with the rig's pose, SfMData::getPose) read from
    .sfm and — written with the rig structure of AlembicExporter.cpp:411-486 and read back — from .abc: the same absolute cameras and the same
    plan as the scene that states every pose on its own; a view with an uninitialised sub-pose is not a camera of the stage"""
    sc, lms, sfm, img, d = scene
    doc = json.load(open(sfm))
    R = [np.asarray(sc.R[i], np.float64) for i in range(6)]
    C = [np.asarray(sc.C[i], np.float64) for i in range(6)]
    H = [np.block([[R[i], (-R[i] @ C[i])[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]]) for i in range(6)]
    rig = json.loads(json.dumps(doc))
    rig_pose_id = doc["views"][0]["poseId"]
    sub = []
    for i, v in enumerate(rig["views"]):
        v["rigId"], v["subPoseId"], v["isPoseIndependant"], v["poseId"] = "7", str(i), "false", rig_pose_id
        Hs = H[i] @ np.linalg.inv(H[0])  # H_i = H_sub * H_rig
        Rs, ts = Hs[:3, :3], Hs[:3, 3]
        Cs = -Rs.T @ ts
        sub.append({"status": "estimated", "pose": {"rotation": ["%.17g" % x for x in Rs.T.reshape(-1)], "center": ["%.17g" % x for x in Cs]}})
    rig["poses"] = [p for p in doc["poses"] if p["poseId"] == rig_pose_id]
    rig["rigs"] = [{"rigId": "7", "subPoses": sub}]
    rig_sfm = str(tmp_path / "rig.sfm")
    json.dump(rig, open(rig_sfm, "w"))
    rig_abc = str(tmp_path / "rig.abc")
    run([TOOL, "sfm-to-abc", rig_sfm, rig_abc])
    plain = _sfm_dump(sfm)
    for path in (rig_sfm, rig_abc):
        got = _sfm_dump(path)
        assert len(got["poses"]) == 1 and [v["rigId"] for v in got["views"]] == [7] * 6 and [v["subPoseId"] for v in got["views"]] == list(range(6))
        for a, b in zip(plain["views"], got["views"]):
            np.testing.assert_allclose(b["absRotation"], a["absRotation"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(b["absCenter"], a["absCenter"], rtol=0, atol=1e-11)
        ref = plan_of(sfm, img, str(tmp_path / "o_plain"), ["--sgmMaxDepths", 48])
        out = plan_of(path, img, str(tmp_path / "o_rig"), ["--sgmMaxDepths", 48])
        for ta, tb in zip(ref["tiles"], out["tiles"]):
            assert ta["sgmTCams"] == tb["sgmTCams"] and np.allclose(ta["depths"], tb["depths"], rtol=1e-5)
    # an uninitialised sub-pose: that view has no pose (SfMData.hpp:288-295), the others keep theirs
    rig["rigs"][0]["subPoses"][2]["status"] = "uninitialized"
    json.dump(rig, open(rig_sfm, "w"))
    got = _sfm_dump(rig_sfm)
    assert ["absCenter" in v for v in got["views"]] == [True, True, False, True, True, True]
    run([TOOL, "sfm-to-abc", rig_sfm, rig_abc])
    got = _sfm_dump(rig_abc)
    assert ["absCenter" in v for v in got["views"]] == [True, True, False, True, True, True]


# ------------------------------------------------------------------------------------------------ TIFF
def _write_tiff(path, a, big_endian=False, tile=None, planar=False, white_is_zero=False):
    """a minimal classic-TIFF writer for the layouts Pillow cannot produce: a (H, W, C) uint8 / uint16 array, uncompressed, in strips of 7
    rows or in tiles, chunky or planar, either byte order"""
    import struct
    H, W, C = a.shape
    bits = 8 * a.dtype.itemsize
    e = ">" if big_endian else "<"
    data = a.astype(a.dtype.newbyteorder(e))
    chunks = []
    planes = [data[..., c:c + 1] for c in range(C)] if planar else [data]
    if tile:
        tw, th = tile
        for pl in planes:
            for y0 in range(0, H, th):
                for x0 in range(0, W, tw):
                    t = np.zeros((th, tw, pl.shape[2]), pl.dtype)
                    blk = pl[y0:y0 + th, x0:x0 + tw]
                    t[:blk.shape[0], :blk.shape[1]] = blk
                    chunks.append(t.tobytes())
    else:
        rps = 7
        for pl in planes:
            for y0 in range(0, H, rps):
                chunks.append(pl[y0:y0 + rps].tobytes())
    entries = []

    def ent(tag, typ, vals):
        entries.append((tag, typ, list(vals)))

    ent(256, 4, [W]), ent(257, 4, [H]), ent(258, 3, [bits] * C), ent(259, 3, [1])
    ent(262, 3, [2 if C >= 3 else (0 if white_is_zero else 1)]), ent(277, 3, [C]), ent(284, 3, [2 if planar else 1])
    if C in (2, 4):
        ent(338, 3, [2])
    if tile:
        ent(322, 4, [tile[0]]), ent(323, 4, [tile[1]])
    else:
        ent(278, 4, [7])
    off_tag, cnt_tag = (324, 325) if tile else (273, 279)
    ent(off_tag, 4, [0] * len(chunks)), ent(cnt_tag, 4, [len(c) for c in chunks])
    entries.sort()
    ifd_size = 2 + 12 * len(entries) + 4
    ext_off = 8 + ifd_size
    blobs, body = [], b""
    for tag, typ, vals in entries:
        fmt = {3: "H", 4: "I"}[typ]
        raw = struct.pack(e + fmt * len(vals), *vals)
        blobs.append(raw)
    # layout: header, IFD, out-of-line values, pixel data
    pos = ext_off
    where = []
    for raw in blobs:
        if len(raw) > 4:
            where.append(pos)
            pos += len(raw) + (len(raw) & 1)
        else:
            where.append(None)
    data_off = pos
    offs = []
    for c in chunks:
        offs.append(data_off)
        data_off += len(c)
    out = (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, 8) + struct.pack(e + "H", len(entries))
    ext = b""
    for (tag, typ, vals), raw, w in zip(entries, blobs, where):
        if tag == off_tag:
            raw = struct.pack(e + "I" * len(offs), *offs)
        if w is None:
            out += struct.pack(e + "HHI", tag, typ, len(vals)) + raw.ljust(4, b"\0")
        else:
            out += struct.pack(e + "HHII", tag, typ, len(vals), w)
            ext += raw + (b"\0" if len(raw) & 1 else b"")
    out += struct.pack(e + "I", 0) + ext + b"".join(chunks)
    open(path, "wb").write(out)


def _tiff_dump(path, tmp_path):
    raw = str(tmp_path / "tiff.raw")
    r = run([TOOL, "tiff-dump", path, raw])
    w, h, c, b, o = (int(v) for v in r.stdout.split())
    return np.fromfile(raw, np.uint8 if b == 8 else np.uint16).reshape(h, w, c)


def test_tiff_reader_layouts(tmp_path):
    """strips and tiles, chunky and planar, both byte orders, 8 and 16 bits, 1 to 4 channels, WhiteIsZero: the samples as stored"""
    rng = np.random.default_rng(8)
    p = str(tmp_path / "t.tif")
    for dt in (np.uint8, np.uint16):
        for C in (1, 2, 3, 4):
            a = rng.integers(0, np.iinfo(dt).max, size=(45, 67, C)).astype(dt)
            for kw in ({}, {"big_endian": True}, {"tile": (32, 16)}, {"planar": True}, {"tile": (16, 32), "planar": True, "big_endian": True}):
                _write_tiff(p, a, **kw)
                assert np.array_equal(_tiff_dump(p, tmp_path), a), (dt, C, kw)
    a = rng.integers(0, 255, size=(9, 11, 1)).astype(np.uint8)
    _write_tiff(p, a, white_is_zero=True)
    assert np.array_equal(_tiff_dump(p, tmp_path), 255 - a)
    open(p, "wb").write(b"II+\0" + b"\0" * 32)
    r = run([TOOL, "tiff-dump", p, str(tmp_path / "x.raw")], check=False)
    assert r.returncode == 1 and "BigTIFF" in r.stderr


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_tiff_reader_against_pillow(tmp_path, compression):
    """files written by Pillow / libtiff: LZW, Deflate, PackBits, with and without the horizontal predictor, 8-bit RGB(A) / grey and 16-bit grey"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(9)
    y, x = np.mgrid[0:211, 0:300]
    base = (np.sin(x / 9.0) * np.cos(y / 7.0) * 0.4 + 0.5)
    for mode, C, dt in (("RGB", 3, np.uint8), ("RGBA", 4, np.uint8), ("L", 1, np.uint8), ("I;16", 1, np.uint16)):
        mx = np.iinfo(dt).max
        a = np.clip(np.stack([base * mx + rng.normal(0, mx * 0.01, base.shape) for _ in range(C)], -1), 0, mx).astype(dt)
        for predictor in ((1,) if compression in (None, "packbits") else (1, 2)):
            p = str(tmp_path / "p.tif")
            kw = {"compression": compression} if compression else {}
            if predictor == 2:
                kw["tiffinfo"] = {317: 2}
            Image.fromarray(a[..., 0] if C == 1 else a).save(p, **kw)
            ref = np.array(Image.open(p))
            got = _tiff_dump(p, tmp_path)
            assert np.array_equal(got, ref.reshape(got.shape)), (mode, compression, predictor)


def test_tiff_images_folder_is_planned_like_the_exr_folder(scene, tmp_path):
    """<viewId>.tif in --imagesFolder: found, sized from the directory, the same plan"""
    sc, lms, sfm, img, d = scene
    tif = str(tmp_path / "tif")
    os.makedirs(tif)
    for i in range(6):
        a = np.clip(sc.images[i].numpy()[..., :3] * 65535.0, 0, 65535).astype(np.uint16)
        _write_tiff(os.path.join(tif, "%d.tif" % scene_io.view_id(i)), a, tile=(128, 64) if i % 2 else None)
    ref = plan_of(sfm, img, str(tmp_path / "o_exr"), ["--sgmMaxDepths", 48])
    got = plan_of(sfm, tif, str(tmp_path / "o_tif"), ["--sgmMaxDepths", 48])
    assert len(ref["tiles"]) == len(got["tiles"]) > 0
    for ta, tb in zip(ref["tiles"], got["tiles"]):
        assert ta["sgmTCams"] == tb["sgmTCams"] and ta["depths"] == tb["depths"]


def test_view_exposures_from_metadata(scene, tmp_path):
    """ImageInfo::getCameraExposureSetting (sfmData/ImageInfo.{hpp,cpp}: key lookup exact, then case-insensitive on the part behind the last
    '/' or ':'; "1/200"-style fractions; FNumber else 2^(ApertureValue / 2); four names for the ISO) and the scene's median exposure
    (SfMData.hpp:406-426: over the DISTINCT exposures), restated here in Python with the formula pinned separately to the reference's own
    class (test_host_ref.py::test_exposure_setting_equals_reference)"""
    import re
    sc, lms, sfm, img, d = scene
    doc = json.load(open(sfm))
    metas = [
        {"ExposureTime": "1/200", "FNumber": "2.8", "Exif:PhotographicSensitivity": "400"},
        {"Exif:ExposureTime": "0.01", "Exif:FNumber": "4", "ISO": "100"},
        {"exif/shutter speed value": "1/60", "ApertureValue": "3", "PhotographicSensitivity": "200"},
        {"ExposureTime": "1/200", "FNumber": "2.8", "Exif:PhotographicSensitivity": "400", "Make": "x"},   # the same exposure as view 0
        {"FNumber": "abc", "Aperture Value": "5", "Photographic Sensitivity": "800"},                          # no shutter: 1 / 200
        {"Make": "nothing about exposure"},
    ]
    for v, m in zip(doc["views"], metas):
        v["metadata"] = m
    p = str(tmp_path / "ev.sfm")
    json.dump(doc, open(p, "w"))
    lines = run([TOOL, "exposures", p]).stdout.strip().splitlines()

    def find(md, name):
        if name in md:
            return md[name]
        for k, v in md.items():  # std::map order = sorted keys
            key = k.lower()
            if len(key) > len(name):
                i = max(key.rfind("/"), key.rfind(":"))
                if i >= 0:
                    key = key[i + 1:]
            if key == name.lower():
                return v
        return None

    def first(md, names):
        for n in names:
            v = find(dict(sorted(md.items())), n)
            if v is not None:
                return v
        return None

    def real(s):
        m = re.search(r"([0-9]+)/([0-9]+)", s)
        try:
            if not m:
                return float(s)
            return int(m.group(1)) / int(m.group(2)) if int(m.group(2)) else 0.0
        except ValueError:
            return -1.0

    def has_digit(md, names):
        for n in names:
            v = find(dict(sorted(md.items())), n)
            if not v:
                continue
            try:
                return float(v) > 0
            except ValueError:
                pass
        return False

    def get_double(md, names):
        v = first(md, names)
        return -1.0 if not v else real(v)

    def exposure(md):
        sh = get_double(md, ["ExposureTime", "Shutter Speed Value"])
        fn = -1.0
        if has_digit(md, ["FNumber"]):
            fn = get_double(md, ["FNumber"])
        elif has_digit(md, ["ApertureValue", "Aperture Value"]):
            fn = 2.0 ** (get_double(md, ["ApertureValue", "Aperture Value"]) / 2.0)
        iso = get_double(md, ["Exif:PhotographicSensitivity", "PhotographicSensitivity", "Photographic Sensitivity", "ISO"])
        ok_s, ok_f = sh > 0, fn > 0
        if not ok_s and not ok_f:
            return sh, fn, iso, -1.0
        s_, f_ = (sh if ok_s else 1.0 / 200.0), (fn if ok_f else 1.0)
        k = np.sqrt(iso / 100.0) if iso > 1e-6 else 1.0
        return sh, fn, iso, s_ * (1.0 / (f_ * k)) ** 2

    want = [exposure(m) for m in metas]
    ordered = sorted(zip([int(v["viewId"]) for v in doc["views"]], want))
    for line, (vid, w) in zip(lines, ordered):
        got = [float(x) for x in line.split()]
        assert int(got[0]) == vid
        np.testing.assert_allclose(got[1:], w, rtol=1e-15)
    distinct = []
    for w in want:
        if w[3] != -1.0 or w[0] > 0 or w[1] > 0:
            if (w[0] > 0 or w[1] > 0) and w[3] not in distinct:
                distinct.append(w[3])
    assert len(distinct) == 4
    assert float(lines[-1].split()[1]) == sorted(distinct)[len(distinct) // 2]
