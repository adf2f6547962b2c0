"""The oracle (oracle/avdm_oracle.c) against THE REFERENCE'S OWN CODE.

Two layers:
  * committed vectors (always run, also on the GPU box where /root/reference is absent): tests/golden/ref_helpers.npz and
    relief_192x144_*.npz were produced by oracle/_ref — the reference's kernel-launch layer compiled unchanged for the CPU
    (oracle/ref/ref_driver.cpp, generator tests/golden/make_golden.py) — and the oracle must reproduce them BIT FOR BIT
    (the PCA normal is the one exception: the oracle uses a different eigen-solver, compared to 1e-4);
  * live comparisons (run when the library can be built or travelled prebuilt): whole tiles through every stage in both texture filter
    modes, odd sizes, ROI offsets, non-default parameters, per-T-camera depth ranges, the optional kernels — all exact.
What stays restated on BOTH sides (and is therefore not pinned by this file): the texture unit's filtering arithmetic and the
fast-math intrinsics, which live in NVIDIA's hardware / toolkit, not in /root/reference.
"""
import ctypes as C
import os

import numpy as np
import pytest

from alicevision_amd import abi
from alicevision_amd.synthetic import plane_depths

from common import make_oracle, small_case

HERE = os.path.dirname(os.path.abspath(__file__))
F8, EX = abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT


@pytest.fixture(scope="module")
def olib():
    from oracle import oracle
    lib = oracle.load()
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    for name, args in {"avo_test_rgb2lab": [vp, i32, vp], "avo_test_cost_yk_from_lab": [vp, vp, i32, f32, f32, vp],
                       "avo_test_sim_stat_wsim": [vp, i32, i32, vp], "avo_test_sigmoid": [vp, i32, f32, f32, f32, f32, vp, vp],
                       "avo_test_project3d": [vp, vp, i32, vp]}.items():
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = args
    return lib


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "ref_helpers.npz"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(a, b, equal_nan=True):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError("%s: differs on %.3g of the entries, max |d| = %.3g" % (what, (d > 0).mean(), np.nanmax(d)))


# ---- committed vectors ---------------------------------------------------------------------------------------------
def test_colour_conversion_vector(olib, g):
    """rgb2xyz + xyz2lab (color.cuh:65-70,124-141)"""
    rgb = np.ascontiguousarray(g["rgb01"])
    out = np.empty_like(rgb)
    olib.avo_test_rgb2lab(_p(rgb), len(rgb), _p(out))
    _eq(out, g["lab"], "xyz2lab(rgb2xyz)")


def test_yoon_kweon_weight_vector(olib, g):
    """CostYKfromLab (color.cuh:167-210) at the SGM and the Refine gammas"""
    dxdy, cc = np.ascontiguousarray(g["yk_dxdy"]), np.ascontiguousarray(g["yk_c1c2"])
    for name, (gc, gp) in {"yk_sgm": (5.5, 8.0), "yk_refine": (15.5, 8.0)}.items():
        out = np.empty(len(dxdy), np.float32)
        olib.avo_test_cost_yk_from_lab(_p(dxdy), _p(cc), len(out), np.float32(1.0) / np.float32(gc), np.float32(1.0) / np.float32(gp), _p(out))
        _eq(out, g[name], name)


def test_weighted_ncc_vector(olib, g):
    """simStat::update + computeWSim (SimStat.cuh:72-113,146-153)"""
    s = np.ascontiguousarray(g["wsim_samples"])
    out = np.empty(s.shape[0], np.float32)
    olib.avo_test_sim_stat_wsim(_p(s), s.shape[1], s.shape[0], _p(out))
    _eq(out, g["wsim"], "computeWSim")
    # cases 0 / 1 have a constant channel: the fp32 variance is cancellation noise, not 0 (the ill-conditioning DESIGN.md §4.1 describes)


def test_sigmoid_vectors(olib, g):
    """sigmoid / sigmoid2 (matrix.cuh:334-346) with the parameter sets of the Refine filter, the adaptive P2 and the optimisation"""
    z = np.ascontiguousarray(g["sig_z"])
    for name, args in {"sig_refine": (0.0, 1.0, 0.7, -0.7), "sig_p2": (80.0, 255.0, 80.0, 100.0), "sig_opt": (5.0, 30.0, 40.0, 20.0)}.items():
        a, b = np.empty(len(z), np.float32), np.empty(len(z), np.float32)
        olib.avo_test_sigmoid(_p(z), len(z), *[np.float32(v) for v in args], _p(a), _p(b))
        _eq(np.stack([a, b]), g[name], name)


def test_projection_vector(olib, g):
    """project3DPoint (matrix.cuh:117-126)"""
    P, pts = np.ascontiguousarray(g["proj_P"]), np.ascontiguousarray(g["proj_pts"])
    out = np.empty((len(pts), 2), np.float32)
    olib.avo_test_project3d(_p(P), _p(pts), len(pts), _p(out))
    _eq(out, g["proj"], "project3DPoint")


def test_exp_p2_against_reference_sigmoid(olib, g):
    """the adaptive P2 (kernels.cuh:696-720) is the reference's sigmoid(80, 255, 80, P2w, deltaC) with libm's expf; the oracle and the
    HIP kernel share a fully specified polynomial instead (avo_exp_p2, so that the integer SGM stage can be compared bit for bit).
    Measured distance between the two over the reference-produced vector: a few ulp of P2, i.e. floor(P2) — all that enters the
    integer recurrence — differs only when P2 lies within ~1e-4 of an integer."""
    from oracle import oracle
    lib = oracle.load()
    z = g["sig_z"]
    want = g["sig_p2"][0]
    got = np.array([np.float32(80.0) + np.float32(175.0) * (np.float32(1.0) / (np.float32(1.0) + np.float32(lib.avo_exp_p2(np.float32(10.0) * ((np.float32(v) - np.float32(100.0)) / np.float32(80.0))))))
                    for v in z], np.float32)
    assert np.abs(got - want).max() < 1e-4
    assert (np.floor(got) != np.floor(want)).mean() < 0.01


def test_pyramids_and_texture_unit_vectors(g):
    """DeviceMipmapImage::fill (image * 255 -> half, optional Gaussian downscale, rgb2lab, the 5 x 5 mip kernel with its unsigned-wrap
    border taps) and tex2DLod probes, both filter modes, min downscale 1 and 2"""
    from oracle import oracle
    rgba = g["img8"].astype(np.float32) / np.float32(255.0)
    uvl = g["tex_uvl"]
    for mode, tag in ((F8, "fixed8"), (EX, "exact")):
        for mds in (1, 2):
            hp = oracle.HostPyramid(rgba[0], mds, mds * 64, mode)
            for l in range(4):
                _eq(hp.level(l), g["pyr_%s_ds%d_l%d" % (tag, mds, l)], "pyramid %s ds%d level %d" % (tag, mds, l))
            if mds == 1:
                out = np.empty((len(uvl), 4), np.float32)
                for i, (u, v, lod) in enumerate(uvl):
                    o4 = (C.c_float * 4)()
                    oracle.load().avo_tex2dlod(C.byref(hp.desc), float(u), float(v), float(lod), C.byref(o4))
                    out[i] = o4[:]
                _eq(out, g["tex_%s" % tag], "tex2DLod probes " + tag)


def _small_scene(g):
    class S:
        pass
    sc = S()
    sc.images = g["img8"].astype(np.float32) / np.float32(255.0)
    sc.K, sc.R, sc.C = g["K"], list(g["R"]), list(g["C"])
    return sc


def test_optional_kernels_vectors(g):
    """bilinear middle-depth upscale (kernel 16), the normal map (kernel 17 + eig33.cuh), the custom patch pattern (patchPattern.cpp +
    Patch.cuh:598-773) and useConsistentScale (Patch.cuh:250-308), against what the reference's code produced"""
    from oracle import oracle
    sc = _small_scene(g)
    depths = g["opt_depths"]
    sgm, rp = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=4, interpolateMiddleDepth=1)
    o = oracle.OracleDepthMap(sc.images, sc.K, sc.R, sc.C, sgm, rp, filter_mode=F8)
    o.run_sgm(0, [1, 2], depths)
    o.run_refine(0, [1, 2])
    # first row / column: the reference's kernel 16 reads SGM-map index -1 there (memory before the buffer; mapKernels.cuh:318-336), the
    # oracle and the library clamp to 0 (DESIGN "deliberate deviations") — compared from the second row / column on, and the optimised
    # map beyond the reach of those pixels after 4 iterations
    _eq(o.sgm_upscaled[2:, 2:], g["bilinear_upscaled"][2:, 2:], "bilinear upscale")
    _eq(o.optimized[8:, 8:], g["bilinear_optimized"][8:, 8:], "optimised (bilinear upscale)")
    H, W = o.optimized.shape[:2]
    nrm = np.zeros((H, W, 3), np.float32)
    rc1 = o.cam(0, 1)
    o.lib.avo_depth_sim_map_compute_normal(oracle.ptr(nrm), W * 12, oracle.ptr(o.optimized), W * 8, C.byref(rc1), 1, abi.ROI.make(0, W, 0, H))
    want = g["normal_map"]
    # interior only: the reference's neighbourhood reads beyond the tile on the upper sides (DESIGN "deliberate deviations")
    a, b = nrm[3:-3, 3:-3], want[3:-3, 3:-3]
    valid = np.isfinite(b).all(-1) & (np.abs(b).sum(-1) > 0)
    assert valid.mean() > 0.5
    # different eigen-solvers (closed form vs tred2 / tql2): same vector up to rounding where the plane is well defined
    dots = np.abs((a[valid] * b[valid]).sum(-1))
    assert np.median(dots) > 1 - 1e-6 and (dots < 1 - 1e-3).mean() < 0.02, (np.median(dots), (dots < 1 - 1e-3).mean())

    subs = (abi.PatchSubpartParams * 2)(abi.PatchSubpartParams(0, 0, 0, 2.0, 0.6), abi.PatchSubpartParams(1, 1, 8, 3.0, 0.4))
    pat = abi.PatchPattern()
    assert o.lib.avo_build_custom_patch_pattern(2, subs, 1, C.byref(pat)) == 0
    want_pat = abi.PatchPattern.from_buffer_copy(g["pattern_bytes"].tobytes())
    assert pat.nbSubparts == want_pat.nbSubparts == 2
    for i in range(pat.nbSubparts):
        a, b = pat.subparts[i], want_pat.subparts[i]
        # (coordinates of a full subpart and entries past nbCoordinates are uninitialised pinned memory in the reference)
        assert (a.nbCoordinates, a.level, a.downscale, a.weight, a.isCircle, a.wsh) == (b.nbCoordinates, b.level, b.downscale, b.weight, b.isCircle, b.wsh), i
        if a.isCircle:
            for c in range(a.nbCoordinates):
                assert tuple(a.coordinates[c]) == tuple(b.coordinates[c]), (i, c)
    sgm2, rp2 = abi.SgmParams.default(useCustomPatchPattern=1), abi.RefineParams.default(optimizationNbIterations=0, useCustomPatchPattern=1)
    o2 = oracle.OracleDepthMap(sc.images, sc.K, sc.R, sc.C, sgm2, rp2, filter_mode=F8)
    o2.run_sgm(0, [1, 2], depths)
    o2.run_refine(0, [1, 2])
    _eq(o2.second[..., :16], g["pattern_second"], "custom pattern: similarity volume")
    _eq(o2.refined, g["pattern_refined"], "custom pattern: refined map")
    sgm3, rp3 = abi.SgmParams.default(useConsistentScale=1), abi.RefineParams.default(optimizationNbIterations=0, useConsistentScale=1)
    o3 = oracle.OracleDepthMap(sc.images, sc.K, sc.R, sc.C, sgm3, rp3, filter_mode=F8)
    o3.run_sgm(0, [1, 2], depths)
    o3.run_refine(0, [1, 2])
    _eq(o3.second[..., :16], g["cs_second"], "consistent scale: similarity volume")
    _eq(o3.refined, g["cs_refined"], "consistent scale: refined map")


# ---- live comparisons ----------------------------------------------------------------------------------------------
def _ref():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref is neither prebuilt nor buildable here")
    ref.load()
    return ref


@pytest.mark.parametrize("W,H,NP,mode,roi,sgm_kw,ref_kw,tcr", [
    (160, 128, 24, EX, None, {}, {}, None),
    (250, 186, 20, F8, None, {}, {}, None),                                             # odd sizes: floor-halved levels vs ceil dims
    (256, 192, 16, F8, (64, 192, 32, 160), {}, {}, [(0, 16), (3, 13)]),                 # tile with offsets, per-T-camera plane ranges
    (256, 192, 16, F8, None, dict(stepXY=1, wsh=3, filteringAxes=b"XY", p2Weighting=30.0), dict(wsh=2, halfNbDepths=7, nbSubsamples=5, sigma=7.0), None),
    (200, 150, 12, F8, None, {}, dict(scale=2, stepXY=1), None),                        # min downscale 2: Gaussian-downscaled level 0
    (192, 144, 12, F8, (40, 168, 24, 120), dict(p2Weighting=-75.5, p1=7.0, depthThicknessInflate=0.3, maxSimilarity=0.8), {}, None),
])
def test_whole_tile_equals_reference(W, H, NP, mode, roi, sgm_kw, ref_kw, tcr):
    """every stage of a tile (pyramids, similarity volumes, SGM aggregation, WTA, smoothing, upscale, Refine volume, sub-sample arg-min,
    variance map, colour optimisation): oracle == the reference's kernels, bit for bit"""
    ref = _ref()
    sc, sgm, rp, depths = small_case(width=W, height=H, n_planes=NP, **sgm_kw)
    rp.optimizationNbIterations = 6
    for k, v in ref_kw.items():
        setattr(rp, k, v)
    o = make_oracle(sc, sgm, rp, filter_mode=mode, roi=roi)
    r = ref.RefDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, rp, filter_mode=mode, roi=roi)
    for l in range(min(o.pyr[0].desc.levels, 6)):
        _eq(o.pyr[0].level(l).astype(np.float32), r.img[0].level(l), "pyramid level %d" % l)
    o.run_sgm(0, [1, 2], depths, tc_ranges=tcr)
    r.run_sgm(0, [1, 2], depths, tc_ranges=tcr)
    Z = len(depths)
    _eq(o.best_raw[..., :Z], r.best_raw[..., :Z], "best similarity volume")
    _eq(o.second[..., :Z], r.second[..., :Z], "second-best similarity volume")
    _eq(o.filtered[..., :Z], r.filtered[..., :Z], "SGM-filtered volume")
    _eq(o.sgm_depth_thickness, r.sgm_depth_thickness, "SGM depth / thickness")
    _eq(o.sgm_depth_sim, r.sgm_depth_sim, "SGM depth / sim")
    wo, wr = o.run_refine(0, [1, 2]), r.run_refine(0, [1, 2])
    Zr = rp.halfNbDepths * 2 + 1
    _eq(o.sgm_depth_thickness_smooth, r.sgm_depth_thickness_smooth, "smoothed thickness")
    _eq(o.sgm_upscaled, r.sgm_upscaled, "upscaled depth / pixSize")
    _eq(o.refine_volume[..., :Zr].astype(np.float32), r.refine_volume.astype(np.float32), "Refine volume")
    _eq(o.refined, r.refined, "refined depth / sim")
    _eq(o.img_variance, r.img_variance, "image variance map")
    _eq(wo, wr, "optimised depth / sim")


@pytest.mark.parametrize("W,H,NP,mode,roi,sgm_kw,ref_kw,tcr,buf", [
    (250, 186, 20, F8, None, {}, {}, None, (1024, 1024)),                               # default buffer, image not divisible by the SGM downscale
    (250, 186, 20, F8, None, {}, {}, None, (250, 186)),                                 # buffer = image
    (256, 192, 16, F8, (64, 192, 32, 160), {}, {}, [(0, 16), (3, 13)], (160, 160)),     # a tile with offsets in a small tile buffer
    (256, 192, 16, F8, None, dict(stepXY=1, wsh=3, filteringAxes=b"XY", p2Weighting=30.0), dict(wsh=2, halfNbDepths=7, nbSubsamples=5, sigma=7.0), None,
     (256, 256)),
    (190, 142, 12, EX, (40, 167, 24, 119), dict(p2Weighting=-75.5, p1=7.0, depthThicknessInflate=0.3, maxSimilarity=0.8), {}, None, (1024, 1024)),
    (200, 152, 12, F8, None, dict(useConsistentScale=1), dict(useConsistentScale=1), None, (200, 152)),                # consistent scale
    (200, 152, 12, F8, None, {}, dict(interpolateMiddleDepth=1), None, (200, 152)),                                      # bilinear upscale (kernel 16)
])
def test_tile_control_flow_equals_reference_host_classes(W, H, NP, mode, roi, sgm_kw, ref_kw, tcr, buf):
    """the per-tile CONTROL FLOW pinned to the reference's own host classes: depthMap/Sgm.cpp and depthMap/Refine.cpp compiled whole
    and unchanged (their buffer sizes — volumes of maxDepths planes, maps of the tile buffer —, the order of the wrapper calls, every
    argument: which camera block at which downscale, which ROI and plane range, the allocated depth as volDimZ) against
    OracleDepthMap.run_sgm / run_refine, which the C++ host's Sgm / Refine classes and the Python harness mirror.  Maps must be
    identical, bit for bit, wherever the tile has data."""
    ref = _ref()
    sc, sgm, rp, depths = small_case(width=W, height=H, n_planes=NP, **sgm_kw)
    rp.optimizationNbIterations = 6
    for k, v in ref_kw.items():
        setattr(rp, k, v)
    Z = len(depths)
    limits = [(a, b - a) for a, b in tcr] if tcr else [(0, Z), (0, Z)]
    for max_depths in (Z, Z + 9):  # the reference sizes its volumes by maxDepths, not by the tile's planes
        o = make_oracle(sc, sgm, rp, filter_mode=mode, roi=roi)
        r = ref.RefTile(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, rp, filter_mode=mode, roi=roi)
        o.run_sgm(0, [1, 2], depths, tc_ranges=tcr, tile_buffer=buf)
        wo = o.run_refine(0, [1, 2], tile_buffer=buf)
        wr = r.run_tile(0, [1, 2], depths, limits, tile_buffer=buf, max_depths=max_depths)
        _eq(o.sgm_depth_sim, r.sgm_depth_sim, "SGM depth / sim (maxDepths %d)" % max_depths)
        _eq(o.sgm_depth_thickness, r.sgm_depth_thickness, "SGM depth / thickness (maxDepths %d)" % max_depths)
        _eq(o.sgm_depth_thickness_smooth, r.sgm_depth_thickness_smooth, "smoothed thickness")
        x0, x1, y0, y1 = roi if roi is not None else (0, W, 0, H)
        if ref_kw.get("interpolateMiddleDepth"):
            # the reference's bilinear upscale reads index -1 on the first row / column of the SGM map (out of bounds: whatever lies
            # before the buffer; the product clamps it to 0) — the frame those values reach is undefined
            b = 4 * (rp.optimizationNbIterations + 2)
            _eq(wo[b:-b, b:-b], wr[b:-b, b:-b], "optimised depth / sim (interior)")
        elif (x1 - x0, y1 - y0) == tuple(buf):
            _eq(wo, wr, "optimised depth / sim")
        else:
            # a buffer larger than the tile: the reference's colour optimisation binds the WHOLE allocated temporary depth map as a texture
            # and its border pixels read texels beyond the tile that no kernel wrote (whatever the allocation held); every
            # iteration carries that one pixel further in — a frame of (iterations + 1) pixels is undefined there, everything inside must agree
            b = rp.optimizationNbIterations + 1
            _eq(wo[b:-b, b:-b], wr[b:-b, b:-b], "optimised depth / sim (interior)")


@pytest.mark.parametrize("fuse,opt", [(False, True), (True, False), (False, False)])
def test_tile_control_flow_switches_equal_reference_host_classes(fuse, opt):
    """Refine::refineRc's other branches in the reference's own Refine.cpp: useRefineFuse off (cuda_depthSimMapCopyDepthOnly instead of the
    Refine volume, Refine.cpp:141-150) and useColorOptimization off (the refined map copied through, :163-170) — the switches behind
    --refineEnabled / --colorOptimizationEnabled — and the SGM normal map of Sgm::sgmRc (Sgm.cpp:167-184)"""
    ref = _ref()
    sc, sgm, rp, depths = small_case(width=200, height=152, n_planes=14)
    rp.optimizationNbIterations = 5
    Z = len(depths)
    o = make_oracle(sc, sgm, rp, filter_mode=F8)
    r = ref.RefTile(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, rp, filter_mode=F8)
    o.run_sgm(0, [1, 2], depths, tile_buffer=(200, 152))
    wo = o.run_refine(0, [1, 2], refine_enabled=fuse, optimize_enabled=opt, tile_buffer=(200, 152))
    wr = r.run_tile(0, [1, 2], depths, [(0, Z), (0, Z)], tile_buffer=(200, 152), max_depths=Z, compute_normal=True, use_refine_fuse=fuse,
                    use_color_optimization=opt)
    _eq(o.sgm_depth_sim, r.sgm_depth_sim, "SGM depth / sim")
    _eq(wo, wr, "final depth / sim")
    # the SGM normal map (cuda_depthSimMapComputeNormal on the SGM depth / sim map, R camera at the SGM scale, Sgm.cpp:170-176).  With a
    # buffer the size of the tile the reference's neighbourhood cannot leave the tile either (deliberate deviations).
    from oracle import oracle
    roiS = o.droi(sgm.scale * sgm.stepXY)
    nrm = np.zeros((roiS.height, roiS.width, 3), np.float32)
    cam = o.cam(0, sgm.scale)
    o.lib.avo_depth_sim_map_compute_normal(oracle.ptr(nrm), roiS.width * 12, oracle.ptr(np.ascontiguousarray(o.sgm_depth_sim)), roiS.width * 8, C.byref(cam),
                                           sgm.stepXY, roiS)
    _eq(nrm, r.sgm_normal, "SGM normal map")


def test_sgm_aggregation_equals_reference_at_scale():
    """cuda_volumeOptimize (deviceSimilarityVolume.cu:262-425, ~3 kernel launches per slice) on a 120 x 90 x 64 volume with adaptive P2
    and ROI offsets, both axis orders: the oracle's aggregate_path loop == the reference's wrapper + kernels.  The reference's P2 uses
    libm's expf, the oracle the specified polynomial: bytes may differ only where floor(P2) does (counted, must be rare)."""
    ref = _ref()
    from oracle import oracle
    sc, sgm, rp, _ = small_case(width=640, height=480, n_views=1, seed=5)
    rng = np.random.RandomState(4)
    X, Y, Z = 120, 90, 64
    for axes, x0, y0, p2w in ((b"YX", 0, 0, 100.0), (b"XY", 21, 9, 20.0)):
        sgm = abi.SgmParams.default(filteringAxes=axes, p2Weighting=p2w)
        o = make_oracle(sc, sgm, rp)
        img = ref.RefImage(sc.images.numpy()[0], 1, 128)
        ref.load().avr_set_filter_mode(F8)
        yy, xx, zz = np.meshgrid(np.arange(Y), np.arange(X), np.arange(Z), indexing="ij")
        valley = Z * (0.5 + 0.3 * np.sin(xx / 17.0) * np.cos(yy / 11.0))
        vin = np.minimum(np.abs(zz - valley) * 1.2 + rng.randint(0, 25, size=(Y, X, Z)), 254).astype(np.uint8)
        vin[rng.rand(Y, X, Z) < 0.02] = 255
        roi = abi.ROI.make(x0, x0 + X, y0, y0 + Y)
        want = np.full_like(vin, 9)
        ref.load().avr_volume_optimize(ref.ptr(want), ref.ptr(vin), X * Z, Z, X, Y, Z, img.h, C.byref(sgm), Z, roi)
        got = np.full_like(vin, 9)
        oracle.load().avo_volume_optimize(oracle.ptr(got), oracle.ptr(vin), X * Z, Z, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
        diff = (got != want)
        assert diff.mean() < 2e-3, diff.mean()
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1


def test_camera_block_against_reference_projection():
    """avo_camera_fill (restating fillHostCameraParameters, DeviceCache.cpp:41-134, which needs MultiViewParams and is not part of
    oracle/_ref): its P / iP / C are consistent with the reference's project3DPoint and get3DPointForPixelAndDepthFromRC
    through the whole-tile equalities above; here: P * X == K [R | -RC] X in double precision to fp32 accuracy"""
    from oracle import oracle
    sc, _, _, _ = small_case(width=160, height=128)
    for ds in (1, 2, 4):
        cam = oracle.camera_fill(sc.K, sc.R[1], sc.C[1], ds)
        Kd = np.diag([1.0 / ds, 1.0 / ds, 1.0]) @ np.asarray(sc.K)
        Pd = Kd @ np.hstack([np.asarray(sc.R[1]), (-np.asarray(sc.R[1]) @ np.asarray(sc.C[1]))[:, None]])
        P = np.array(cam.P[:], np.float64).reshape(4, 3).T  # column-major 3 x 4
        assert np.abs(P - Pd).max() <= 4e-7 * np.abs(Pd).max()
