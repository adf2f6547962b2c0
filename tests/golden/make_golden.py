"""Generates tests/golden/*.npz FROM THE REFERENCE'S OWN CODE:   python tests/golden/make_golden.py      (this container only)

The outputs are produced by oracle/_ref — the reference's kernel-launch layer (every `cuda_*` wrapper, kernel and device helper of
/root/reference/src/aliceVision/depthMap/cuda, DeviceMipmapImage, buildCustomPatchPattern) compiled unchanged for the CPU over the
stand-in CUDA runtime of oracle/ref/shim (see oracle/ref/ref_driver.cpp) — on seeded inputs; the fixtures travel to the GPU box where
/root/reference does not exist.  tests/test_oracle.py and tests/test_oracle_ref.py hold oracle/avdm_oracle.c to them bit for bit.

  relief_192x144_{fixed8,exact}.npz   one full tile (3 views, 24 planes) through SGM + Refine, both texture filter modes
  ref_helpers.npz                     device helpers called directly (rgb2xyz/xyz2lab, CostYKfromLab, simStat::computeWSim, sigmoid /
                                      sigmoid2, cuda_stat3d plane fit, project3DPoint), texture probes of a small pyramid, and the
                                      optional kernels (normal map, bilinear upscale, min-downscale-2 pyramid, custom patch pattern)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from alicevision_amd import abi  # noqa: E402
from alicevision_amd.synthetic import make_scene, plane_depths  # noqa: E402
from oracle import ref  # noqa: E402

CASES = {
    # name: (width, height, views, planes, seed, filter mode)
    "relief_192x144_fixed8": (192, 144, 3, 24, 3, abi.FILTER_CUDA_FIXED8),
    "relief_192x144_exact": (192, 144, 3, 24, 3, abi.FILTER_EXACT),
}


def run_case(w, h, n, z, seed, mode):
    sc = make_scene(n, w, h, seed=seed)
    sgm, rp = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=10)
    depths = plane_depths(sc, z)
    o = ref.RefDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, rp, filter_mode=mode)
    o.run_sgm(0, [1, 2], depths)
    out = o.run_refine(0, [1, 2])
    Z = len(depths)
    return {
        "depths": depths, "level1_L": o.img[0].level(1)[..., 0].astype(np.float16),
        "second": o.second[..., :Z].copy(), "filtered": o.filtered[..., :Z].copy(), "sgm_depth_thickness": o.sgm_depth_thickness.copy(),
        "refine_volume_s4": o.refine_volume.view(np.uint16)[::4, ::4].copy(), "refined": o.refined.copy(),
        "optimized": out.copy(), "gt_depth": sc.gt_depth.numpy(),
    }


def helper_inputs(seed=5):
    """seeded inputs of the helper-function vectors (stored in the file: the test does not regenerate them)"""
    rng = np.random.RandomState(seed)
    n = 512
    d = {}
    d["rgb01"] = np.concatenate([rng.rand(n - 6, 3), [[0, 0, 0], [1, 1, 1], [0.001, 0.002, 0.0005], [1, 0, 0], [0, 1, 0], [0, 0, 1]]]).astype(np.float32)
    d["yk_dxdy"] = rng.randint(-4, 5, size=(n, 2)).astype(np.int32)
    d["yk_c1c2"] = (rng.rand(n, 8) * np.array([255, 120, 120, 255] * 2) - np.array([0, 60, 60, 0] * 2)).astype(np.float32)
    m = 49
    g = rng.rand(64, m, 3).astype(np.float32)
    g[..., 0] = g[..., 0] * 200 + 20
    g[..., 1] = g[..., 0] * rng.uniform(0.5, 1.5, size=(64, 1)) + rng.randn(64, m) * rng.uniform(0.1, 30, size=(64, 1))
    g[..., 2] = np.exp(-3 * g[..., 2])
    g[0, :, 1] = 7.0      # zero variance -> non-finite -> 1
    g[1, :, 0] = 255.0    # idem on x
    d["wsim_samples"] = g.astype(np.float32)
    d["sig_z"] = np.concatenate([np.linspace(-3, 3, 200), rng.uniform(-50, 300, 56)]).astype(np.float32)
    pts = rng.randn(32, 49, 4).astype(np.float32)
    pts[..., 2] = 0.3 * pts[..., 0] - 0.2 * pts[..., 1] + 0.01 * pts[..., 2] + 5.0
    pts[..., 3] = rng.rand(32, 49) + 0.1
    d["plane_pts"] = pts
    d["proj_P"] = np.array([800, 0, 0, 0, 800, 0, 320, 240, 1, 10, -20, 3], np.float32)  # column-major 3x4
    d["proj_pts"] = (rng.randn(n, 3) * np.array([1, 1, 0.5]) + np.array([0, 0, 6])).astype(np.float32)
    return d


def run_helpers():
    lib = ref.load()
    d = helper_inputs()
    out = dict(d)
    p = ref.ptr
    n = len(d["rgb01"])
    lab = np.empty((n, 3), np.float32)
    lib.avr_rgb2lab(p(d["rgb01"]), n, p(lab))
    out["lab"] = lab
    for name, (gc, gp) in {"yk_sgm": (5.5, 8.0), "yk_refine": (15.5, 8.0)}.items():
        w = np.empty(len(d["yk_dxdy"]), np.float32)
        lib.avr_cost_yk_from_lab(p(d["yk_dxdy"]), p(d["yk_c1c2"]), len(w), np.float32(1.0) / np.float32(gc), np.float32(1.0) / np.float32(gp), p(w))
        out[name] = w
    g = d["wsim_samples"]
    ws = np.empty(g.shape[0], np.float32)
    lib.avr_sim_stat_wsim(p(g), g.shape[1], g.shape[0], p(ws))
    out["wsim"] = ws
    for name, args in {"sig_refine": (0.0, 1.0, 0.7, -0.7), "sig_p2": (80.0, 255.0, 80.0, 100.0), "sig_opt": (5.0, 30.0, 40.0, 20.0)}.items():
        a, b = np.empty(len(d["sig_z"]), np.float32), np.empty(len(d["sig_z"]), np.float32)
        lib.avr_sigmoid(p(d["sig_z"]), len(a), *[np.float32(v) for v in args], p(a), p(b))
        out[name] = np.stack([a, b])
    pts = d["plane_pts"]
    pl, ok = np.empty((pts.shape[0], 6), np.float32), np.empty(pts.shape[0], np.int32)
    lib.avr_stat3d_plane(p(pts), pts.shape[1], pts.shape[0], p(pl), p(ok))
    out["plane"], out["plane_ok"] = pl, ok
    pr = np.empty((len(d["proj_pts"]), 2), np.float32)
    lib.avr_project3d(p(d["proj_P"]), p(d["proj_pts"]), len(pr), p(pr))
    out["proj"] = pr

    # ---- texture unit + pyramids on a small 8-bit image (stored, so the test needs no renderer) ----
    sc = make_scene(3, 100, 76, seed=9)
    img8 = np.clip(np.round(sc.images.numpy() * 255.0), 0, 255).astype(np.uint8)
    out["img8"] = img8
    rgba = img8.astype(np.float32) / np.float32(255.0)
    rng = np.random.RandomState(3)
    uvl = np.stack([rng.uniform(-0.05, 1.05, 400), rng.uniform(-0.05, 1.05, 400), rng.uniform(-0.5, 7.0, 400)], 1).astype(np.float32)
    uvl[:200, 2] = np.floor(uvl[:200, 2].clip(0, 6))
    out["tex_uvl"] = uvl
    for mode, tag in ((abi.FILTER_CUDA_FIXED8, "fixed8"), (abi.FILTER_EXACT, "exact")):
        lib.avr_set_filter_mode(mode)
        for mds in (1, 2):
            im = ref.RefImage(rgba[0], mds, mds * 64)
            for l in range(4):
                out["pyr_%s_ds%d_l%d" % (tag, mds, l)] = im.level(l).astype(np.float16)
            if mds == 1:
                out["tex_%s" % tag] = im.tex2dlod(uvl)

    # ---- optional kernels on a small tile: SGM normals are not on the default path (useSgmNormalMap is a const false), the normal-map
    #      kernel, the bilinear middle-depth upscale and the custom patch pattern are ----
    sgm, rp = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=4, interpolateMiddleDepth=1)
    depths = plane_depths(sc, 16)
    out["opt_depths"] = depths
    o = ref.RefDepthMap(rgba, sc.K, sc.R, sc.C, sgm, rp, filter_mode=abi.FILTER_CUDA_FIXED8)
    o.run_sgm(0, [1, 2], depths)
    o.run_refine(0, [1, 2])
    out["bilinear_upscaled"] = o.sgm_upscaled.copy()
    out["bilinear_optimized"] = o.optimized.copy()
    H, W = o.optimized.shape[:2]
    nrm = np.zeros((H, W, 3), np.float32)
    lib.avr_depth_sim_map_compute_normal(p(nrm), W * 12, p(o.optimized), W * 8, W, H, o.slot(0, 1), 1, abi.ROI.make(0, W, 0, H))
    out["normal_map"] = nrm
    # custom patch pattern: one full level-0 subpart + one circle on level 1 (patchPattern.cpp), grouped per level
    subs = (abi.PatchSubpartParams * 2)(abi.PatchSubpartParams(0, 0, 0, 2.0, 0.6), abi.PatchSubpartParams(1, 1, 8, 3.0, 0.4))
    pat = abi.PatchPattern()
    assert lib.avr_build_custom_patch_pattern(2, subs, 1, C.byref(pat)) == 0  # grouped: the non-grouped form reads an uninitialised count (patchPattern.cpp:196-201)
    out["pattern_bytes"] = np.frombuffer(bytes(pat), np.uint8).copy()
    sgm2, rp2 = abi.SgmParams.default(useCustomPatchPattern=1), abi.RefineParams.default(optimizationNbIterations=0, useCustomPatchPattern=1)
    o2 = ref.RefDepthMap(rgba, sc.K, sc.R, sc.C, sgm2, rp2, filter_mode=abi.FILTER_CUDA_FIXED8)
    o2.run_sgm(0, [1, 2], depths)
    o2.run_refine(0, [1, 2])
    out["pattern_second"] = o2.second[..., :16].copy()
    out["pattern_refined"] = o2.refined.copy()
    sgm3, rp3 = abi.SgmParams.default(useConsistentScale=1), abi.RefineParams.default(optimizationNbIterations=0, useConsistentScale=1)
    o3 = ref.RefDepthMap(rgba, sc.K, sc.R, sc.C, sgm3, rp3, filter_mode=abi.FILTER_CUDA_FIXED8)
    o3.run_sgm(0, [1, 2], depths)
    o3.run_refine(0, [1, 2])
    out["cs_second"] = o3.second[..., :16].copy()
    out["cs_refined"] = o3.refined.copy()
    out["K"] = np.asarray(sc.K, np.float64)
    out["R"] = np.asarray(sc.R, np.float64)
    out["C"] = np.asarray(sc.C, np.float64)
    return out


if __name__ == "__main__":
    for name, cfg in CASES.items():
        r = run_case(*cfg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **r)
        print(name, {k: v.shape for k, v in r.items()})
    h = run_helpers()
    np.savez_compressed(os.path.join(HERE, "ref_helpers.npz"), **h)
    print("ref_helpers", {k: v.shape for k, v in h.items()}, os.path.getsize(os.path.join(HERE, "ref_helpers.npz")))
