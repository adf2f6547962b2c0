"""Measured parity table: everything on the GPU (C ABI) against everything in the CPU oracle, per stage and end to end, in the
oracle's LITERAL mode (fp32 NCC sums, re-projected R pixel in the border test: the reference's arithmetic as written) and in its
well-posed mode (double-precision sums, exact pixel), with NO trimming of the depth error.

    python scripts/parity_report.py [--cases cfg1,crop2,crop3] [--out profiles/r02_parity_table.json]

Run on the GPU box (gpurun); the numbers go into DESIGN.md §2 and are asserted by tests/test_gpu_parity.py::test_parity_table_*.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from alicevision_amd import abi  # noqa: E402
from alicevision_amd.synthetic import make_scene, plane_depths  # noqa: E402

CASES = {
    # name: (n_views, W, H, n_planes, seed, roi or None, sgm kw) — cfg1 is SURVEY 8d.1 (single tile => stepXY 1)
    "cfg1": dict(n_views=3, W=640, H=480, Z=64, seed=1, roi=None, sgm=dict(stepXY=1)),
    # 512 x 512 crops of the cfg2 / cfg3 geometry: same cameras / image size / plane count, 4 T cameras
    "crop2": dict(n_views=5, W=1920, H=1080, Z=128, seed=2, roi=(704, 1216, 284, 796), sgm={}),
    "crop3": dict(n_views=5, W=4000, H=3000, Z=256, seed=3, roi=(1744, 2256, 1244, 1756), sgm={}),
    # cfg3's REAL shape (round 4): the image corners (border rejection Patch.cuh:486-496, clamp addressing) and all 10 T cameras of the
    # 11-view scene bench.py runs (the outer rings: baselines 2 x and 3 x the inner ring's)
    "crop3_corner": dict(n_views=5, W=4000, H=3000, Z=256, seed=3, roi=(0, 512, 0, 512), sgm={}),
    "crop3_far_corner": dict(n_views=5, W=4000, H=3000, Z=256, seed=3, roi=(3488, 4000, 2488, 3000), sgm={}),
    "crop3_10T": dict(n_views=11, W=4000, H=3000, Z=256, seed=3, roi=(1744, 2256, 1244, 1756), sgm={}),
    # two tiles of the DEFAULT tiling of a 12 MP image (mvsUtils::getTileRoiList: buffer 1024, padding 64 -> 5 x 4 tiles of 864 x 816), laid out
    # and aggregated over the tile BUFFER like the reference (OracleDepthMap(tile_buffer=...), pinned to Sgm.cpp / Refine.cpp): tile (2, 1) in
    # the interior and tile (4, 3) at the far image corner (800 x 744: clipped); 2 T cameras (the oracle's time goes with pixels x T cameras)
    "tile12mp_interior": dict(n_views=3, W=4000, H=3000, Z=256, seed=3, roi=(1600, 2464, 752, 1568), sgm={}, tile_buffer=(1024, 1024)),
    "tile12mp_corner": dict(n_views=3, W=4000, H=3000, Z=256, seed=3, roi=(3200, 4000, 2256, 3000), sgm={}, tile_buffer=(1024, 1024)),
    # the same clipped corner tile with all TEN T cameras of the bench's 11-view scene: best / second-best merging over 10 T cameras at a clipped
    # tile is the bench's real shape (deviceSimilarityVolumeKernels.cuh:221-232)
    "tile12mp_corner_10T": dict(n_views=11, W=4000, H=3000, Z=256, seed=3, roi=(3200, 4000, 2256, 3000), sgm={}, tile_buffer=(1024, 1024)),
    # BASELINE configuration 5 at its OWN shape (round 5): 24 MP frame (6000 x 4000), `--tileBufferWidth 1664 --tileBufferHeight 1152 --tilePadding 64`
    # -> 4 x 4 tiles of 1564 x 1064 (TileParams.cpp:15-61; tests/test_host_ref.py pins the grid to the reference's own getTileRoiList): tile (1, 1) in
    # the interior and tile (3, 3) at the far image corner (1500 x 1000: clipped), volumes laid out and aggregated over the NON-SQUARE tile buffer
    # (416 x 288 SGM columns / rows, deviceSimilarityVolume.cu:278-283), 256 planes, 2 T cameras
    "tile24mp_interior": dict(n_views=3, W=6000, H=4000, Z=256, seed=5, roi=(1500, 3064, 1000, 2064), sgm={}, tile_buffer=(1664, 1152)),
    "tile24mp_corner": dict(n_views=3, W=6000, H=4000, Z=256, seed=5, roi=(4500, 6000, 3000, 4000), sgm={}, tile_buffer=(1664, 1152)),
}


def level_hist(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    n = d.size
    return {"0": float((d == 0).sum() / n), "1": float((d == 1).sum() / n), "2": float((d == 2).sum() / n), "3+": float((d >= 3).sum() / n),
            "validity_differs": float(((a == 255) != (b == 255)).mean())}


def depth_stats(got, want, pix=None):
    vg, vw = got[..., 0] > 0, want[..., 0] > 0
    both = vg & vw
    err = (got[..., 0] - want[..., 0])[both].astype(np.float64)
    s = np.sort(err ** 2)
    out = {"valid_both": float(both.mean()), "validity_differs": float((vg != vw).mean()),
           "rmse_untrimmed": float(np.sqrt(s.mean())), "rmse_best_99.5pct": float(np.sqrt(s[: int(0.995 * s.size)].mean())),
           "rmse_best_99pct": float(np.sqrt(s[: int(0.99 * s.size)].mean())),
           "median_abs": float(np.median(np.abs(err))), "p99_abs": float(np.percentile(np.abs(err), 99)), "max_abs": float(np.abs(err).max()),
           "frac_abs_gt_1e-3": float((np.abs(err) > 1e-3).mean())}
    # scale-free forms: relative to the depth itself, and in units of the pixel size
    out["median_depth"] = float(np.median(want[..., 0][both]))
    out["rmse_untrimmed_relative"] = float(np.sqrt(((err / want[..., 0][both]) ** 2).mean()))
    if pix is not None:  # in units of the pixel size (depth step of one Refine plane): SURVEY 8c's "1e-3 * pixSize"
        e = err / pix[both]
        out["rmse_untrimmed_in_pixsize"] = float(np.sqrt((e ** 2).mean()))
        out["median_pixsize"] = float(np.median(pix[both]))
    return out


def sim_stats(got, want):
    """the similarity channel of the final map as the program writes it (mvsUtils/mapIO.cpp:403-540: <view>_simMap.exr is ONE HALF per pixel): share of
    identical halfs and the distance of the rest, over the pixels valid in both maps"""
    both = (got[..., 0] > 0) & (want[..., 0] > 0)
    g, w = got[..., 1].astype(np.float16)[both], want[..., 1].astype(np.float16)[both]
    d = np.abs(g.astype(np.float32) - w.astype(np.float32))
    return {"identical_halfs": float((g.view(np.uint16) == w.view(np.uint16)).mean()), "max_abs": float(d.max()), "p99_abs": float(np.percentile(d, 99)),
            "rmse": float(np.sqrt((d.astype(np.float64) ** 2).mean())), "frac_abs_gt_1e-2": float((d > 1e-2).mean())}


def gpu_run(pyr, sc, sgm, ref, roi, tcs, depths, tile_buffer=None):
    """one tile through the C ABI: every intermediate the table compares"""
    import torch
    from alicevision_amd.pipeline import DepthMapTile
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=roi, tile_buffer=tile_buffer)
    h.run_sgm(0, tcs, depths, keep_raw=True)
    Z = len(depths)
    Y, X = h.sgm_depth_sim.shape[:2]  # the tile's ROI in the corner of the (buffer-sized) volumes
    g = {"second": h.second.cpu().numpy()[:Y, :X, :Z], "filtered": h.best.cpu().numpy()[:Y, :X, :Z], "sgm": h.sgm_depth_sim.cpu().numpy().copy()}
    g["final"] = h.run_refine(0, tcs).cpu().numpy().copy()
    g["refvol"] = h.refine_volume.cpu().numpy()[..., : h.Zr].astype(np.float32)
    g["refined"] = h.refined.cpu().numpy().copy()
    g["Zr"] = h.Zr
    torch.cuda.synchronize()
    return g


# AVDM_SIM_LITERAL_DEV bits (csrc/avdm_literal.hip): the deviations of the default kernels, introduced into the literal evaluation one at a time
# ("exact_border_r3": the border test on the exact pixel — a deviation of rounds 1-3 that the default kernels no longer have)
DEVIATIONS = {"shifted_sums": 1, "merged_exp": 2, "homogeneous_v_rcp": 4, "exact_centre": 8, "shared_R": 16, "all": 31, "exact_border_r3": 32, "all_r3": 63}


def run_case(name, spec, filter_mode, with_ref=False, gpu_literal=False, spread=False, deviations=(), modes=("well_posed", "literal"), strict=()):
    """modes: the oracle evaluations to run, "literal" last (the literal run's volumes feed the later blocks); the GPU suite runs the 24 MP tile
    against the literal evaluation only (the oracle's time goes with the pixels)"""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    t0 = time.time()
    sc = make_scene(spec["n_views"], spec["W"], spec["H"], seed=spec["seed"], device="cuda")  # rendered on the GPU (12 MP views take a minute on the CPU)
    images_np = sc.images.cpu().numpy()
    sgm = abi.SgmParams.default(**spec["sgm"])
    ref = abi.RefineParams.default()
    depths = plane_depths(sc, spec["Z"])
    tcs = list(range(1, spec["n_views"]))
    roi = spec["roi"]
    res = {"case": name, "image": [spec["W"], spec["H"]], "planes": spec["Z"], "t_cams": len(tcs), "roi": roi,
           "filter": "FIXED8" if filter_mode == abi.FILTER_CUDA_FIXED8 else "EXACT"}

    pyr = [DevicePyramid(sc.images[i], 1, 128, filter_mode) for i in range(spec["n_views"])]
    Z = len(depths)
    tb = spec.get("tile_buffer")
    okw = {"tile_buffer": tb} if tb is not None else {}
    g = gpu_run(pyr, sc, sgm, ref, roi, tcs, depths, tb)
    g_second, g_filtered, g_sgm, g_final, g_refvol, g_refined, Zr = g["second"], g["filtered"], g["sgm"], g["final"], g["refvol"], g["refined"], g["Zr"]
    res["t_gpu_s"] = time.time() - t0

    o = oracle.OracleDepthMap(images_np, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi)
    wants = {}
    for mode in modes:
        t1 = time.time()
        if mode == "well_posed":
            with oracle.well_posed():
                o.run_sgm(0, tcs, depths, **okw)
                want = o.run_refine(0, tcs, **okw).copy()
        else:
            o.run_sgm(0, tcs, depths, **okw)
            want = o.run_refine(0, tcs, **okw).copy()
        r = {}
        r["similarity_volume_levels"] = level_hist(o.second[..., :Z], g_second)
        r["sgm_filtered_volume_levels"] = level_hist(o.filtered[..., :Z], g_filtered)
        r["sgm_wta_depth_differs"] = float((o.sgm_depth_sim[..., 0] != g_sgm[..., 0]).mean())
        d = np.abs(o.refine_volume[..., : Zr].astype(np.float32) - g_refvol)
        r["refine_volume_abs"] = {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())}
        pix = o.sgm_upscaled[..., 1]
        r["refined_depth"] = depth_stats(g_refined, o.refined, pix)
        r["refined_sim"] = sim_stats(g_refined, o.refined)  # the Refine stage's similarity, BEFORE the colour optimisation re-draws it (see final_sim)
        r["final_depth"] = depth_stats(g_final, want, pix)
        r["final_sim"] = sim_stats(g_final, want)
        gt = sc.gt_depth.cpu().numpy()
        if roi is not None:
            gt = gt[roi[2]:roi[3], roi[0]:roi[1]]
        both = (g_final[..., 0] > 0) & (want[..., 0] > 0)
        r["median_abs_vs_ground_truth"] = {"gpu": float(np.median(np.abs(g_final[..., 0] - gt)[both])),
                                           "oracle": float(np.median(np.abs(want[..., 0] - gt)[both]))}
        r["t_oracle_s"] = time.time() - t1
        res[mode] = r
        wants[mode] = (want, o.second[..., :Z].copy())
    # the Lab pyramids: the GPU's against the oracle's, texel for texel (round 6: glibc's cbrtf restated on the device, no contraction)
    pdiff = []
    for v in range(spec["n_views"]):
        for l in range(min(pyr[v].desc.levels, o.pyr[v].desc.levels)):
            a, b = o.pyr[v].level(l), pyr[v].level(l).cpu().numpy()
            pdiff.append(float((a.view(np.uint16) != b.view(np.uint16)).mean()))
    res["pyramid_texels_differing"] = {"max_over_levels": max(pdiff), "mean": float(np.mean(pdiff))}
    # THE PRODUCT'S REFERENCE-ARITHMETIC MODE (round 6; avdm_sgm_params_t / avdm_refine_params_t::referenceArithmetic, the CLI's
    # --sgmReferenceArithmetic / --refineReferenceArithmetic): everything on the GPU from the GPU's OWN pyramids against the literal oracle.
    # "sgm": the SGM sweep alone in the reference's arithmetic (the default Refine kernels); "all": both sweeps.
    for smode in strict:
        t1 = time.time()
        kw = dict(spec["sgm"])
        sgm_s = abi.SgmParams.default(referenceArithmetic=1, **kw)
        ref_s = abi.RefineParams.default(referenceArithmetic=1 if smode == "all" else 0)
        gs = gpu_run(pyr, sc, sgm_s, ref_s, roi, tcs, depths, tb)
        want_l, second_l = wants["literal"]
        d = np.abs(o.refine_volume[..., :Zr].astype(np.float32) - gs["refvol"])
        res["reference_arithmetic_" + smode + "_vs_oracle_literal"] = {
            "similarity_volume_levels": level_hist(second_l, gs["second"]),
            "sgm_filtered_volume_levels": level_hist(o.filtered[..., :Z], gs["filtered"]),
            "sgm_wta_depth_differs": float((o.sgm_depth_sim[..., 0] != gs["sgm"][..., 0]).mean()),
            "refine_volume_abs": {"identical": float((d == 0).mean()), ">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())},
            "refined_depth": depth_stats(gs["refined"], o.refined, o.sgm_upscaled[..., 1]), "refined_sim": sim_stats(gs["refined"], o.refined),
            "final_depth": depth_stats(gs["final"], want_l, o.sgm_upscaled[..., 1]), "final_sim": sim_stats(gs["final"], want_l), "t_s": time.time() - t1}
    # AVDM_SIM_LITERAL=1: the reference's similarity arithmetic as written, ON THE GPU (csrc/avdm_literal.hip), against the oracle's literal
    # mode on the oracle's own pyramids — what is left when the conditioning of the NCC sums is taken out of the comparison
    if gpu_literal:
        t1 = time.time()  # (the oracle object still holds the literal run: "literal" is the last mode of the loop above)
        opyr = [DevicePyramid.from_host_bytes(p.desc, p.buf) for p in o.pyr]
        os.environ["AVDM_SIM_LITERAL"] = "1"
        try:
            gl = gpu_run(opyr, sc, sgm, ref, roi, tcs, depths, tb)
        finally:
            os.environ.pop("AVDM_SIM_LITERAL", None)
        d = np.abs(o.refine_volume[..., :Zr].astype(np.float32) - gl["refvol"])
        res["gpu_literal_vs_oracle_literal"] = {
            "similarity_volume_levels": level_hist(o.second[..., :Z], gl["second"]),
            "sgm_filtered_volume_levels": level_hist(o.filtered[..., :Z], gl["filtered"]),
            "sgm_wta_depth_differs": float((o.sgm_depth_sim[..., 0] != gl["sgm"][..., 0]).mean()),
            "refine_volume_abs": {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())},
            "final_depth": depth_stats(gl["final"], want, o.sgm_upscaled[..., 1]), "t_s": time.time() - t1}
    from oracle import ref as refmod
    pix = o.sgm_upscaled[..., 1]
    # the literal evaluation on the GPU with ONE deviation of the default kernels introduced at a time (AVDM_SIM_LITERAL_DEV): which of them
    # carries the distance between the default kernels and the reference's arithmetic
    if deviations:
        opyr = [DevicePyramid.from_host_bytes(p.desc, p.buf) for p in o.pyr]
        res["literal_plus_deviation"] = {}
        for dname in deviations:
            os.environ["AVDM_SIM_LITERAL"] = "1"
            os.environ["AVDM_SIM_LITERAL_DEV"] = str(DEVIATIONS[dname])
            try:
                gd = gpu_run(opyr, sc, sgm, ref, roi, tcs, depths, tb)
            finally:
                os.environ.pop("AVDM_SIM_LITERAL", None)
                os.environ.pop("AVDM_SIM_LITERAL_DEV", None)
            res["literal_plus_deviation"][dname] = {
                "vs_literal": {"final_depth": depth_stats(gd["final"], wants["literal"][0], pix), "similarity_volume_levels": level_hist(wants["literal"][1], gd["second"])},
                "vs_well_posed": {"final_depth": depth_stats(gd["final"], wants["well_posed"][0], pix),
                                  "similarity_volume_levels": level_hist(wants["well_posed"][1], gd["second"])}}
    # THE REFERENCE'S OWN PLATFORM SPREAD: the same reference sources evaluated the way an nvcc build evaluates them as far as this container
    # can tell (oracle/_ref/libavdm_ref_cuda.so: FMA contraction + the documented error model of the fast intrinsics) against their evaluation
    # with every fp32 operation as written (= the literal oracle) — the yardstick for the default kernels' distance to either
    if spread and refmod.available("cuda"):
        t1 = time.time()
        if tb is not None:
            # a tile of the tile workflow: the reference's own host classes (Sgm.cpp / Refine.cpp compiled whole) over the CUDA-like kernels — maps
            # only (the classes keep their volumes); the yardstick of the tile cases (tile12mp_corner)
            # Both evaluations go through RefTile: where the tile is smaller than its buffer the reference's colour optimisation reads texels
            # beyond the tile that no kernel wrote (a frame of iterations + 1 pixels) — the same "whatever" in both of them.  Against the
            # product and the oracle (which clamp at the tile) the comparison is on the INTERIOR, without that frame.
            rt = refmod.RefTile(images_np, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi, variant="cuda")
            want_cuda = rt.run_tile(0, tcs, depths, [(0, Z)] * len(tcs), tile_buffer=tb, max_depths=Z).copy()
            fb = ref.optimizationNbIterations + 1
            inner = lambda a: a[fb:-fb, fb:-fb]
            # (the reference's literal evaluation of the tile = the literal oracle, bit for bit on the interior: tests/test_oracle_ref.py::
            # test_tile_control_flow_equals_reference_host_classes; scripts/platform_spread.py runs both sides through RefTile on the CPU)
            res["platform_spread"] = {
                "interior_frame": fb,
                "cuda_vs_literal_interior": {"final_depth": depth_stats(inner(want_cuda), inner(wants["literal"][0]), inner(pix))},
                "well_posed_vs_literal_interior": {"final_depth": depth_stats(inner(wants["well_posed"][0]), inner(wants["literal"][0]), inner(pix))} if "well_posed" in wants else None,
                "default_vs_literal_interior": {"final_depth": depth_stats(inner(g_final), inner(wants["literal"][0]), inner(pix))},
                "default_vs_cuda_interior": {"final_depth": depth_stats(inner(g_final), inner(want_cuda), inner(pix))},
                "t_s": time.time() - t1}
        rc_ = None if tb is not None else refmod.RefDepthMap(images_np, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi, variant="cuda")
        if tb is None:
            rc_.run_sgm(0, tcs, depths)
            want_cuda = rc_.run_refine(0, tcs).copy()
            res["platform_spread"] = {
                "cuda_vs_literal": {"final_depth": depth_stats(want_cuda, wants["literal"][0], pix), "similarity_volume_levels": level_hist(wants["literal"][1], rc_.second[..., :Z]),
                                    # the similarity channel of the reference against itself: after the colour optimisation it is an energy term that ANY
                                    # perturbation of the depths re-draws (cfg1: 1.5 % identical halfs, p99 |d| 7.0) — the yardstick of final_sim
                                    "final_sim": sim_stats(want_cuda, wants["literal"][0]), "refined_sim": sim_stats(rc_.refined, o.refined)},
                "cuda_vs_well_posed": {"final_depth": depth_stats(want_cuda, wants["well_posed"][0], pix)},
                "well_posed_vs_literal": {"final_depth": depth_stats(wants["well_posed"][0], wants["literal"][0], pix),
                                          "similarity_volume_levels": level_hist(wants["literal"][1], wants["well_posed"][1])},
                "default_vs_cuda": {"final_depth": depth_stats(g_final, want_cuda, pix), "similarity_volume_levels": level_hist(rc_.second[..., :Z], g_second)},
                "t_s": time.time() - t1}
    # the REFERENCE'S OWN kernels (oracle/_ref, prebuilt library travelling with the snapshot): the literal oracle must equal them bit for
    # bit, which makes the "literal" block above GPU-vs-reference-code numbers
    if with_ref and refmod.available():
        t1 = time.time()
        r = refmod.RefDepthMap(images_np, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi)
        r.run_sgm(0, tcs, depths)
        want_ref = r.run_refine(0, tcs)
        o.run_sgm(0, tcs, depths)
        want_lit = o.run_refine(0, tcs)
        res["reference_code"] = {
            "oracle_literal_equals_reference": bool(np.array_equal(o.second[..., :Z], r.second[..., :Z]) and np.array_equal(o.filtered[..., :Z], r.filtered[..., :Z])
                                                    and np.array_equal(o.refined, r.refined) and np.array_equal(want_lit, want_ref)),
            "final_depth_gpu_vs_reference": depth_stats(g_final, want_ref, r.sgm_upscaled[..., 1]),
            "similarity_volume_levels_gpu_vs_reference": level_hist(r.second[..., :Z], g_second),
            "t_reference_s": time.time() - t1}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="cfg1,crop2,crop3")
    ap.add_argument("--filters", default="fixed8")
    ap.add_argument("--out", default=None)
    ap.add_argument("--ref-cases", default="cfg1", help="cases also run through oracle/_ref (the reference's own kernels on the CPU)")
    ap.add_argument("--literal-cases", default="cfg1,crop2,crop3", help="cases also run with AVDM_SIM_LITERAL=1 on the GPU")
    ap.add_argument("--spread-cases", default="", help="cases also run through oracle/_ref's CUDA-like evaluation (the reference's platform spread)")
    ap.add_argument("--modes", default="well_posed,literal", help="oracle evaluations to compare with (literal last)")
    ap.add_argument("--strict", default="", help="reference-arithmetic modes of the product to run against the literal oracle: sgm, all (comma separated)")
    a = ap.parse_args()
    out = []
    for name in a.cases.split(","):
        for f in a.filters.split(","):
            mode = abi.FILTER_CUDA_FIXED8 if f == "fixed8" else abi.FILTER_EXACT
            r = run_case(name, CASES[name], mode, with_ref=name in a.ref_cases.split(","), gpu_literal=name in a.literal_cases.split(","),
                         spread=name in a.spread_cases.split(","), modes=tuple(a.modes.split(",")),
                         strict=tuple(m for m in a.strict.split(",") if m))
            out.append(r)
            print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
