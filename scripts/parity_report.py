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
    if pix is not None:  # in units of the pixel size (depth step of one Refine plane): SURVEY 8c's "1e-3 * pixSize"
        e = err / pix[both]
        out["rmse_untrimmed_in_pixsize"] = float(np.sqrt((e ** 2).mean()))
    return out


def gpu_run(pyr, sc, sgm, ref, roi, tcs, depths):
    """one tile through the C ABI: every intermediate the table compares"""
    import torch
    from alicevision_amd.pipeline import DepthMapTile
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=roi)
    h.run_sgm(0, tcs, depths, keep_raw=True)
    Z = len(depths)
    g = {"second": h.second.cpu().numpy()[..., :Z], "filtered": h.best.cpu().numpy()[..., :Z], "sgm": h.sgm_depth_sim.cpu().numpy().copy()}
    g["final"] = h.run_refine(0, tcs).cpu().numpy().copy()
    g["refvol"] = h.refine_volume.cpu().numpy()[..., : h.Zr].astype(np.float32)
    g["refined"] = h.refined.cpu().numpy().copy()
    g["Zr"] = h.Zr
    torch.cuda.synchronize()
    return g


def run_case(name, spec, filter_mode, with_ref=False, gpu_literal=False):
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
    g = gpu_run(pyr, sc, sgm, ref, roi, tcs, depths)
    g_second, g_filtered, g_sgm, g_final, g_refvol, g_refined, Zr = g["second"], g["filtered"], g["sgm"], g["final"], g["refvol"], g["refined"], g["Zr"]
    res["t_gpu_s"] = time.time() - t0

    o = oracle.OracleDepthMap(images_np, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi)
    for mode in ("well_posed", "literal"):
        t1 = time.time()
        if mode == "well_posed":
            with oracle.well_posed():
                o.run_sgm(0, tcs, depths)
                want = o.run_refine(0, tcs).copy()
        else:
            o.run_sgm(0, tcs, depths)
            want = o.run_refine(0, tcs).copy()
        r = {}
        r["similarity_volume_levels"] = level_hist(o.second[..., :Z], g_second)
        r["sgm_filtered_volume_levels"] = level_hist(o.filtered[..., :Z], g_filtered)
        r["sgm_wta_depth_differs"] = float((o.sgm_depth_sim[..., 0] != g_sgm[..., 0]).mean())
        d = np.abs(o.refine_volume[..., : Zr].astype(np.float32) - g_refvol)
        r["refine_volume_abs"] = {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())}
        pix = o.sgm_upscaled[..., 1]
        r["refined_depth"] = depth_stats(g_refined, o.refined, pix)
        r["final_depth"] = depth_stats(g_final, want, pix)
        gt = sc.gt_depth.cpu().numpy()
        if roi is not None:
            gt = gt[roi[2]:roi[3], roi[0]:roi[1]]
        both = (g_final[..., 0] > 0) & (want[..., 0] > 0)
        r["median_abs_vs_ground_truth"] = {"gpu": float(np.median(np.abs(g_final[..., 0] - gt)[both])),
                                           "oracle": float(np.median(np.abs(want[..., 0] - gt)[both]))}
        r["t_oracle_s"] = time.time() - t1
        res[mode] = r
    # AVDM_SIM_LITERAL=1: the reference's similarity arithmetic as written, ON THE GPU (csrc/avdm_literal.hip), against the oracle's literal
    # mode on the oracle's own pyramids — what is left when the conditioning of the NCC sums is taken out of the comparison
    if gpu_literal:
        t1 = time.time()  # (the oracle object still holds the literal run: "literal" is the last mode of the loop above)
        opyr = [DevicePyramid.from_host_bytes(p.desc, p.buf) for p in o.pyr]
        os.environ["AVDM_SIM_LITERAL"] = "1"
        try:
            gl = gpu_run(opyr, sc, sgm, ref, roi, tcs, depths)
        finally:
            os.environ.pop("AVDM_SIM_LITERAL", None)
        d = np.abs(o.refine_volume[..., :Zr].astype(np.float32) - gl["refvol"])
        res["gpu_literal_vs_oracle_literal"] = {
            "similarity_volume_levels": level_hist(o.second[..., :Z], gl["second"]),
            "sgm_filtered_volume_levels": level_hist(o.filtered[..., :Z], gl["filtered"]),
            "sgm_wta_depth_differs": float((o.sgm_depth_sim[..., 0] != gl["sgm"][..., 0]).mean()),
            "refine_volume_abs": {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())},
            "final_depth": depth_stats(gl["final"], want, o.sgm_upscaled[..., 1]), "t_s": time.time() - t1}
    # the REFERENCE'S OWN kernels (oracle/_ref, prebuilt library travelling with the snapshot): the literal oracle must equal them bit for
    # bit, which makes the "literal" block above GPU-vs-reference-code numbers
    from oracle import ref as refmod
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
    a = ap.parse_args()
    out = []
    for name in a.cases.split(","):
        for f in a.filters.split(","):
            mode = abi.FILTER_CUDA_FIXED8 if f == "fixed8" else abi.FILTER_EXACT
            r = run_case(name, CASES[name], mode, with_ref=name in a.ref_cases.split(","), gpu_literal=name in a.literal_cases.split(","))
            out.append(r)
            print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
