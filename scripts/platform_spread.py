"""The reference's own platform spread: the SAME reference sources (oracle/_ref, compiled from /root/reference) evaluated several equally
faithful ways, and how far their depth maps are from each other.

    python scripts/platform_spread.py [--cases smoke,cfg1,crop2,crop3] [--variants fm,fma,cuda] [--out profiles/r04_platform_spread.json]

base  = every fp32 operation as written, the fast intrinsics as the exact operation (libavdm_ref.so: the library the literal oracle equals bit for bit)
fm    = the fast intrinsics with the error model the CUDA programming guide documents (__expf = ex2(x * log2e), __fdividef = x * rcp(y))
fma   = a * b + c contracted into one FMA wherever the compiler may (nvcc's default, -fmad=true)
cuda  = both
CPU only (no GPU, nothing of the product): test infrastructure.  BASELINE.json's bar is "depth RMSE vs the reference CUDA path < 1e-3"; the
CUDA path itself is not reproducible here, so the distance between two faithful evaluations of its source is the yardstick for any third one.
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
from scripts.parity_report import CASES, depth_stats, level_hist, sim_stats  # noqa: E402

ALL_CASES = dict(CASES)
# the scene of __graft_entry__.smoke() in rounds 1-3 (3 views 320 x 240, 48 planes, default parameters)
ALL_CASES["smoke"] = dict(n_views=3, W=320, H=240, Z=48, seed=11, roi=None, sgm={})


def run_variant(variant, images, sc, sgm, ref, roi, tcs, depths, filter_mode, tile_buffer=None):
    from oracle import ref as refmod
    t0 = time.time()
    if tile_buffer is not None:
        # a tile of the tile workflow: the reference's OWN host classes (Sgm.cpp / Refine.cpp compiled whole, oracle/ref/tile_driver.cpp) over the
        # variant's kernels — volumes and maps allocated for the tile buffer, aggregation over the buffer extent.  Maps only (the classes keep their volumes).
        r = refmod.RefTile(images, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi, variant=variant)
        Z = len(depths)
        final = r.run_tile(0, tcs, depths, [(0, Z)] * len(tcs), tile_buffer=tile_buffer, max_depths=Z).copy()
        return {"sgm": r.sgm_depth_sim.copy(), "final": final, "t_s": time.time() - t0}
    r = refmod.RefDepthMap(images, sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi, variant=variant)
    r.run_sgm(0, tcs, depths)
    final = r.run_refine(0, tcs).copy()
    Z = len(depths)
    return {"second": r.second[..., :Z].copy(), "filtered": r.filtered[..., :Z].copy(), "sgm": r.sgm_depth_sim.copy(), "refvol": r.refine_volume.astype(np.float32),
            "refined": r.refined.copy(), "final": final, "pix": r.sgm_upscaled[..., 1].copy(), "t_s": time.time() - t0}


def compare(a, b, pix=None):
    if "refvol" not in a:  # tile cases: maps only
        return {"sgm_wta_depth_differs": float((a["sgm"][..., 0] != b["sgm"][..., 0]).mean()), "final_depth": depth_stats(b["final"], a["final"], pix),
                "final_sim": sim_stats(b["final"], a["final"])}
    d = np.abs(a["refvol"] - b["refvol"])
    return {"similarity_volume_levels": level_hist(a["second"], b["second"]), "sgm_filtered_volume_levels": level_hist(a["filtered"], b["filtered"]),
            "sgm_wta_depth_differs": float((a["sgm"][..., 0] != b["sgm"][..., 0]).mean()),
            "refine_volume_abs": {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean()), "max": float(d.max())},
            "refined_depth": depth_stats(b["refined"], a["refined"], a["pix"]), "final_depth": depth_stats(b["final"], a["final"], a["pix"]),
            # the similarity channel as the program writes it (one half per pixel): the reference against itself (VERDICT r5 #5's yardstick)
            "refined_sim": sim_stats(b["refined"], a["refined"]), "final_sim": sim_stats(b["final"], a["final"])}


def run_case(name, variants, filter_mode=abi.FILTER_CUDA_FIXED8):
    spec = ALL_CASES[name]
    sc = make_scene(spec["n_views"], spec["W"], spec["H"], seed=spec["seed"])
    images = sc.images.numpy()
    sgm = abi.SgmParams.default(**spec["sgm"])
    ref = abi.RefineParams.default()
    depths = plane_depths(sc, spec["Z"])
    tcs = list(range(1, spec["n_views"]))
    res = {"case": name, "image": [spec["W"], spec["H"]], "planes": spec["Z"], "t_cams": len(tcs), "roi": spec["roi"]}
    tb = spec.get("tile_buffer")
    base = run_variant("", images, sc, sgm, ref, spec["roi"], tcs, depths, filter_mode, tb)
    res["t_base_s"] = base["t_s"]
    for v in variants:
        got = run_variant(v, images, sc, sgm, ref, spec["roi"], tcs, depths, filter_mode, tb)
        res["base_vs_" + v] = compare(base, got)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="smoke,cfg1")
    ap.add_argument("--variants", default="fm,fma,cuda")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = []
    for name in a.cases.split(","):
        r = run_case(name, a.variants.split(","))
        out.append(r)
        print(json.dumps(r), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
