"""Generates tests/golden/*.npz — run by hand when the oracle is changed ON PURPOSE:   python tests/golden/make_golden.py

The reference has no test, golden vector or fixture for depthMap and cannot be built here, so these files pin
the ORACLE against itself (regression) — "parity unpinned" in the sense of DESIGN.md.  What they add over a self-comparison:
the scene is analytic, so tests/test_oracle.py also checks the stored depth map against the known surface.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from alicevision_amd import abi  # noqa: E402
from alicevision_amd.synthetic import make_scene, plane_depths  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    # name: (width, height, views, planes, seed, filter mode)
    "relief_192x144_fixed8": (192, 144, 3, 24, 3, abi.FILTER_CUDA_FIXED8),
    "relief_192x144_exact": (192, 144, 3, 24, 3, abi.FILTER_EXACT),
}


def run_case(w, h, n, z, seed, mode):
    sc = make_scene(n, w, h, seed=seed)
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=10)
    depths = plane_depths(sc, z)
    o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref, filter_mode=mode)
    o.run_sgm(0, [1, 2], depths)
    out = o.run_refine(0, [1, 2])
    Z = len(depths)
    return {
        "depths": depths, "level1_L": o.pyr[0].level(1)[..., 0].copy(),
        "second": o.second[..., :Z].copy(), "filtered": o.filtered[..., :Z].copy(), "sgm_depth_thickness": o.sgm_depth_thickness.copy(),
        "refine_volume_s4": o.refine_volume.view(np.uint16)[::4, ::4].copy(), "refined": o.refined.copy(),
        "optimized": out.copy(), "gt_depth": sc.gt_depth.numpy(),
    }


if __name__ == "__main__":
    for name, cfg in CASES.items():
        r = run_case(*cfg)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **r)
        print(name, {k: v.shape for k, v in r.items()})
