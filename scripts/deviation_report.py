"""Attribution of the distance between the default similarity kernels and the reference's arithmetic, DEVIATION BY DEVIATION, next to the
reference's own platform spread.

    python scripts/deviation_report.py [--cases smoke,cfg1,crop2,crop3] [--out profiles/r04_deviation_table.json]

Every variant runs one tile through the C ABI on the GPU, on the ORACLE's pyramids (so that only the similarity arithmetic differs), and is
compared with
    lit   the literal oracle = the reference's own kernels compiled for the CPU (oracle/_ref/libavdm_ref.so), bit for bit
    cuda  the same reference sources evaluated the way an nvcc build evaluates them as far as this container can tell (FMA contraction + the
          documented error model of the fast intrinsics: oracle/_ref/libavdm_ref_cuda.so) — a second faithful evaluation of the reference
    wp    the oracle's well-posed mode (double-precision NCC sums): the value both approximate
Variants:
    default               the product kernels                         default_own_pyramids   the same on the GPU-built pyramids
    -shared_R             AVDM_SIM_PLANE_PAIRS=0: R side per plane    -dot2_taps             AVDM_SIM_PACKED=0: plain fp32 bilinear blend
    -shifted_sums / -merged_exp / -v_rcp                               variant builds of the FAST path with ONE deviation reverted
                                                                       (scripts/build_variant.sh dev_* -DAVDM_DEV_*=1), all_reverted = all of them
    literal               AVDM_SIM_LITERAL=1: the reference's arithmetic as written, on the GPU
    literal+<deviation>   AVDM_SIM_LITERAL_DEV=<bit>: the literal evaluation with ONE deviation of the default kernels introduced
Run on the GPU box (gpurun); the table goes into DESIGN.md section 2 and is asserted by tests/test_gpu_parity.py::test_deviation_attribution.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from alicevision_amd import abi  # noqa: E402
from alicevision_amd.synthetic import make_scene, plane_depths  # noqa: E402
from scripts.parity_report import depth_stats, level_hist  # noqa: E402
from scripts.platform_spread import ALL_CASES  # noqa: E402

AB = os.path.join(ROOT, "scripts", "ab")
# tag -> (library or None, environment)
VARIANTS = {
    "default_own_pyramids": (None, {}),
    "default": (None, {}),
    "-shared_R": (None, {"AVDM_SIM_PLANE_PAIRS": "0"}),
    "-dot2_taps": (None, {"AVDM_SIM_PACKED": "0"}),
    "-shifted_sums": ("dev_unshifted", {}),
    "-merged_exp": ("dev_twoexp", {}),
    "-v_rcp": ("dev_ieeediv", {}),
    "all_reverted": ("dev_all3", {"AVDM_SIM_PLANE_PAIRS": "0"}),
    "literal": (None, {"AVDM_SIM_LITERAL": "1"}),
    "literal+shifted_sums": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "1"}),
    "literal+merged_exp": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "2"}),
    "literal+homogeneous_v_rcp": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "4"}),
    "literal+exact_centre": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "8"}),
    "literal+shared_R": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "16"}),
    "literal+all_but_shifted_sums": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "30"}),
    "literal+all": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "31"}),
    # rounds 1-3: the border test on the exact pixel too (no longer a deviation of the default kernels)
    "literal+exact_border(r1-r3)": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "32"}),
    "literal+all(r1-r3)": (None, {"AVDM_SIM_LITERAL": "1", "AVDM_SIM_LITERAL_DEV": "63"}),
}
FIELDS = ("second", "filtered", "sgm", "refvol", "refined", "final")


def scene_of(name, device):
    spec = ALL_CASES[name]
    sc = make_scene(spec["n_views"], spec["W"], spec["H"], seed=spec["seed"], device=device)
    sgm = abi.SgmParams.default(**spec["sgm"])
    ref = abi.RefineParams.default()
    return spec, sc, sgm, ref, plane_depths(sc, spec["Z"]), list(range(1, spec["n_views"]))


def child(cases, tags, dump):
    """GPU runs of `tags` (all on the library this process loaded) -> <dump>/<case>__<tag>.npz"""
    import torch
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    for name in cases:
        spec, sc, sgm, ref, depths, tcs = scene_of(name, "cuda")
        images = sc.images.cpu().numpy()
        o = oracle.OracleDepthMap(images, sc.K, sc.R, sc.C, sgm, ref, filter_mode=abi.FILTER_CUDA_FIXED8, roi=spec["roi"])
        opyr = [DevicePyramid.from_host_bytes(p.desc, p.buf) for p in o.pyr]
        own = None
        for tag in tags:
            env = VARIANTS[tag][1]
            if tag == "default_own_pyramids":
                if own is None:
                    own = [DevicePyramid(sc.images[i], 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(spec["n_views"])]
                pyr = own
            else:
                pyr = opyr
            keep = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=spec["roi"])
                t0 = time.time()
                h.run_sgm(0, tcs, depths, keep_raw=True)
                Z = len(depths)
                g = {"second": h.second.cpu().numpy()[..., :Z], "filtered": h.best.cpu().numpy()[..., :Z], "sgm": h.sgm_depth_sim.cpu().numpy().copy()}
                g["final"] = h.run_refine(0, tcs).cpu().numpy().copy()
                g["refvol"] = h.refine_volume.cpu().numpy()[..., : h.Zr].astype(np.float32)
                g["refined"] = h.refined.cpu().numpy().copy()
                torch.cuda.synchronize()
                g["t_s"] = np.float64(time.time() - t0)
            finally:
                for k, v in keep.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            np.savez(os.path.join(dump, "%s__%s.npz" % (name, tag)), **g)
            print("child: %s %s %.2fs" % (name, tag, float(g["t_s"])), flush=True)


def references(name, with_spread):
    """the CPU side: literal oracle (= the reference's code), well-posed oracle, the reference evaluated the CUDA way"""
    from oracle import oracle
    from oracle import ref as refmod
    spec, sc, sgm, ref, depths, tcs = scene_of(name, "cpu" if os.environ.get("AVDM_REPORT_CPU_SCENE") else "cuda")
    images = sc.images.cpu().numpy()
    Z = len(depths)
    out, timing = {}, {}
    o = oracle.OracleDepthMap(images, sc.K, sc.R, sc.C, sgm, ref, filter_mode=abi.FILTER_CUDA_FIXED8, roi=spec["roi"])

    def grab(final):
        return {"second": o.second[..., :Z].copy(), "filtered": o.filtered[..., :Z].copy(), "sgm": o.sgm_depth_sim.copy(),
                "refvol": o.refine_volume[..., : 2 * ref.halfNbDepths + 1].astype(np.float32), "refined": o.refined.copy(), "final": final.copy(),
                "pix": o.sgm_upscaled[..., 1].copy()}
    t0 = time.time()
    with oracle.well_posed():
        o.run_sgm(0, tcs, depths)
        out["wp"] = grab(o.run_refine(0, tcs))
    timing["wp"] = time.time() - t0
    t0 = time.time()
    o.run_sgm(0, tcs, depths)
    out["lit"] = grab(o.run_refine(0, tcs))
    timing["lit"] = time.time() - t0
    if with_spread and refmod.available("cuda"):
        t0 = time.time()
        r = refmod.RefDepthMap(images, sc.K, sc.R, sc.C, sgm, ref, filter_mode=abi.FILTER_CUDA_FIXED8, roi=spec["roi"], variant="cuda")
        r.run_sgm(0, tcs, depths)
        fin = r.run_refine(0, tcs)
        out["cuda"] = {"second": r.second[..., :Z].copy(), "filtered": r.filtered[..., :Z].copy(), "sgm": r.sgm_depth_sim.copy(),
                       "refvol": r.refine_volume.astype(np.float32), "refined": r.refined.copy(), "final": fin.copy(), "pix": r.sgm_upscaled[..., 1].copy()}
        timing["cuda"] = time.time() - t0
    gt = sc.gt_depth.cpu().numpy()
    if spec["roi"] is not None:
        roi = spec["roi"]
        gt = gt[roi[2]:roi[3], roi[0]:roi[1]]
    return out, timing, gt


def compare(want, got):
    d = np.abs(want["refvol"] - got["refvol"])
    return {"similarity_volume_levels": level_hist(want["second"], got["second"]),
            "sgm_wta_depth_differs": float((want["sgm"][..., 0] != got["sgm"][..., 0]).mean()),
            "refine_volume_abs": {">2e-3": float((d > 2e-3).mean()), ">2e-2": float((d > 2e-2).mean())},
            "final_depth": depth_stats(got["final"], want["final"], want["pix"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="smoke,cfg1,crop2,crop3")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-spread", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--dump", default=None)
    a = ap.parse_args()
    cases, tags = a.cases.split(","), a.variants.split(",")
    if a.child:
        child(cases, tags, a.dump)
        return
    dump = a.dump or tempfile.mkdtemp(prefix="avdm_dev_")
    # one child process per library (the run-time switches are read at each call: all their variants share a process)
    groups = {}
    for t in tags:
        lib = VARIANTS[t][0]
        if lib is not None and not os.path.exists(os.path.join(AB, lib, "libavdm.so")):
            print("skipping %s: scripts/ab/%s/libavdm.so is not built" % (t, lib), flush=True)
            continue
        groups.setdefault(lib, []).append(t)
    procs = []
    for lib, ts in groups.items():
        env = dict(os.environ)
        if lib is not None:
            env["AVDM_LIB"] = os.path.join(AB, lib, "libavdm.so")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "--cases", a.cases, "--variants=" + ",".join(ts), "--dump", dump],
                                      env=env))
    out = []
    refs = {}
    for name in cases:  # the CPU side runs while the children use the GPU
        for p in procs:
            if p.poll() not in (None, 0):
                raise SystemExit("a child failed")
        refs[name] = references(name, not a.no_spread)
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("a child failed")
    for name in cases:
        want, timing, gt = refs[name]
        spec = ALL_CASES[name]
        res = {"case": name, "image": [spec["W"], spec["H"]], "planes": spec["Z"], "t_cams": spec["n_views"] - 1, "roi": spec["roi"], "t_cpu_s": timing,
               "references": {}, "variants": {}}
        for a_, b_ in (("lit", "wp"), ("lit", "cuda"), ("cuda", "wp")):
            if a_ in want and b_ in want:
                res["references"]["%s_vs_%s" % (b_, a_)] = compare(want[a_], want[b_])
        for t in [t for ts in groups.values() for t in ts]:
            f = os.path.join(dump, "%s__%s.npz" % (name, t))
            got = dict(np.load(f))
            r = {"t_gpu_s": float(got["t_s"])}
            for k in want:
                r["vs_" + k] = compare(want[k], got)
            both = got["final"][..., 0] > 0
            r["median_abs_vs_ground_truth"] = float(np.median(np.abs(got["final"][..., 0] - gt)[both]))
            res["variants"][t] = r
        for k in want:
            both = want[k]["final"][..., 0] > 0
            res["references"]["median_abs_vs_ground_truth_" + k] = float(np.median(np.abs(want[k]["final"][..., 0] - gt)[both]))
        out.append(res)
        # one line per variant: untrimmed final-depth RMSE against the three references + identical voxels of the similarity volume
        print("== %s" % name)
        for k, v in res["references"].items():
            if isinstance(v, dict):
                print("  %-32s rmse %.3e   volume identical %.4f" % (k, v["final_depth"]["rmse_untrimmed"], v["similarity_volume_levels"]["0"]))
        for t, r in res["variants"].items():
            print("  %-32s " % t + "  ".join("vs %s %.3e (vol %.4f)" % (k, r["vs_" + k]["final_depth"]["rmse_untrimmed"], r["vs_" + k]["similarity_volume_levels"]["0"])
                                             for k in want), flush=True)
        if a.out:
            with open(a.out, "w") as f:
                json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
