"""profiles/<round>_sim_pmc.json from the three counter passes of scripts/pmc_similarity.sh (gpurun_out/<tag>/simpmc_*.csv): per T-camera launch of
the two default similarity kernels — VALU wave-instructions, VALU-active / wave cycles, LDS bank-conflict cycles — stamped with the sha256 of
csrc/avdm_similarity.hip so that bench.py only quotes them while the kernel source is the one that was profiled.

    python scripts/collect_sim_pmc.py r04_e
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, out_name=None):
    src = os.path.join(ROOT, "gpurun_out", tag)
    per = {}
    files = sorted(glob.glob(os.path.join(src, "simpmc_*.csv")))
    for f in files:
        shutil.copy(f, os.path.join(ROOT, "profiles", "%s_%s" % (tag, os.path.basename(f))))
        for row in csv.DictReader(open(f)):
            k = "refine" if "refine_similarity_kernel" in row["kernel"] else ("sgm" if "similarity_kernel" in row["kernel"] else None)
            if k is None:
                continue
            d = per.setdefault(k, {"kernel": row["kernel"], "launches": int(float(row["launches"]))})
            for name, v in row.items():
                if name.endswith("_per_launch"):
                    d[name[: -len("_per_launch")]] = float(v)
            if row.get("us_per_launch") and "GRBM_GUI_ACTIVE_per_launch" in row:
                d["us_per_launch_under_pmc"] = float(row["us_per_launch"])  # of the pass that carries GRBM_GUI_ACTIVE
    for d in per.values():
        # a wave issues a VALU instruction in SQ_ACTIVE_INST_VALU of its SQ_WAVE_CYCLES resident cycles (both in quad-cycles); two waves share a SIMD
        d["valu_active_per_wave"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        d["wait_lds_frac_of_active"] = d["SQ_WAIT_INST_LDS"] / d["SQ_ACTIVE_INST_ANY"] if "SQ_WAIT_INST_LDS" in d and d.get("SQ_ACTIVE_INST_ANY") else None
        if d.get("GRBM_GUI_ACTIVE") and d.get("us_per_launch_under_pmc"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; over the launch's duration under the same pass = the clock the kernel really ran at
            clk = d["GRBM_GUI_ACTIVE"] / 8.0 / (d["us_per_launch_under_pmc"] * 1e-6)
            simd_cycles = 1024.0 * clk * d["us_per_launch_under_pmc"] * 1e-6
            d["clock_GHz"] = clk / 1e9
            d["valu_issue_frac_at_measured_clock"] = d["SQ_INSTS_VALU"] * 4.0 / simd_cycles         # one wave64 VALU instruction = 4 cycles of a SIMD
            d["valu_busy_frac_at_measured_clock"] = d["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles    # quad-cycles a SIMD's VALU was held (transcendentals: longer)
            if d.get("SQ_WAIT_ANY"):
                d["waves_parked_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]                  # s_waitcnt / barrier
            if d.get("SQ_WAIT_INST_ANY"):
                d["waves_issue_stalled_frac"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
            if d.get("SQ_LDS_IDX_ACTIVE"):
                d["lds_busy_frac"] = d["SQ_LDS_IDX_ACTIVE"] / (256.0 * clk * d["us_per_launch_under_pmc"] * 1e-6)
    sha = hashlib.sha256(open(os.path.join(ROOT, "alicevision_amd", "csrc", "avdm_similarity.hip"), "rb").read()).hexdigest()
    out = {"source": "profiles/%s_simpmc_*.csv (rocprofv3 --pmc, counters only: scripts/pmc_similarity.sh)" % tag, "per_kernel": per, "kernel_source_sha256": sha,
           "waves_per_simd": 2}
    rnd = tag.split("_")[0]
    json.dump(out, open(os.path.join(ROOT, "profiles", out_name or ("%s_sim_pmc.json" % rnd)), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
