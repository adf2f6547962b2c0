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


def main(tag):
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
    for d in per.values():
        # a wave issues a VALU instruction in SQ_ACTIVE_INST_VALU of its SQ_WAVE_CYCLES resident cycles (both in quad-cycles); two waves share a SIMD
        d["valu_active_per_wave"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        d["lds_conflict_frac"] = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
        d["wait_lds_frac_of_active"] = d["SQ_WAIT_INST_LDS"] / d["SQ_ACTIVE_INST_ANY"] if "SQ_WAIT_INST_LDS" in d and d.get("SQ_ACTIVE_INST_ANY") else None
    sha = hashlib.sha256(open(os.path.join(ROOT, "alicevision_amd", "csrc", "avdm_similarity.hip"), "rb").read()).hexdigest()
    out = {"source": "profiles/%s_simpmc_*.csv (rocprofv3 --pmc, counters only: scripts/pmc_similarity.sh)" % tag, "per_kernel": per, "kernel_source_sha256": sha,
           "waves_per_simd": 2}
    rnd = tag.split("_")[0]
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_sim_pmc.json" % rnd), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
