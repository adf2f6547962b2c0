"""Copy the judged summaries of one GPU session (gpurun_out/<tag>/, written by scripts/gpu_round.sh) into profiles/ and derive
profiles/<round>_sgm_pmc.json (HBM traffic per sgm_path_kernel launch) from the FETCH_SIZE / WRITE_SIZE passes.

    python scripts/collect_profiles.py r01_b

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 this rocprofv3 reports exactly half of the bytes of a wide coalesced
streaming read in FETCH_SIZE (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"); WRITE_SIZE matched the known byte count of this kernel
(192 000 000 B per path) to the byte, so it is used as is.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for name in os.listdir(src):
        if name.endswith(".csv") or name in ("bench.json", "bench_generic.json", "pytest.log", "microbench.txt", "fuse_microbench.txt"):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    fetch = {r["kernel"]: float(r["FETCH_SIZE_per_launch"]) for r in csv.DictReader(open(os.path.join(src, "pmc_FETCH_SIZE.csv")))}
    write = {r["kernel"]: float(r["WRITE_SIZE_per_launch"]) for r in csv.DictReader(open(os.path.join(src, "pmc_WRITE_SIZE.csv")))}
    per = {}
    for k in fetch:
        if "sgm_path_kernel" in k or "sgm_pair_kernel" in k:
            per[k] = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
    import hashlib
    # the kernel these counters belong to: bench.py only quotes them while csrc/avdm_sgm.hip still hashes to this (roofline.traffic cannot go stale)
    sha = hashlib.sha256(open(os.path.join(ROOT, "alicevision_amd", "csrc", "avdm_sgm.hip"), "rb").read()).hexdigest()
    out = {"source": f"profiles/{tag}_pmc_FETCH_SIZE.csv + {tag}_pmc_WRITE_SIZE.csv", "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB -> B)",
           "per_kernel_bytes": per, "hbm_bytes_per_launch": sum(per.values()) / max(len(per), 1), "kernel_source_sha256": sha}
    rnd = tag.split("_")[0]
    json.dump(out, open(os.path.join(dst, f"{rnd}_sgm_pmc.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
