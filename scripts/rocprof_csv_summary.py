"""Condense rocprofv3 CSV output (-f csv) into one small CSV.

  python scripts/rocprof_csv_summary.py <dir> <out.csv>            kernel trace -> calls / total / avg / min / max (us) per kernel
  python scripts/rocprof_csv_summary.py <dir> <out.csv> counters   counter collection -> per kernel: launches, mean of every counter per launch
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def kernel_trace(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    meta = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""),
                       r.get("Scratch_Size", ""), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", "")))
    tot = sum(sum(v) for v in agg.values()) or 1.0
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_gpu_time", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes", "wg", "grid"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), f"{sum(v):.1f}", f"{sum(v) / len(v):.2f}", f"{min(v):.2f}", f"{max(v):.2f}", f"{100 * sum(v) / tot:.3f}"] + list(meta[k]))


def counters(d, out):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(dict)  # kernel -> dispatch -> ns (the duration of the launch UNDER the counter pass: what GRBM_GUI_ACTIVE is divided by)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    names = sorted({c for v in agg.values() for c in v})
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "us_per_launch"] + [n + "_per_launch" for n in names])
        for k in sorted(agg):
            n = max(len(disp[k]), 1)
            us = (sum(dur[k].values()) / len(dur[k]) / 1e3) if dur[k] else 0.0
            w.writerow([k, n, f"{us:.2f}"] + [f"{agg[k].get(c, 0.0) / n:.1f}" for c in names])


if __name__ == "__main__":
    (counters if len(sys.argv) > 3 and sys.argv[3] == "counters" else kernel_trace)(sys.argv[1], sys.argv[2])
    print(open(sys.argv[2]).read())
