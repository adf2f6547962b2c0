"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small CSV: one row per kernel with calls / total / average / min / max (us).
usage: python scripts/rocpd_summary.py <results.db> <out.csv> [name-filter]"""
import csv
import sqlite3
import sys


def main(dbp, out, flt=""):
    db = sqlite3.connect(dbp)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
                      "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_gpu_time", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes"])
        for r in rows:
            if flt and flt not in r[0]:
                continue
            name = r[0].split("(")[0].replace("void ", "")
            w.writerow([name, r[1], f"{r[2] / 1e3:.1f}", f"{r[3] / 1e3:.2f}", f"{r[4] / 1e3:.2f}", f"{r[5] / 1e3:.2f}", f"{100.0 * r[2] / tot:.3f}"] + list(r[6:]))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
