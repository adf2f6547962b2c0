#!/bin/bash
# session r05_c (short): the outlier-list test with its numbers; A/B of the anchored T window of the Refine kernel (tree) against the round-4 tiers
# (scripts/ab/noanchor); the kernel trace of two steps (what the list's kernel costs); in the background, on the host cores: the parity
# measurements of the tile cases the suite no longer carries in full (12 MP corner tile with the reference's own spread, both 24 MP tiles).
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( OMP_NUM_THREADS=80 timeout 900 python scripts/parity_report.py --cases tile24mp_corner --literal-cases tile24mp_corner --spread-cases tile24mp_corner --ref-cases none --out $OUT/parity_tile24mp_corner.json > $OUT/parity_bg1.log 2>&1 ) &
( OMP_NUM_THREADS=80 timeout 900 python scripts/parity_report.py --cases tile24mp_interior --literal-cases tile24mp_interior --ref-cases none --out $OUT/parity_tile24mp_interior.json > $OUT/parity_bg2.log 2>&1 ) &
( OMP_NUM_THREADS=80 timeout 900 python scripts/parity_report.py --cases tile12mp_corner,crop3_10T --literal-cases tile12mp_corner,crop3_10T --spread-cases tile12mp_corner --ref-cases none --out $OUT/parity_tile12mp_corner_10T.json > $OUT/parity_bg3.log 2>&1 ) &
sleep 45   # their scenes are rendered and swept on the GPU first; the benches below want it to themselves
echo "== tests"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -s -k "outlier_list or four_planes_per_pass or refine_similarity_experiment or (switch_matrix and (OUTLIER or PLANES8))" 2>&1 | grep -E "passed|failed|^E  |outlier list:|four vs eight|vs default" | cut -c1-400
echo "== A/B bench: anchored window"
run() { # name, env...
  N=$1; shift
  env "$@" AVDM_REFINE_OUTLIER_STATS=1 timeout 200 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$N.json
  python - $OUT/bench_$N.json $N <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('%-12s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f  refine each %s  units %s' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], r['similarity_ms_each']['refine_similarity'], r.get('refine_outlier_units')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run noanchor_a AVDM_LIB=$ROOT/scripts/ab/noanchor/libavdm.so
run anchored_a X=1
run noanchor_b AVDM_LIB=$ROOT/scripts/ab/noanchor/libavdm.so
run anchored_b X=1
echo "== rocprofv3 kernel trace (bench, 3 steps)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1
head -8 $OUT/kernel_stats.csv | cut -c1-150
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/trace
echo "== waiting for the parity measurements"
wait
python - $OUT <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "parity_*.json"))):
    for r in json.load(open(f)):
        fd = lambda m: (m["final_depth"]["rmse_untrimmed"], m["final_depth"].get("rmse_untrimmed_relative"), m["final_depth"].get("rmse_untrimmed_in_pixsize"))
        line = "%-18s wp %.2e rel %.1e pix %.2f | lit %.2e rel %.1e pix %.2f | gpu-lit %.2e | vol0 wp %.3f lit %.3f | t %.0f %.0f" % ((r["case"],) + fd(r["well_posed"]) + fd(r["literal"]) + (r["gpu_literal_vs_oracle_literal"]["final_depth"]["rmse_untrimmed"], r["well_posed"]["similarity_volume_levels"]["0"], r["literal"]["similarity_volume_levels"]["0"], r["well_posed"]["t_oracle_s"], r["literal"]["t_oracle_s"]))
        sp = r.get("platform_spread")
        if sp:
            for k in ("cuda_vs_literal_interior", "default_vs_literal_interior", "default_vs_cuda_interior", "well_posed_vs_literal_interior"):
                if sp.get(k):
                    line += " %s %.2e" % (k.replace("_interior", "_in").replace("_vs_", "/"), sp[k]["final_depth"]["rmse_untrimmed"])
            line += " t %.0f" % sp["t_s"]
        print(line)
PY
tail -3 $OUT/parity_bg*.log | cut -c1-300
echo "== done"
