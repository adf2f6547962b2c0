#!/bin/bash
# session G: the measurement session of round 3 — smoke, the GPU suite, the bench line (with the program's end-to-end rate), the kernel trace,
# the PMC passes of the similarity kernels and the FETCH / WRITE passes of the SGM pair kernel
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -14 $OUT/pytest.log
echo "== bench"
timeout 900 python bench.py --steps 11 --warmup 2 --cli-e2e 11 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"], "roofline", {k: r["roofline"].get(k) for k in ("frac","frac_whole_call","ms_per_launch_by_axis","box_copy_GBps","traffic")})
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items() if v > 0.4})
print("cli", r.get("cli_end_to_end")); print("cpu", r.get("cpu_baseline"))
PY
echo "== rocprofv3 kernel trace (bench)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -8 $OUT/kernel_stats.csv | cut -c1-160
echo "== PMC passes: similarity kernels"
bash scripts/pmc_similarity.sh $TAG 2>&1 | tail -12
echo "== PMC passes: SGM pair kernel (FETCH_SIZE / WRITE_SIZE)"
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/pmc_$PASS -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/pmc_$PASS.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/pmc_$PASS $OUT/pmc_$PASS.csv counters > /dev/null 2>&1
  head -4 $OUT/pmc_$PASS.csv
done
echo "== cfg5 (100 views x 24 MP, 4 x 4 tiles per depth map, tile buffers; 1 step on this one GPU)"
timeout 600 python bench.py --workload cfg5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "cfg5 exit $?"
python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_cfg5.json")); print("cfg5 value", r["value"], "ms/step", r["ms_per_step"], "roofline", {k: r["roofline"].get(k) for k in ("frac","volumes_per_launch","alg_bytes_per_launch")}, "valid", r["valid_fraction"])
except Exception as e:
    print("cfg5 failed", e, open("$OUT/bench_cfg5.err").read()[-600:])
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
