#!/bin/bash
# round 2, session e: span-dispatched SGM pair kernel, point-map colour optimisation, Taylor reciprocal in the NCC loop, device resize:
# full GPU test-suite, SGM micro-benchmark, bench, kernel trace
TAG=${1:-r02_e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
# a box whose GPU does not complete a trivial job and the SGM micro-benchmark is not worth the session (r02_f: a faulty box cost 15 minutes)
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
timeout 120 python scripts/sgm_microbench.py 1 2>&1 | grep -q tiles || { echo "SGM micro-benchmark failed on this box"; exit 1; }
echo "== SGM microbench" | tee -a $OUT/log.txt
timeout 300 python scripts/sgm_microbench.py 1 8 2>&1 | grep tiles | tee $OUT/microbench.txt
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -12 $OUT/pytest.log
echo "== bench" | tee -a $OUT/log.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("frac","ms_per_launch","ms_whole_call_per_volume")})
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items()})
PY
tail -3 $OUT/bench.err
echo "== rocprofv3 kernel trace (bench)" | tee -a $OUT/log.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -16 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
echo "== done" | tee -a $OUT/log.txt
