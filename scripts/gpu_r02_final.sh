#!/bin/bash
# round 2, closing session: the library after the per-stream scratch (no stream-ordered allocator): full GPU test-suite and the bench line
TAG=${1:-r02_n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
if [ "$2" != "nomicro" ]; then
timeout 120 python scripts/sgm_microbench.py 1 2>&1 | grep tiles | tee $OUT/microbench.txt
grep -q tiles $OUT/microbench.txt || { echo "SGM micro-benchmark failed on this box"; exit 1; }
fi
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -14 $OUT/pytest.log
echo "== bench" | tee -a $OUT/log.txt
timeout 300 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("frac","ms_per_launch","ms_whole_call_per_volume","box_copy_GBps","achieved_over_box_copy")})
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items()})
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["kind"])
PY
tail -3 $OUT/bench.err
echo "== done" | tee -a $OUT/log.txt
