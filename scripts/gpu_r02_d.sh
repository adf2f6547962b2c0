#!/bin/bash
# round 2, session d: SGM pair kernel with scalar-base addressing (bit-exact tests, micro-benchmark), point-map colour optimisation, bench
TAG=${1:-r02_d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
echo "== SGM microbench" | tee -a $OUT/log.txt
timeout 300 python scripts/sgm_microbench.py 1 8 2>&1 | grep tiles | tee $OUT/microbench.txt
echo "== pytest (sgm + optimise)" | tee -a $OUT/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "sgm or optimize" > $OUT/pytest_sgm.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -15 $OUT/pytest_sgm.log
echo "== bench" | tee -a $OUT/log.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("frac","ms_per_launch","ms_whole_call_per_volume")})
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items()})
PY
tail -3 $OUT/bench.err
echo "== done" | tee -a $OUT/log.txt
