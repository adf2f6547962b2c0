#!/bin/bash
# session r05_rbd: the Refine arg-min (refine_best_depth) with two sub-samples per packed register — its bit-exact tests, the end-to-end test of
# the small case, and the driver's bench command
cd "$(dirname "$0")/../.."
TAG=${1:-r05_rbd}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -k "refine_best_depth or end_to_end or smooth_and_upscale or optimize_parity or fractional" > $OUT/pytest.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|^FAILED|^E   " $OUT/pytest.log | cut -c1-300 | tail
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
PY
echo "== done"
