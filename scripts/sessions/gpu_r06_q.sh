#!/bin/bash
# session r06_q: extreme planes that leave the T image count in the hull: quick parity tests, bench, pass counters
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== quick parity tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "similarity_volume_parity or refine_volume_parity or end_to_end_depth_rmse or real_shape_of_cfg3" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
for V in new new; do
  timeout 400 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench_$V.err > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']
print(sys.argv[2], '%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step']), 'sgm %.1f refine %.1f' % (s['sgm_similarity'], s['refine_similarity']), r['ms_per_step_each'])
PY
done
echo "== pass counters per camera (variant build of the new default)"
AVDM_LIB=$ROOT/scripts/ab/leanstats/libavdm.so AVDM_LEAN_STATS=1 AVDM_REFINE_OUTLIER_STATS=1 timeout 400 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/lean.err > $OUT/lean.json
python - $OUT/lean.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r.get('similarity_ms_each')); print(r.get('refine_outlier_units'))
for row in r.get('lean_pass_counters_each') or []: print(row[:24])
PY
echo "== done"
