#!/bin/bash
# session r06_p: windows that reach outside the T image (clamped staging): the similarity parity tests, A/B of the bench against the variant that
# keeps them inside (rounds 1-6a), pass counters of the new default
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== similarity / end-to-end parity tests (the quick ones)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "similarity or refine_volume or end_to_end or switch_matrix or corner or cfg1" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
for V in new inside new inside; do
  LIBV=$ROOT/alicevision_amd/csrc/libavdm.so; [ $V = inside ] && LIBV=$ROOT/scripts/ab/inside/libavdm.so
  AVDM_LIB=$LIBV timeout 400 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench_$V.err > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']
print(sys.argv[2], '%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step']), 'sgm %.1f refine %.1f' % (s['sgm_similarity'], s['refine_similarity']), r['ms_per_step_each'])
PY
done
echo "== pass counters per camera (variant build of the new default)"
AVDM_LIB=$ROOT/scripts/ab/leanstats/libavdm.so AVDM_LEAN_STATS=1 timeout 400 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/lean.err > $OUT/lean.json
python - $OUT/lean.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r.get('similarity_ms_each'))
for row in r.get('lean_pass_counters_each') or []: print(row[:24])
PY
echo "== done"
