#!/bin/bash
# session r04_q: both eight-plane switches together against the default on one box (8 steps each)
cd "$(dirname "$0")/../.."
TAG=${1:-r04_q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in 0 1; do
  AVDM_SIM_PLANES8=$V AVDM_REFINE_PLANES8=$V timeout 30 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_both_$V.json
  python - $OUT/bench_both_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('SIM+REFINE PLANES8=%s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
echo "== done"
