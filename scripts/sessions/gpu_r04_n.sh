#!/bin/bash
# session r04_n: the library as committed — AVDM_SIM_PLANES8=1 (rotating prefetch, unroll 1) against the default, the equality test of the
# experiments and their rows of the switch matrix
cd "$(dirname "$0")/../.."
TAG=${1:-r04_n}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in 0 1; do
  AVDM_SIM_PLANES8=$V timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_p8_$V.json
  python - $OUT/bench_p8_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('PLANES8=%s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
timeout 300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "experiments_equal or (switch_matrix and (PLANES8 or DEINT))" 2>&1 | grep -E "passed|failed|^E  |vs default" | cut -c1-300
echo "== done"
