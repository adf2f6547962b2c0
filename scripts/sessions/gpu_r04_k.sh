#!/bin/bash
# session r04_k: de-interleaved T windows (AVDM_SIM_DEINT=1) and eight planes per pass (AVDM_SIM_PLANES8=1) of the SGM similarity kernel: A/B
# bench against the default and the equality test of the experiments
cd "$(dirname "$0")/../.."
TAG=${1:-r04_k}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in 00 10 11 00 10; do
  AVDM_SIM_DEINT=${V:0:1} AVDM_SIM_PLANES8=${V:1:1} timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('DEINT,PLANES8=%s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
timeout 300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "experiments_equal" 2>&1 | grep -E "passed|failed|^E  |vs default" | cut -c1-300
echo "== done"
