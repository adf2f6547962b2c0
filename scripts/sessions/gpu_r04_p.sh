#!/bin/bash
# session r04_p (the last 70 GPU seconds of the round): first contact of AVDM_REFINE_PLANES8=1 — its equality test and one bench run
cd "$(dirname "$0")/../.."
TAG=${1:-r04_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 40 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "refine_similarity_experiment" 2>&1 | grep -E "passed|failed|^E  |vs default" | cut -c1-300
AVDM_REFINE_PLANES8=1 timeout 40 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_refine8.json
python - $OUT/bench_refine8.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('REFINE_PLANES8=1 %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
echo "== done"
