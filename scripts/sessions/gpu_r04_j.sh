#!/bin/bash
# session r04_j: first GPU contact of the eight-planes-per-pass form of the SGM similarity kernel (AVDM_SIM_PLANES8=1, experimental): A/B bench
# against the default, and the similarity parity tests under the switch
cd "$(dirname "$0")/../.."
TAG=${1:-r04_j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in 0 1 0 1; do
  AVDM_SIM_PLANES8=$V timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_p8_$V.json
  python - $OUT/bench_p8_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('PLANES8=%s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
AVDM_SIM_PLANES8=1 timeout 400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rP -k "similarity_volume_parity or plane_pairs_equal or end_to_end_depth or crops_of_the_full_size_geometry and crop2" 2>&1 | grep -E "passed|failed|^E  |planes per pass" | cut -c1-300
echo "== done"
