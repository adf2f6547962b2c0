#!/bin/bash
# session r05_b (prepared at the end of round 4, not run yet): the A/B forms of the eight-plane passes that round 4 built and could no longer
# measure.  Build the variants first (CPU, ~30 s each):
#   scripts/build_variant.sh p8_pipe3 -DAVDM_NCC_MULTI_PIPE=3         # + the R taps of the next sample requested during the last pair
#   scripts/build_variant.sh p8_pipe4 -DAVDM_NCC_MULTI_PIPE=4         # + the first pair's taps requested before the R side's arithmetic
#   scripts/build_variant.sh r8_partial -DAVDM_REFINE_OCTO_PARTIAL=1  # Refine: the last chunk (7 of 8 planes in range) through the eight-plane pass too
#   scripts/build_variant.sh p8_pipe4_partial -DAVDM_NCC_MULTI_PIPE=4 -DAVDM_REFINE_OCTO_PARTIAL=1
# Each run: both eight-plane switches on; 11 steps; ~10 s per run.
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in default p8_pipe3 p8_pipe4 r8_partial p8_pipe4_partial default; do
  if [ $V = default ]; then unset AVDM_LIB; else export AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so; fi
  [ $V != default ] && [ ! -f "$AVDM_LIB" ] && { echo "$V: not built"; continue; }
  AVDM_SIM_PLANES8=1 AVDM_REFINE_PLANES8=1 timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-18s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
unset AVDM_LIB
AVDM_LIB=$ROOT/scripts/ab/r8_partial/libavdm.so timeout 120 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "refine_similarity_experiment or (switch_matrix and REFINE_PLANES8)" 2>&1 | grep -E "passed|failed|^E  |vs default" | cut -c1-300
echo "== done"
