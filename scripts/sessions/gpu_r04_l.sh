#!/bin/bash
# session r04_l: scheduling variants of the eight-plane pass (scripts/ab/p8_*: rotating prefetch, unroll 1 / 3, all taps in flight) under
# AVDM_SIM_PLANES8=1 against the default
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r04_l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in default p8_g2_u1 p8_pipe_u1 p8_pipe_u3 p8_g4_u1 default; do
  if [ $V = default ]; then unset AVDM_LIB; P8=0; else export AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so; P8=1; fi
  AVDM_SIM_PLANES8=$P8 timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-12s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
echo "== done"
