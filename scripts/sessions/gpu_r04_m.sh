#!/bin/bash
# session r04_m: rotating prefetch across the samples of a row (AVDM_NCC_MULTI_PIPE=2) in the eight-plane pass, and the rotating forms in the
# four-plane pass (scripts/ab/q_*: AVDM_QUAD_VIA_MULTI builds) against the default
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r04_m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in default p8_pipe_u1 p8_pipe2_u1 p8_pipe2_u3 q_pipe2_u3 q_pipe2_u1 q_pipe1_u3 default; do
  if [ $V = default ]; then unset AVDM_LIB; else export AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so; fi
  case $V in p8_*) P8=1;; *) P8=0;; esac
  AVDM_SIM_PLANES8=$P8 timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-12s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
echo "== done"
