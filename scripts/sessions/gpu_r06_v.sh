#!/bin/bash
# session r06_v: what bounds the eight-plane pass, second batch (WRONG results): 6 no s_load of the proximity table, 7 no LDS traffic and no s_load at all,
# 8 = 7 without the transcendentals; onewg = the shipped loops at ONE workgroup per compute unit (one wave per SIMD)
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_v}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
for V in default wi6k1 wi7k1 wi8k1 wi7k2 wi8k2 onewg default; do
  LIBV=$ROOT/alicevision_amd/csrc/libavdm.so; [ $V != default ] && LIBV=$ROOT/scripts/ab/$V/libavdm.so; unset AVDM_WHATIF_ONE_WG; [ $V = onewg ] && { LIBV=$ROOT/scripts/ab/wi9k1/libavdm.so; export AVDM_WHATIF_ONE_WG=1; }
  AVDM_LIB=$LIBV timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench_$V.err > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']
    print(sys.argv[2], '%.1f ms' % r['ms_per_step'], 'sgm %.1f refine %.1f' % (s['sgm_similarity'], s['refine_similarity']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "== done"
