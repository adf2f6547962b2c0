#!/bin/bash
# session r06_u: what bounds the eight-plane pass — timing variants with WRONG results (AVDM_WHATIF: 1 every T tap from one LDS address (broadcast), 2 conflict-free
# addresses, 3 = 2 without the transcendentals, 5 no T taps at all; 4 no s_nop after the dot2 blocks), per kernel (k1 = SGM sweep, k2 = Refine), on one box
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_u}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
for V in default wi1k1 wi2k1 wi3k1 wi5k1 wi1k2 wi2k2 wi3k2 wi5k2 wi4 default; do
  LIBV=$ROOT/alicevision_amd/csrc/libavdm.so; [ $V != default ] && LIBV=$ROOT/scripts/ab/$V/libavdm.so
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
