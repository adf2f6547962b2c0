#!/bin/bash
# session r06_x: timing experiment (WRONG results): the SGM sweep's row taps as ONE ds_read_b128 each from the same 12-byte-record windows (address masked to
# 16 bytes: + 20 VALU instructions per sample) — what 16-byte T records would buy the SGM sweep
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
OUT=gpurun_out/r06_x; mkdir -p $OUT
export TMPDIR=/tmp
for V in default wi10 default wi10; do
  LIBV=$ROOT/alicevision_amd/csrc/libavdm.so; [ $V != default ] && LIBV=$ROOT/scripts/ab/$V/libavdm.so
  AVDM_LIB=$LIBV timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench_$V.err > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']
print(sys.argv[2], '%.1f ms' % r['ms_per_step'], 'sgm %.1f refine %.1f' % (s['sgm_similarity'], s['refine_similarity']))
PY
done
