#!/bin/bash
# session r05_bench3: bench.py after the numerator became a function — the line still comes out (cfg3, 5 steps; cfg5 accounting with 2 steps)
cd "$(dirname "$0")/../.."
TAG=${1:-r05_bench3}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cli-e2e 0 2> $OUT/b.err > $OUT/b.json; python - $OUT/b.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); ro=r['roofline']
print('%.4f maps/s  frac %.4f kernels %.4f with_p2 %.4f  bytes/volume %.4e (%s)' % (r['value'], ro['frac'], ro['frac_kernels_only'], ro['frac_with_p2_map'], ro['alg_bytes_per_volume'], ro['alg_bytes']))
PY
tail -2 $OUT/b.err | cut -c1-300
echo "== done"
