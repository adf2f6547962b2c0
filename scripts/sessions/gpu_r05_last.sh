#!/bin/bash
# session r05_last: the driver's bench command once more on the tree as committed after the closing session (the counter summary of the shipped
# kernels is in profiles/ now: the line carries similarity.valu_issue)
cd "$(dirname "$0")/../.."
TAG=${1:-r05_last}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_with_p2_map','box_copy_GBps')})
print(json.dumps(r['similarity']['valu_issue'])[:1500])
print(r['cli_end_to_end']['value'], r['cli_end_to_end']['value_over_kernel_only_rate'])
PY
echo "== done"
