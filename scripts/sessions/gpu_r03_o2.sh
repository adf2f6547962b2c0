#!/bin/bash
# probe: do two depth maps in flight (two processes on the one GPU) finish more depth maps per second than one?
cd "$(dirname "$0")/../.."
TAG=${1:-r03_o2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $OUT/single.json 2> $OUT/single.err
python -c "
import json; r=json.load(open('$OUT/single.json')); print('single  value %.4f  ms/step %.1f' % (r['value'], r['ms_per_step']))"
(timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $OUT/a.json 2> $OUT/a.err &)
timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
sleep 8
python -c "
import json
a=json.load(open('$OUT/a.json')); b=json.load(open('$OUT/b.json'))
print('two processes: %.4f + %.4f = %.4f depth maps / s (ms/step %.1f, %.1f)' % (a['value'], b['value'], a['value']+b['value'], a['ms_per_step'], b['ms_per_step']))"
