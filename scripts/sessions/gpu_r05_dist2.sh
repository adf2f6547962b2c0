#!/bin/bash
# session r05_dist2: after the fix — with a process group (RCCL prints a banner through C stdio) the JSON line is the LAST line of stdout
cd "$(dirname "$0")/../.."
TAG=${1:-r05_dist2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for M in default stream; do
  X=""; [ $M = stream ] && X="--stream-views"
  timeout 300 python bench.py --gpus 1 --force-dist --workload cfg4 --steps 6 --warmup 2 --no-cpu-baseline --cli-e2e 0 $X 2> $OUT/bench_$M.err > $OUT/bench_$M.out
  python - $OUT/bench_$M.out $M <<'PY'
import json,sys
txt=open(sys.argv[1]).read().strip().split('\n')
print(sys.argv[2], 'stdout lines:', len(txt), '| last line is the JSON line:', txt[-1].startswith('{'), '| before it:', [t[:40] for t in txt[:-1]])
r=json.loads(txt[-1]); s=r['stages_ms']
print('   %.4f maps/s  %.1f ms  collectives %d  setup %.3f s  exchange %s commit %s  fixed_job %s' % (r['value'], r['ms_per_step'], r['config']['pyramid_exchange_collectives'], r['config']['pyramid_setup_broadcast_s'], s.get('pyramid_exchange'), s.get('pyramid_commit'), {k: r['fixed_job'][k] for k in ('cameras_per_rank','makespan_s','depth_maps_per_s')}))
json.dump(r, open(sys.argv[1].replace('.out', '.json'), 'w'))
PY
done
# and the driver's launcher with one rank: torchrun + nccl
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --workload cfg1 --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 2> $OUT/torchrun.err > $OUT/torchrun.out
python - $OUT/torchrun.out <<'PY'
import sys
txt=open(sys.argv[1]).read().strip().split('\n')
print('torchrun, WORLD_SIZE 1: stdout lines', len(txt), '| last is JSON', txt[-1].startswith('{'), '|', txt[-1][:120])
PY
echo "== done"
