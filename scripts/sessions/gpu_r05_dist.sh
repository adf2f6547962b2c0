#!/bin/bash
# session r05_dist: the multi-GPU code path at full size with ONE rank (nccl process group): cfg4 (20 views x 12 MP, the default workload at N > 1)
# with the pyramids handed over once (the default) and as the streaming job (--stream-views)
cd "$(dirname "$0")/../.."
TAG=${1:-r05_dist}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for M in default stream; do
  X=""; [ $M = stream ] && X="--stream-views"
  timeout 300 python bench.py --gpus 1 --force-dist --workload cfg4 --steps 6 --warmup 2 --no-cpu-baseline --cli-e2e 0 $X 2> $OUT/bench_$M.err > $OUT/bench_$M.json
  python - $OUT/bench_$M.json $M <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('%-8s %.4f maps/s  %.1f ms  collectives %d  setup %.3f s  exchange %s commit %s  fixed_job %s' % (sys.argv[2], r['value'], r['ms_per_step'], r['config']['pyramid_exchange_collectives'], r['config']['pyramid_setup_broadcast_s'], s.get('pyramid_exchange'), s.get('pyramid_commit'), {k: r['fixed_job'][k] for k in ('cameras_per_rank','makespan_s','depth_maps_per_s')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
echo "== done"
