#!/bin/bash
# session r06_w (final): the whole GPU suite in one process on the committed tree (the outlier-list test re-parametrised), smoke, BASELINE configuration 5
# on one GPU, the driver's bench command
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_w}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== the whole GPU suite"
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=12 > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-600 | tail -20
grep -E "^[0-9.]+s (call|setup)" $OUT/pytest.log | head -12
echo "== smoke"
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/smoke.txt | cut -c1-400
echo "== BASELINE configuration 5 on one GPU (100 views x 24 MP, 16 tiles per depth map)"
timeout 400 python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_cfg5.err > $OUT/bench_cfg5.json; python - $OUT/bench_cfg5.json <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('cfg5 %.4f maps/s  %.1f ms  frac %.3f kernels %.3f' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['frac_kernels_only'])); print({k: round(v, 3) for k, v in s.items()})
except Exception as e:
    print('cfg5 FAILED', e)
PY
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_call_span','ms_whole_call_per_volume','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('cli_end_to_end',{}).get('value'), r.get('cli_end_to_end',{}).get('value_over_kernel_only_rate')); print(r['similarity'].get('valu_issue_frac'))
PY
tail -3 $OUT/bench.err | cut -c1-300
echo "== done"
