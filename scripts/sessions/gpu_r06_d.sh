#!/bin/bash
# session r06_d: the whole GPU suite in one process (the driver's command; its limit is 20 minutes), smoke, the driver's bench command
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== the whole GPU suite"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=25 > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-600 | tail -40
grep -E "^[0-9.]+s (call|setup)" $OUT/pytest.log | head -25
grep -E "final_sim: reference vs itself" $OUT/pytest.log | cut -c1-900
echo "== smoke"
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke.txt | cut -c1-900
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_call_span','ms_whole_call_per_volume','ms_whole_call_with_per_launch_events','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('reference_arithmetic')); print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r.get('fixed_job'))
PY
tail -3 $OUT/bench.err | cut -c1-300
echo "== done"
