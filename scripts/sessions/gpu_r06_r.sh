#!/bin/bash
# session r06_r (closing, on the windows-outside kernels of 80a24e4): the whole GPU suite in one process with its parity dumps, smoke, counter passes
# over the shipped similarity kernels (-> profiles/r06_sim_pmc.json re-stamped), kernel trace of the bench, the driver's bench command
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== the whole GPU suite"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=25 > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-600 | tail -40
grep -E "^[0-9.]+s (call|setup)" $OUT/pytest.log | head -12
echo "== smoke"
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke.txt | cut -c1-900
echo "== PMC: similarity kernels"
bash scripts/pmc_similarity.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -16 | cut -c1-300
echo "== rocprofv3 kernel trace (bench, 3 steps, default mode)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1
head -8 $OUT/kernel_stats.csv | cut -c1-160; grep -i "outlier\|sgm_pair" $OUT/kernel_stats.csv | cut -c1-140
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/trace
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_call_span','ms_whole_call_per_volume','ms_whole_call_with_per_launch_events','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('reference_arithmetic')); print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r['similarity'].get('valu_issue_frac'))
PY
tail -3 $OUT/bench.err | cut -c1-300
echo "== done"
