#!/bin/bash
# session r05_final: the library as committed — the whole GPU suite in one process (the driver's command and its 20-minute limit), smoke, the
# driver's bench command, counter passes over the similarity kernels (-> profiles/r05_sim_pmc.json, which the bench line quotes), kernel trace,
# BASELINE configuration 5 on one GPU
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== the whole GPU suite"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 1190 python -m pytest tests -m gpu -q --no-header --durations=30 > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-500 | tail -30
grep -E "^[0-9.]+s (call|setup)" $OUT/pytest.log | head -16
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err > $OUT/bench_final.json; python - $OUT/bench_final.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_with_p2_map','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r.get('fixed_job')); print(r['similarity'].get('valu_issue_frac')); print(r['similarity_ms_each']['refine_similarity'])
PY
echo "== PMC: similarity kernels"
bash scripts/pmc_similarity.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300
echo "== rocprofv3 kernel trace (bench, 3 steps)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1
head -6 $OUT/kernel_stats.csv | cut -c1-150; grep -i "outlier\|sgm_pair" $OUT/kernel_stats.csv | cut -c1-120
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/trace
echo "== smoke"
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/smoke.txt | cut -c1-600
echo "== BASELINE configuration 5 on one GPU (100 views x 24 MP, 16 tiles per depth map)"
timeout 400 python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_cfg5.err > $OUT/bench_cfg5.json; python - $OUT/bench_cfg5.json <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('cfg5 %.4f maps/s  %.1f ms  frac %.3f kernels %.3f' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['frac_kernels_only'])); print({k: round(v, 3) for k, v in s.items()})
except Exception as e:
    print('cfg5 FAILED', e)
PY
echo "== done"
