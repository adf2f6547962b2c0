#!/bin/bash
# session r06_e (closing): counter passes over the shipped kernels (-> profiles/r06_sim_pmc.json, r06_sgm_pmc.json, which the bench line quotes),
# rocprofv3 kernel trace of the bench (default mode, and the parity mode's kernels), BASELINE configuration 5 on one GPU, the driver's bench command
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== PMC: similarity kernels"
bash scripts/pmc_similarity.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -16 | cut -c1-300
echo "== PMC: SGM pair kernel (FETCH_SIZE / WRITE_SIZE, micro-benchmark, 1 volume)"
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/pmc_$PASS -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/pmc_$PASS.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/pmc_$PASS $OUT/pmc_$PASS.csv counters > /dev/null 2>&1
  cat $OUT/pmc_$PASS.csv | head -5 | cut -c1-200
done
echo "== rocprofv3 kernel trace (bench, 3 steps, default mode)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1
head -8 $OUT/kernel_stats.csv | cut -c1-160; grep -i "outlier\|sgm_pair" $OUT/kernel_stats.csv | cut -c1-140
echo "== rocprofv3 kernel trace (bench, 1 step, both sweeps in the reference's arithmetic)"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace_ra -o kt -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --cli-e2e 0 --reference-arithmetic all > $ROOT/$OUT/trace_ra.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace_ra $OUT/kernel_stats_reference_arithmetic.csv > /dev/null 2>&1
head -5 $OUT/kernel_stats_reference_arithmetic.csv | cut -c1-160
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/trace $OUT/trace_ra
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
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_call_span','ms_whole_call_per_volume','ms_whole_call_with_per_launch_events','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('reference_arithmetic')); print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r['similarity'].get('valu_issue_frac'))
PY
echo "== done"
