#!/bin/bash
# session r04_f: the library as it stands at the end of round 4 (knife-edge rows on the reference's own border test; magic-number weight
# quantisation + R sums without w * dLR; colour optimisation on hardware rcp / rsq / exp2) — smoke, the driver's bench command, counter passes
# (SGM FETCH / WRITE, similarity VALU / LDS), kernel trace, then the whole GPU suite with the parity measurements kept
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r04_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/smoke.txt
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r['similarity'].get('valu_issue_frac'))
PY
echo "== PMC: SGM pair kernel FETCH / WRITE"
bash scripts/gpu_pmc_sgm.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12
echo "== PMC: similarity kernels"
bash scripts/pmc_similarity.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12
echo "== rocprofv3 kernel trace (bench, 3 steps)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-e2e 0 > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -12 $OUT/kernel_stats.csv | cut -c1-170
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete; rm -rf $OUT/trace
echo "== the whole GPU suite"
AVDM_PARITY_DUMP=$OUT timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rP > $OUT/pytest.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|^FAILED|^E   " $OUT/pytest.log | cut -c1-400 | tail -30
grep -E "^optimize parity|^cfg1 |^crop3 |untrimmed" $OUT/pytest.log | cut -c1-700
echo "== done"
