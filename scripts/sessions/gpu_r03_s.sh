#!/bin/bash
# session S: the SGM kernel timer that hands its events to the launch (hipExtLaunchKernelGGL) against the bracketing events and the
# kernel trace; the volume-export test; PMC FETCH / WRITE passes of the rebuilt avdm_sgm.hip (hash stamp of roofline.traffic)
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "volume_exports or alembic or sgm_aggregation or sgm_pair or offset_tile" > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_new.log
echo "== per-axis launch times: events handed to the launch (default)"
timeout 200 python scripts/sgm_axis_probe.py 1000x750x256 1000x750x256 2>&1 | grep -v amdgpu.ids | tee $OUT/axis_ext.txt
echo "== per-axis launch times: events recorded around the launch (AVDM_SGM_TIMER=record)"
AVDM_SGM_TIMER=record timeout 200 python scripts/sgm_axis_probe.py 1000x750x256 1000x750x256 2>&1 | grep -v amdgpu.ids | tee $OUT/axis_record.txt
echo "== kernel trace of the same"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/scripts/sgm_axis_probe.py 1000x750x256 > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1; grep -E "^kernel|sgm_pair" $OUT/kernel_stats.csv | cut -c1-150
grep 1000x750 $OUT/trace.log
echo "== PMC FETCH / WRITE + bench (5 steps)"
bash scripts/gpu_pmc_sgm.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12
echo "== bench, bracketing events (3 steps)"
AVDM_SGM_TIMER=record timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_record.json 2> $OUT/bench_record.err
python -c "
import json
r=json.load(open('$OUT/bench_record.json'))
print('value', r['value']); print({k: v for k, v in r['roofline'].items() if k in ('frac','ms_per_launch','ms_per_launch_by_axis','box_copy_GBps','frac_whole_call')})"
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
