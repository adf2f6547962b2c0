#!/bin/bash
# One lean GPU-box session: GPU parity tests, bench, rocprofv3 kernel trace of the bench, PMC passes of the SGM kernel on the
# micro-benchmark (counters only: never combined with tracing), rocprofv3 --list-avail.
# usage (through gpurun): bash scripts/gpu_round2.sh <tag> [nopytest]
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
if [ "$2" != "nopytest" ]; then
  echo "== pytest -m gpu" | tee -a $OUT/log.txt
  timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/log.txt
  tail -8 $OUT/pytest.log
fi
echo "== bench" | tee -a $OUT/log.txt
AVDM_SIM_STATS=1 timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== SGM microbench" | tee -a $OUT/log.txt
timeout 300 python scripts/sgm_microbench.py 1 2 4 8 2>&1 | grep tiles | tee $OUT/microbench.txt
echo "== rocprofv3 kernel trace (bench)" | tee -a $OUT/log.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -24 $OUT/kernel_stats.csv
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== rocprofv3 pmc $PASS (microbench, 1 volume)" | tee -a $OUT/log.txt
  (cd /tmp && timeout 300 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_pair_kernel|sgm_path_kernel" -f csv -d $ROOT/$OUT/pmc_$NAME -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/pmc_$NAME.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/pmc_$NAME $OUT/pmc_$NAME.csv counters >> $OUT/log.txt 2>&1
  cat $OUT/pmc_$NAME.csv | head -8
done
echo "== depth-map filtering: micro-benchmark + kernel trace" | tee -a $OUT/log.txt
timeout 300 python scripts/fuse_microbench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuse_microbench.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/fuse_trace -o kt -- python $ROOT/scripts/fuse_microbench.py 10 4000 3000 --no-cpu > $ROOT/$OUT/fuse_trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/fuse_trace $OUT/fuse_kernel_stats.csv >> $OUT/log.txt 2>&1
grep -i "fuse\|kernel" $OUT/fuse_kernel_stats.csv | head -6
(cd /tmp && timeout 120 rocprofv3 --list-avail > $ROOT/$OUT/list_avail.txt 2>&1)
grep -c "" $OUT/list_avail.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
echo "== done" | tee -a $OUT/log.txt
