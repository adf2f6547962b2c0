#!/bin/bash
# One GPU-box session: parity tests, bench (LDS path and generic path), rocprofv3 kernel trace and PMC passes.
# usage (through gpurun): bash scripts/gpu_round.sh <tag> [quick]
# Everything is written under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r01_x}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
echo "== build check" | tee $OUT/log.txt
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" >> $OUT/log.txt 2>&1

echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -25 $OUT/pytest.log

echo "== bench (default)" | tee -a $OUT/log.txt
AVDM_SIM_STATS=1 timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
cat $OUT/bench.json
tail -5 $OUT/bench.err

if [ "$MODE" != "quick" ]; then
echo "== bench (generic path: no LDS staging)" | tee -a $OUT/log.txt
AVDM_SIM_LDS=0 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_generic.json 2> $OUT/bench_generic.err
cat $OUT/bench_generic.json

echo "== rocprofv3 kernel trace" | tee -a $OUT/log.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
cat $OUT/kernel_stats.csv | head -30

for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  echo "== rocprofv3 pmc $PASS" | tee -a $OUT/log.txt
  (cd /tmp && timeout 900 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_path_kernel|similarity_kernel" -f csv -d $ROOT/$OUT/pmc_$NAME -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $ROOT/$OUT/pmc_$NAME.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/pmc_$NAME $OUT/pmc_$NAME.csv counters >> $OUT/log.txt 2>&1
  cat $OUT/pmc_$NAME.csv | head -20
done
# keep the merge small: raw traces are large
find $OUT -name "*.csv" -size +2M -delete
fi
echo "== done" | tee -a $OUT/log.txt
