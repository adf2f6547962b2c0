#!/bin/bash
# PMC passes (counters only) over the two similarity kernels of one bench step: VALU / LDS activity and LDS bank conflicts.
# usage (through gpurun): bash scripts/pmc_similarity.sh <tag>
TAG=${1:-r01_x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
# (GRBM_GUI_ACTIVE over the launch's duration under the pass = the clock the kernel really ran at; the fourth pass: waves parked on a counter /
# barrier, vector-memory activity — scratch traffic would show there — and the scalar / transcendental share of the instruction stream)
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 400 rocprofv3 --pmc $PASS --kernel-include-regex "similarity_kernel" -f csv -d $ROOT/$OUT/simpmc_$NAME -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost > $ROOT/$OUT/simpmc_$NAME.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/simpmc_$NAME $OUT/simpmc_$NAME.csv counters >> $OUT/log.txt 2>&1
  cat $OUT/simpmc_$NAME.csv | head -6
done
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
