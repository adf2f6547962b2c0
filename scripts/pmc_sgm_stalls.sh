#!/bin/bash
# PMC stall breakdown of the SGM pair kernel on the micro-benchmark (counters only, one group per pass).
# usage (through gpurun): bash scripts/pmc_sgm_stalls.sh <tag> [n_tiles]
TAG=${1:-r01_x}; NT=${2:-1}
OUT=gpurun_out/$TAG; mkdir -p $OUT
ROOT=$(pwd)
export TMPDIR=/tmp
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VMEM_RD"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/stall_$NAME -o pmc -- python $ROOT/scripts/sgm_microbench.py $NT > $ROOT/$OUT/stall_$NAME.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/stall_$NAME $OUT/stall_${NT}t_$NAME.csv counters > /dev/null 2>&1
  cat $OUT/stall_${NT}t_$NAME.csv
done
find $OUT -name "*.db" -delete
