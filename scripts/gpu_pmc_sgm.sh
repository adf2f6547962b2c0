#!/bin/bash
# HBM traffic of the SGM pair kernel from the PMC counters (counters only, one per pass, never combined with tracing), on the
# micro-benchmark with one cfg3 volume; bench line with the live copy bandwidth
TAG=${1:-r03_pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
for PASS in "FETCH_SIZE" "WRITE_SIZE"; do
  echo "== rocprofv3 pmc $PASS (microbench, 1 volume)"
  (cd /tmp && timeout 300 rocprofv3 --pmc $PASS --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/pmc_$PASS -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/pmc_$PASS.log 2>&1)
  python scripts/rocprof_csv_summary.py $OUT/pmc_$PASS $OUT/pmc_$PASS.csv counters > /dev/null 2>&1
  cat $OUT/pmc_$PASS.csv | head -5
done
echo "== bench"
timeout 600 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json
r=json.load(open('$OUT/bench.json'))
print('value', r['value']); print({k: v for k, v in r['roofline'].items() if k in ('frac','ms_per_launch','box_copy_GBps','achieved_over_box_copy','frac_whole_call')})"
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
