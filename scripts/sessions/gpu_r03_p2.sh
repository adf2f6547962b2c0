#!/bin/bash
# session P2: the fused two-axis P2-map kernel — bit-exactness of everything downstream of it (SGM tests, tiled runs), the whole-call
# fraction with it and with the per-axis kernel (AVDM_SGM_P2_MAP=legacy), kernel trace, PMC FETCH / WRITE re-stamp
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_p2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "sgm or offset_tile or tiled_run or single_tile or end_to_end or strict or quirk" > $OUT/pytest_sgm.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_sgm.log
for MODE in fused legacy fused legacy; do
  echo -n "$MODE: "
  AVDM_SGM_P2_MAP=$MODE timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=r['roofline']; print('value %.4f frac %.4f whole %.4f ms_whole %.4f' % (r['value'], f['frac'], f['frac_whole_call'], f['ms_whole_call_per_volume']))"
done | tee $OUT/p2_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv > /dev/null 2>&1; grep -E "^kernel|sgm_p" $OUT/kernel_stats.csv | cut -c1-150
echo "== PMC FETCH / WRITE + bench (5 steps)"
bash scripts/gpu_pmc_sgm.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
