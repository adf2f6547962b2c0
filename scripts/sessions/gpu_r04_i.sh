#!/bin/bash
# session r04_i: the deviation-attribution table on the FINAL kernels (biased magic-number quantisation, sums without w * dLR, knife-edge rows)
cd "$(dirname "$0")/../.."
TAG=${1:-r04_i}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/deviation_report.py --cases smoke,cfg1,crop2,crop3 --out $OUT/deviation_table.json 2>&1 | grep -v amdgpu.ids | tee $OUT/deviation_report.txt | grep -v "^child" | cut -c1-200 | grep -E "^==|default  |literal  |cuda_vs_lit|all_reverted|literal\+all "
echo "== done"
