#!/bin/bash
# session r04_c: the deviation-attribution table alone (variant libraries re-linked against the current objects)
cd "$(dirname "$0")/../.."
TAG=${1:-r04_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python scripts/deviation_report.py --cases smoke,cfg1,crop2,crop3 --out $OUT/deviation_table.json 2>&1 | grep -v amdgpu.ids | tee $OUT/deviation_report.txt | grep -v "^child"
echo "== done"
