#!/bin/bash
# session U: ring depth / granularity / cache-policy variants of the SGM pair kernel (scripts/ab/sgm_*), per-axis launch times on one box
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_u}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
for V in ${VARIANTS:-new buf plain ns2st0 wpb3}; do
  echo -n "$V: "
  AVDM_LIB=$ROOT/scripts/ab/sgm_$V/libavdm.so timeout 120 python scripts/sgm_axis_probe.py 1000x750x256 2>&1 | grep 1000x750
done
done | tee $OUT/variants.txt
timeout 60 python scripts/sgm_microbench.py 8 2>&1 | grep tiles
