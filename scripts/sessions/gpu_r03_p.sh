#!/bin/bash
# probe: how fast does ONE wave per SIMD step?  columns per workgroup 4 / 2 / 1 (8 / 4 / 2 waves) over shapes that leave 2, 1, 0.5 waves per SIMD
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for V in new wpb2 wpb1; do
  echo "== $V"
  AVDM_LIB=$ROOT/scripts/ab/sgm_$V/libavdm.so timeout 120 python scripts/sgm_axis_probe.py 1000x750x256 500x750x256 250x750x256 125x750x256 2>&1 | grep "x750x"
done | tee $OUT/lone_wave.txt
