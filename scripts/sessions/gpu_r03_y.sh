#!/bin/bash
# session Y: the library with the shallower SGM ring (4 slots x 4 steps, plain stores) and the JPEG path: the new GPU tests, the SGM
# bit-exactness tests, per-axis times, PMC FETCH / WRITE (hash stamp of roofline.traffic), a bench line
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_y}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "jpeg or sgm or offset_tile or volume_exports or alembic or tiled_run or single_tile" > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_new.log
echo "== per-axis launch times"
timeout 200 python scripts/sgm_axis_probe.py 1000x750x256 1000x750x256 2>&1 | grep -v amdgpu.ids | tee $OUT/axis.txt
timeout 120 python scripts/sgm_microbench.py 1 8 2>&1 | grep tiles | tee -a $OUT/axis.txt
echo "== PMC FETCH / WRITE + bench (5 steps)"
bash scripts/gpu_pmc_sgm.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -12
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
