#!/bin/bash
# session r04_o: smoke() (cfg1: the default-vs-oracle and default-vs-reference distances) with AVDM_SIM_PLANES8=1, for the record
cd "$(dirname "$0")/../.."
TAG=${1:-r04_o}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
AVDM_SIM_PLANES8=1 timeout 105 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/smoke_planes8.txt
echo "== done"
