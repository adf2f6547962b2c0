#!/bin/bash
# last session of the round: smoke + the whole GPU suite on the library as committed (fused P2 maps, rigs in the host programs)
cd "$(dirname "$0")/../.."
TAG=${1:-r03_last}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
