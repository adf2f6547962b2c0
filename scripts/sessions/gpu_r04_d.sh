#!/bin/bash
# session r04_d: the new parity tests — cfg3 at its real shape (image corners, 10 T cameras), two default tiles of a 12 MP image, the deviation
# attribution against the reference's platform spread, the switch matrix, the untrimmed bars of the tile tests — with their measurements kept
cd "$(dirname "$0")/../.."
TAG=${1:-r04_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
AVDM_PARITY_DUMP=$OUT timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rP \
  -k "real_shape or default_tiles or deviation_attribution or switch_matrix or offset_tile or tiled_run" > $OUT/pytest.log 2>&1; echo "pytest exit $?"
grep -v "^child" $OUT/pytest.log | grep -E "passed|failed|FAILED|Error|assert|rmse|untrimmed|^cfg1|^crop3" | cut -c1-600 | tail -60
echo "== done"
