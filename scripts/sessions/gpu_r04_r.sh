#!/bin/bash
# session r04_r: the switch-matrix row of AVDM_REFINE_PLANES8
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 25 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "switch_matrix and REFINE_PLANES8" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300
echo "== done"
