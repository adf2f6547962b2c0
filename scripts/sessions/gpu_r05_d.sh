#!/bin/bash
# session r05_d (one minute of GPU work): the outlier-list test alone, with its numbers and its failure text if it fails
cd "$(dirname "$0")/../.."
TAG=${1:-r05_d}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -s -x -k "outlier_list or four_planes_per_pass or refine_similarity_experiment" > $OUT/pytest.log 2>&1; echo "exit $?"
grep -E "passed|failed|^E  |outlier list:|four vs eight|vs default|Error|assert" $OUT/pytest.log | cut -c1-500 | tail -40
echo "== done"
