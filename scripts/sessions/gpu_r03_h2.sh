#!/bin/bash
# the host programs as rebuilt after the closing session (camera rigs, Alembic depth guard): their GPU tests once more
cd "$(dirname "$0")/../.."
TAG=${1:-r03_h2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_host_cli_gpu.py tests/test_fuse_gpu.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_host.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_host.log
