#!/bin/bash
# session r04_a: the deviation-attribution table (default kernels, one deviation reverted at a time on the fast path, the literal kernel with one
# deviation introduced at a time) against the literal oracle, the reference evaluated the CUDA way (oracle/_ref/libavdm_ref_cuda.so) and the
# well-posed oracle; the literal / parity-table tests on the changed literal kernel
cd "$(dirname "$0")/../.."
TAG=${1:-r04_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
nproc | tee $OUT/nproc.txt
timeout 1500 python scripts/deviation_report.py --cases smoke,cfg1,crop2,crop3 --out $OUT/deviation_table.json 2>&1 | grep -v amdgpu.ids | tee $OUT/deviation_report.txt | grep -v "^child"
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "literal or parity_table or plane_pairs" > $OUT/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_sel.log
echo "== done"
