#!/bin/bash
# session r04_b: the deviation-attribution table (see gpu_r04_a.sh; its first run died on an argparse slip), the whole GPU suite on the library
# with the split aggregation (avdm_volume_optimize_prepare / _tiles_prepared) and the arena-based pyramid exchange, and a bench line
cd "$(dirname "$0")/../.."
TAG=${1:-r04_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python scripts/deviation_report.py --cases smoke,cfg1,crop2,crop3 --out $OUT/deviation_table.json 2>&1 | grep -v amdgpu.ids | tee $OUT/deviation_report.txt | grep -v "^child"
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 11 --warmup 2 2> $OUT/bench.err | tee $OUT/bench.json | cut -c1-600
echo "== done"
