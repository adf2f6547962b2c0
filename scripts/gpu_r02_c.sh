#!/bin/bash
# round 2, session c: SGM probe 2 (scalar-base addressing, merged launches), full GPU test-suite, bench
TAG=${1:-r02_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
echo "== sgm probe 2" | tee -a $OUT/log.txt
timeout 300 scripts/probes/sgm_probe2 2>&1 | tee $OUT/sgm_probe2.txt
if [ "$2" != "nopytest" ]; then
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -5 $OUT/pytest.log
fi
if [ "$3" != "nobench" ]; then
echo "== bench" | tee -a $OUT/log.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
cat $OUT/bench.json; tail -3 $OUT/bench.err
fi
echo "== done" | tee -a $OUT/log.txt
