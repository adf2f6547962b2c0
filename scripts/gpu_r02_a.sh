#!/bin/bash
# round 2, session a: new SGM oracle tests at scale, SGM probe (what bounds the pair kernel), literal-mode parity table, box state
TAG=${1:-r02_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
echo "== sgm probe" | tee -a $OUT/log.txt
timeout 300 scripts/probes/sgm_probe 2>&1 | tee $OUT/sgm_probe.txt
echo "== pytest sgm" | tee -a $OUT/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --no-header -p no:cacheprovider -k "sgm_aggregation" > $OUT/pytest_sgm.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -5 $OUT/pytest_sgm.log
echo "== box probe" | tee -a $OUT/log.txt
timeout 300 bash scripts/box_probe.sh 2>&1 | tee $OUT/box_probe.txt
echo "== parity report" | tee -a $OUT/log.txt
timeout 1200 python scripts/parity_report.py --out $OUT/parity_table.json > $OUT/parity_report.log 2>&1
echo "parity exit $?" | tee -a $OUT/log.txt
tail -c 3000 $OUT/parity_report.log
echo "== done" | tee -a $OUT/log.txt
