#!/bin/bash
# session R: the GPU cases added after the closing session (Alembic scene input, volume exports) and a diagnosis of the SGM pair kernel
# on whichever kind of box this lands on: per-axis launch times over volume shapes, the core clock during the kernel (SQ_BUSY_CYCLES
# over its duration), the HBM probe and the device state
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "alembic or volume_exports or split_launches or volume_init" > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest_new.log
echo "== per-axis launch times over shapes"
timeout 300 python scripts/sgm_axis_probe.py 1000x750x256 500x750x256 250x750x256 1000x376x256 1000x750x128 1000x750x512 2000x750x256 2>&1 | grep -v amdgpu.ids | tee $OUT/axis_probe.txt
echo "== micro-benchmark 1 / 8 volumes"
timeout 120 python scripts/sgm_microbench.py 1 8 2>&1 | grep tiles | tee $OUT/microbench.txt
echo "== core clock during the pair kernel: SQ_BUSY_CYCLES / duration"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/clk_pmc -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_pmc.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_pmc $OUT/clk_pmc.csv counters > /dev/null 2>&1; cat $OUT/clk_pmc.csv
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/clk_pmc2 -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_pmc2.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_pmc2 $OUT/clk_pmc2.csv counters > /dev/null 2>&1; cat $OUT/clk_pmc2.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/clk_trace -o kt -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_trace $OUT/clk_kernel_stats.csv > /dev/null 2>&1; grep -E "^kernel|sgm_pair" $OUT/clk_kernel_stats.csv | cut -c1-150
echo "== box probe"
bash scripts/box_probe.sh 2>&1 | grep -v amdgpu.ids | tee $OUT/box_probe.txt
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
