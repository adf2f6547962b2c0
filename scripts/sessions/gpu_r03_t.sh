#!/bin/bash
# session T: the core clock during the SGM pair kernel (SQ_BUSY_CYCLES over the kernel's duration) + per-axis times, to compare the two kinds of box
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r03_t}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python scripts/sgm_axis_probe.py 1000x750x256 250x750x256 1000x750x128 2>&1 | grep -v amdgpu.ids | tee $OUT/axis_probe.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/clk_pmc -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_pmc.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_pmc $OUT/clk_pmc.csv counters > /dev/null 2>&1; cat $OUT/clk_pmc.csv
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "sgm_pair_kernel" -f csv -d $ROOT/$OUT/clk_pmc2 -o pmc -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_pmc2.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_pmc2 $OUT/clk_pmc2.csv counters > /dev/null 2>&1; cat $OUT/clk_pmc2.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/clk_trace -o kt -- python $ROOT/scripts/sgm_microbench.py 1 > $ROOT/$OUT/clk_trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/clk_trace $OUT/clk_kernel_stats.csv > /dev/null 2>&1; grep -E "^kernel|sgm_pair" $OUT/clk_kernel_stats.csv | cut -c1-150
rocm-smi --showclocks --showpower 2>&1 | grep -E "GPU\[0\]" | cut -c1-100
cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1
find $OUT -name "*.db" -delete
