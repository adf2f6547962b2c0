#!/bin/bash
# round 2, session b: full GPU test-suite (exact SGM fallback, pyramid quirk), SGM probe with clocks, A/B of the 7-tap row schedule in the
# similarity kernels (rebuilt here: hipcc is on the box), bench + kernel trace
TAG=${1:-r02_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
python -c "from alicevision_amd import abi; abi.load(); print('libavdm ok')" > $OUT/log.txt 2>&1
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -15 $OUT/pytest.log
echo "== sgm probe" | tee -a $OUT/log.txt
timeout 300 scripts/probes/sgm_probe 2>&1 | tee $OUT/sgm_probe.txt | head -12
echo "== similarity A/B (7-tap rows)" | tee -a $OUT/log.txt
CS=alicevision_amd/csrc
cp $CS/avdm_similarity.o /tmp/sim_default.o
for M in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DAVDM_NCC_W3_MODE=$M -c $CS/avdm_similarity.hip -o $CS/avdm_similarity.o 2> /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libavdm.so $CS/avdm_image.o $CS/avdm_similarity.o $CS/avdm_sgm.o $CS/avdm_maps.o $CS/avdm_fuse.o
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_w3m$M.json 2> $OUT/bench_w3m$M.err
  python - <<PY | tee -a $OUT/ab.txt
import json
r=json.load(open("$OUT/bench_w3m$M.json"))
s=r.get("stages_ms",{})
print("W3_MODE $M: value %.4f ms/step %.1f sgm_similarity %.1f refine_similarity %.1f" % (r["value"], r["ms_per_step"], s.get("sgm_similarity",0), s.get("refine_similarity",0)))
PY
done
cp /tmp/sim_default.o $CS/avdm_similarity.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libavdm.so $CS/avdm_image.o $CS/avdm_similarity.o $CS/avdm_sgm.o $CS/avdm_maps.o $CS/avdm_fuse.o
echo "== bench" | tee -a $OUT/log.txt
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprofv3 kernel trace (bench)" | tee -a $OUT/log.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -16 $OUT/kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
echo "== done" | tee -a $OUT/log.txt
