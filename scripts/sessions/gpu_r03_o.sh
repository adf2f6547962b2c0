#!/bin/bash
# session O: combined kernels as the default again; the split behind AVDM_SIM_SPLIT=1 (equality test + its cost); Refine rows as 3 + 3 + 1
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_o; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "split_launches or plane_pairs or similarity_volume or refine_volume or end_to_end or chunk_window or offset_tile or fractional" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
for V in combined split q331; do
  unset AVDM_SIM_SPLIT AVDM_LIB
  [ $V = split ] && export AVDM_SIM_SPLIT=1
  [ $V = q331 ] && export AVDM_LIB=$(pwd)/scripts/ab/q331/libavdm.so
  timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  python - <<PY
import json
r=json.load(open("$OUT/bench_$V.json")); s=r["stages_ms"]
print("$V value %.4f sgm_sim %.1f refine_sim %.1f frac %.3f" % (r["value"], s["sgm_similarity"], s["refine_similarity"], r["roofline"]["frac"]), "refine per step", r["similarity_ms_each"]["refine_similarity"])
PY
done
