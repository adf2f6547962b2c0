#!/bin/bash
# session J: Refine default instantiation with half of the LDS (larger T windows) — similarity / refine tests + bench with per-step times
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${TAG:-r03_j}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "not full_size and not parity_table_crops" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
r=json.load(open("$OUT/bench.json")); s=r["stages_ms"]
print("value %.4f sgm_sim %.1f refine_sim %.1f frac %.3f" % (r["value"], s["sgm_similarity"], s["refine_similarity"], r["roofline"]["frac"]), "per step", r.get("ms_per_step_each")); print(r["similarity_ms_each"])
PY
