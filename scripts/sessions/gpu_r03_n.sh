#!/bin/bash
# session N: the similarity kernels as fast + fix-up launches — smoke, GPU suite, bench with per-step times
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${TAG:-r03_n}; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest.log
timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
r=json.load(open("$OUT/bench.json")); s=r["stages_ms"]
print("value %.4f sgm_sim %.1f refine_sim %.1f frac %.3f" % (r["value"], s["sgm_similarity"], s["refine_similarity"], r["roofline"]["frac"]), "per step", r.get("ms_per_step_each")); print(r["similarity_ms_each"])
PY
