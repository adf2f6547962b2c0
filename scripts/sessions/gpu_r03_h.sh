#!/bin/bash
# session H: 12-byte LDS records in the SGM similarity kernel — GPU suite + bench with per-step times; A/B against the half-paired records
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_h; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -a "planes per pass (4 / 2) vs single planes: SGM best 0" $OUT/pytest.log | head -1; tail -6 $OUT/pytest.log
for V in rec12 half8; do
  [ $V = half8 ] && export AVDM_SIM_REC12=0
  timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  python - <<PY
import json
r=json.load(open("$OUT/bench_$V.json")); s=r["stages_ms"]
print("$V value %.4f sgm_sim %.1f refine_sim %.1f frac %.3f" % (r["value"], s["sgm_similarity"], s["refine_similarity"], r["roofline"]["frac"]), "per step", r.get("ms_per_step_each"))
PY
done
