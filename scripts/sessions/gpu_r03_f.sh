#!/bin/bash
# session F: smoke + the whole GPU suite with the tree's library (Refine quad default, fractional mip levels), then the SGM aggregation A/B
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_f; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -a "planes per pass (4 / 2) vs single planes: SGM best 0" $OUT/pytest.log; tail -14 $OUT/pytest.log
STEPS=3 bash scripts/gpu_ab.sh r03_f tree wpb2 wpb3 nt0
python - <<'PY'
import json
for v in ("tree","wpb2","wpb3","nt0"):
    try:
        r=json.load(open("gpurun_out/r03_f/bench_%s.json"%v)); print(v, "per axis", r["roofline"].get("ms_per_launch_by_axis"), "mean", r["roofline"]["ms_per_launch"], "whole call", r["stages_ms"]["sgm_optimize"], "copy", r["roofline"].get("box_copy_GBps"))
    except Exception as e: print(v, e)
PY
