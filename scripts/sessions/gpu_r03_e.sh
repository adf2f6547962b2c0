#!/bin/bash
# session E: smoke + the whole GPU suite (new parity tests) with the tree's library, then A/B benches
OUT=gpurun_out/r03_e; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -a "planes per pass" $OUT/pytest.log; tail -14 $OUT/pytest.log
TESTLIBS="rq4" TESTK="plane_pairs or refine_volume or end_to_end or parity_table_cfg1" bash scripts/gpu_ab.sh r03_e tree rq4 rq4_1 wpb2 wpb3
python - <<'PY'
import json
for v in ("tree","wpb2","wpb3"):
    try:
        r=json.load(open("gpurun_out/r03_e/bench_%s.json"%v)); print(v, r["roofline"].get("ms_per_launch_by_axis"), r["roofline"]["ms_per_launch"], r["stages_ms"]["sgm_optimize"])
    except Exception as e: print(v, e)
PY
