#!/bin/bash
# round 2: plane pairs in the similarity kernels — parity of the affected stages, the pairs-vs-single statistics, and a short bench
TAG=${1:-r02_o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider \
    -k "plane_pairs or similarity_volume or refine_volume or chunk_window or end_to_end or odd_sizes" > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
grep -v "^$" $OUT/pytest.log | tail -25
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"])
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items() if v > 1})
PY
tail -3 $OUT/bench.err
echo "== done"
