#!/bin/bash
# session r06_o: which pass the waves of the default sweeps take, per reference camera (variant build with counters); the program once more
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_o}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== pass counters per camera (variant build)"
AVDM_LIB=$ROOT/scripts/ab/leanstats/libavdm.so AVDM_LEAN_STATS=1 timeout 400 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/lean.err > $OUT/lean.json
python - $OUT/lean.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r['ms_per_step_each']); print(r.get('similarity_ms_each'))
for row in r.get('lean_pass_counters_each') or []: print(row)
PY
echo "== the program"
REPEAT=2 timeout 600 python scripts/cli_e2e_ab.py $OUT "new:" 2>&1 | grep -v amdgpu.ids
grep -E "result tiles of set|released in|freed in" $OUT/new_0.log | cut -c1-220
echo "== done"
