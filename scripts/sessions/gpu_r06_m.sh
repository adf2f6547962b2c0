#!/bin/bash
# session r06_m: the driver's bench command with the new program (cli_end_to_end), and the similarity sweeps' tap-source counters per camera
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== the driver's command"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']; f=r['roofline']
print('%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v,3) for k,v in s.items()})
print({k: f.get(k) for k in ('frac','frac_kernels_only')}); print(json.dumps(r.get('cli_end_to_end'))[:1500])
PY
echo "== tap sources per camera"
AVDM_SIM_STATS=1 timeout 400 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/stats.err > $OUT/stats.json
python - $OUT/stats.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r['ms_per_step_each']); print(r.get('similarity_ms_each')); print(r.get('similarity_plane_workgroups_each'))
PY
echo "== done"
