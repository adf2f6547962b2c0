#!/bin/bash
# session r06_h: after the outlier list was sized for every unit — the Refine tests, the program's tests (every one: the log must say "no unit
# refused"), the program on the wide-baseline scene that overflowed the old list (r06_g: 10.6 s, 348 670 refused units), a short bench
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== Refine tests + the program's tests"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_host_cli_gpu.py tests/test_filtering_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "refine or host_cli or filtering or end_to_end or switch_matrix or full_size_cfg3" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
echo "== the program on the wide-baseline scene (11 cameras, 12 MP, default tiling)"
AVDM_E2E_FILTER=0 AVDM_E2E_LOG=$ROOT/$OUT/cli_log.txt timeout 600 python scripts/cli_e2e_cfg3.py 11 2>&1 | grep -v amdgpu.ids | grep -E "wall|valid|exit" | cut -c1-200
grep -E "Refine outlier|Task done|set-up|waited|images decoded" $OUT/cli_log.txt | cut -c1-200 | head -12
echo "== bench"
timeout 400 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench.err > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']; f=r['roofline']
print('%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v,3) for k,v in s.items()})
print({k: f.get(k) for k in ('frac','frac_kernels_only','traffic','traffic_source')}); print(r['similarity'].get('valu_issue_frac'), (r['similarity'].get('valu_issue') or {}).get('source'))
PY
echo "== done"
