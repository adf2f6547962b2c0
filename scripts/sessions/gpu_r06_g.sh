#!/bin/bash
# session r06_g: the tests touched after r06_d (outlier list against the oracle, the program's log assertions, the 12 MP corner tile against the
# literal oracle only) and the program's own timeline on the 11-camera job (every timestamped log line: where the 1.7 s outside the tiles go)
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== touched tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -k "refine_outlier_list or single_tile_equals or tiled_run_equals or reference_arithmetic_flags or (parity_of_default_tiles and corner)" 2>&1 | grep -E "outlier list|passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
echo "== the program's timeline (11 cameras, 12 MP, default tiling)"
AVDM_E2E_FILTER=0 AVDM_E2E_LOG=$ROOT/$OUT/cli_log.txt timeout 600 python scripts/cli_e2e_cfg3.py 11 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -30
grep -E "^\[" $OUT/cli_log.txt | cut -c1-170 | awk 'NR<=40 || /Worker|Batch 1\/|Batch 6\/|Task done|waited|published/' | head -90
echo "== done"
