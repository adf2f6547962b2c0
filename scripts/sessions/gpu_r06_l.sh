#!/bin/bash
# session r06_l: streams created beside the background tasks, mapped files not cached, two-channel writer, fast exit: the program's tests, A/B
# the device side released beside the last batch's merge + write: the program's tests, then the A/B on the bench's own scene
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== the program's tests"
timeout 900 python -m pytest tests/test_host_cli_gpu.py tests/test_filtering_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
echo "== A/B on the bench's scene"
REPEAT=3 timeout 900 python scripts/cli_e2e_ab.py $OUT "new:" "normal_exit:AVDM_HOST_EXIT=normal" "t1:AVDM_HOST_INGEST_THREADS=1" 2>&1 | grep -v amdgpu.ids
echo "== done"
