#!/bin/bash
# session r06_j: EXR scan lines de-interleaved on the device (avdm_image_decode_exr_lines; an uncompressed file is mapped and uploaded as it
# lies), first batch's ingest beside the set-up: kernel test, the program's tests, A/B on the bench's own scene
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null)  nproc $(nproc)  OMP_NUM_THREADS=$OMP_NUM_THREADS"
echo "== kernel test + the program's tests"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_cli_gpu.py tests/test_filtering_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "exr_lines or host_cli or filtering" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
echo "== A/B on the bench's scene"
REPEAT=2 timeout 900 python scripts/cli_e2e_ab.py $OUT "old:AVDM_HOST_ARENA=0,AVDM_HOST_INGEST=serial,AVDM_HOST_EXR=host" "host_exr:AVDM_HOST_EXR=host" "new:" "new_serial:AVDM_HOST_INGEST=serial" 2>&1 | grep -v amdgpu.ids
echo "== done"
