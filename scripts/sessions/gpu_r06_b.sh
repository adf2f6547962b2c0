#!/bin/bash
# session r06_b: after the level kernel's wrapped taps were fixed — the pyramid bit for bit, the reference-arithmetic stage tests, the parity-table
# tests with their new assertions (identical volumes in the parity mode), the CLI's flags, the bench with alternating instrumentation
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== libm restatements on THIS host"
timeout 300 python -m pytest tests/test_libm.py -q --no-header -p no:cacheprovider 2>&1 | tail -2
echo "== stage tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider --durations=12 -k "pyramid_parity or reference_arithmetic or texture_unit or refine_outlier or refine_volume_parity or end_to_end_depth" 2>&1 | tail -25 | cut -c1-400
echo "== parity-table tests"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider --durations=12 -k "parity_table_cfg1 or parity_of_default_tiles or ten_t_camera or cfg5_tiles or real_shape" 2>&1 | tail -40 | cut -c1-600
echo "parity-table tests: $(( $(date +%s) - T0 )) s"
echo "== the program's flags"
timeout 600 python -m pytest tests/test_host_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "reference_arithmetic or single_tile or tiled_run_equals" 2>&1 | tail -8 | cut -c1-400
echo "== bench"
timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench.err > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']; f=r['roofline']
print('%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v,3) for k,v in s.items()})
print({k: f.get(k) for k in ('frac','frac_kernels_only','frac_call_span','ms_whole_call_per_volume','ms_whole_call_with_per_launch_events','ms_per_launch_by_axis','depth_maps_with_per_launch_events','depth_maps_without','box_copy_GBps')})
PY
tail -3 $OUT/bench.err
echo "== done"
