#!/bin/bash
# session r05_e2e: the host program after the first batch's decode moved beside the device set-up: its GPU tests (bit-for-bit against the harness
# and the oracle) and the program end to end on the bench's scene
cd "$(dirname "$0")/../.."
TAG=${1:-r05_e2e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python -m pytest tests/test_host_cli_gpu.py tests/test_filtering_cli_gpu.py -m gpu -q --no-header > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-400 | tail -12
for i in 1 2; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench_$i.err > $OUT/bench_$i.json; python - $OUT/bench_$i.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); e=r['cli_end_to_end']
print('e2e %.4f maps/s wall %.2f s | split %s | over kernel-only rate %.3f, tiles %.3f' % (e['value'], e['wall_s'], {k: round(v, 3) if isinstance(v, float) else v for k, v in e['split'].items()}, e['value_over_kernel_only_rate'], e['tiles_s_over_kernel_only_s']))
PY
done
echo "== done"
