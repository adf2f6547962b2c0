#!/bin/bash
# session r05_sanity: the tree exactly as committed at the end of the round, with the in-tree libraries as rebuilt by build(): smoke(), a slice of
# the GPU suite through every library (kernels, host program, filtering), a short bench
cd "$(dirname "$0")/../.."
TAG=${1:-r05_sanity}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
timeout 400 python -m pytest tests -m gpu -x -q --no-header -k "library_loaded or similarity_volume_parity or refine_volume_parity or sgm_aggregation_parity or end_to_end or outlier_list or single_tile_equals or fuse_filter or test_abi or switch_matrix" > $OUT/pytest.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|^FAILED|^E   " $OUT/pytest.log | cut -c1-300 | tail -5
timeout 200 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench.json; python - $OUT/bench.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); print('%.4f maps/s  %.1f ms  frac %.3f' % (r['value'], r['ms_per_step'], r['roofline']['frac']), r['similarity']['valu_issue_frac'])
PY
echo "== done"
