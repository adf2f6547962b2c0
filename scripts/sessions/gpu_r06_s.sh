#!/bin/bash
# session r06_s: chunk windows at a compile-time row pitch (AVDM_SGM_TPITCH / AVDM_REFINE_TPITCH: the bottom-row taps of the eight-plane pass become
# immediate offsets, 8 v_add_u32 fewer per sample): the widths of the bench's chunk windows (variant build with counters), A/B of the bench
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== widths of the chunk windows (variant build with counters): [24..26] SGM fits at 72 / 88 / wider, [28..30] Refine fits at 40 / 56 / wider"
AVDM_LIB=$ROOT/scripts/ab/leanstats/libavdm.so AVDM_LEAN_STATS=1 timeout 400 python bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/lean.err > $OUT/lean.json
python - $OUT/lean.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for row in r.get('lean_pass_counters_each') or []: print(row[16:32])
PY
for V in default tp_sgm tp_ref tp_ref56 tp_sgm56 default tp_sgm tp_ref tp_ref56 tp_sgm56; do
  LIBV=$ROOT/alicevision_amd/csrc/libavdm.so; [ $V != default ] && LIBV=$ROOT/scripts/ab/$V/libavdm.so
  AVDM_LIB=$LIBV timeout 400 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2> $OUT/bench_$V.err > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=r['stages_ms']
print(sys.argv[2], '%.4f maps/s %.1f ms' % (r['value'], r['ms_per_step']), 'sgm %.1f refine %.1f' % (s['sgm_similarity'], s['refine_similarity']))
PY
done
echo "== quick parity tests on the variant with both"
AVDM_LIB=$ROOT/scripts/ab/tpitch/libavdm.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "similarity_volume_parity or refine_volume_parity or end_to_end_depth_rmse or real_shape_of_cfg3 or four_planes_per_pass" 2>&1 | grep -E "passed|failed|^E  |FAILED" | cut -c1-400 | tail -20
echo "== done 1"
echo "== outlier-list test: units on the list for several wrong-depth factors / densities (the committed library)"
for F in 0.35 0.5 0.65 0.8 1.5; do for D in 0.02 0.06; do
  echo "factor $F density $D: $(AVDM_TEST_OUTLIER_FACTOR=$F AVDM_TEST_OUTLIER_DENSITY=$D timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k test_refine_outlier_list_equals_the_wave_fallback 2>&1 | grep -E "units worked|vs the ORACLE|AssertionError|passed|failed" | cut -c1-330 | tr '\n' '|')"
done; done
echo "== done 2"
