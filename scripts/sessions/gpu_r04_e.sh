#!/bin/bash
# session r04_e: the knife-edge rows evaluated with the reference's own border test (lit:: in avdm_similarity.hip) — deviation table again, the
# tests that failed in r04_d + every similarity parity test; A/B benches: default / without the knife-edge evaluation / the instruction-count
# experiments (magic-number weight quantisation + R sums without w * dLR) / the colour optimisation on hardware rcp, rsq, sqrt, exp2
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r04_e}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B benches (11 steps each)"
for V in default noknife exp_a exp_opt default; do
  if [ $V = default ]; then unset AVDM_LIB; else export AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so; fi
  timeout 300 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-8s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f  color_opt %.2f  sgm_opt %.3f (frac %.3f, kernels %.3f)  p2map %.3f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], s['color_optimize'], s['sgm_optimize'], r['roofline']['frac'], r['roofline']['frac_kernels_only'], s.get('sgm_p2_map', 0)))
PY
done
unset AVDM_LIB
echo "== parity of the experiments (their own libraries)"
AVDM_LIB=$ROOT/scripts/ab/exp_a/libavdm.so timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "similarity_volume_parity or refine_volume_parity or plane_pairs_equal or end_to_end_depth or odd_sizes or refine_chunk_window" 2>&1 | tail -3
AVDM_LIB=$ROOT/scripts/ab/exp_opt/libavdm.so timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "optimize" 2>&1 | tail -3
echo "== deviation table"
timeout 1500 python scripts/deviation_report.py --cases smoke,cfg1,crop2,crop3 --out $OUT/deviation_table.json 2>&1 | grep -v amdgpu.ids | tee $OUT/deviation_report.txt | grep -v "^child" | cut -c1-200
echo "== tests"
AVDM_PARITY_DUMP=$OUT timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rP \
  -k "real_shape or default_tiles or deviation_attribution or switch_matrix or similarity_volume_parity or refine_volume_parity or odd_sizes or end_to_end_depth or consistent_scale or fractional or custom_patch or parity_table or offset_tile or plane_pairs or split_launches or chunk_window" > $OUT/pytest.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|^FAILED|^E   " $OUT/pytest.log | cut -c1-400 | tail -40
grep -E "^cfg1 |^crop3 " $OUT/pytest.log | cut -c1-1200
echo "== done"
