#!/bin/bash
# session r05_a (prepared at the end of round 4, not run yet): do AVDM_SIM_PLANES8=1 and AVDM_REFINE_PLANES8=1 keep the parity tables of DESIGN.md
# section 2?
#   1. parity tables of the cases the switch touches (12-byte records: the crops and tiles of the 4000 x 3000 geometry; cfg1's windows fit the
#      16-byte records and never reach the eight-plane pass), with the literal kernel and the reference's platform spread beside them;
#   2. the whole GPU suite under the switch: which thresholds move (expected: test_split_launches_equal_the_combined_kernels — the split
#      launches keep the four-plane pass, the combined kernel takes eight — and nothing else);
#   3. A/B bench on this box.
# If 1 holds the < 1e-3 lines and 2 shows only that one test: flip the defaults in avdm_volume_compute_similarity / avdm_volume_refine_similarity
# (planes8 = !(p8 && p8[0] == '0')),
# re-run scripts/pmc_similarity.sh + scripts/collect_sim_pmc.py (the default kernel changes: the certificate of r04 no longer applies).
cd "$(dirname "$0")/../.."
TAG=${1:-r05_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export AVDM_SIM_PLANES8=1 AVDM_REFINE_PLANES8=1
timeout 900 python scripts/parity_report.py --cases crop2,crop3,crop3_corner,tile12mp_corner --literal-cases crop2,crop3,crop3_corner,tile12mp_corner --spread-cases crop3 \
    --out $OUT/parity_planes8.json 2>&1 | grep -v amdgpu.ids | tail -40
AVDM_PARITY_DUMP=$OUT timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rP > $OUT/pytest_planes8.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|^FAILED|^E   " $OUT/pytest_planes8.log | cut -c1-400 | tail -30
unset AVDM_SIM_PLANES8 AVDM_REFINE_PLANES8
for V in 0 1 0 1; do
  AVDM_SIM_PLANES8=$V AVDM_REFINE_PLANES8=$V timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_p8_$V.json
  python - $OUT/bench_p8_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('PLANES8=%s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
echo "== done"
