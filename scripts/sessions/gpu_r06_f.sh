#!/bin/bash
# session r06_f (for the record): the fast path with the reference's FORM of the NCC sums in the SGM sweep ONLY (variant build
# scripts/ab/unsh_sgm: -DAVDM_DEV_UNSHIFTED_SUMS=1 -DAVDM_DEV_UNSHIFTED_REFINE=0) on the three tile cases, crop3 and cfg1, and what it costs
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_f}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( AVDM_LIB=$ROOT/scripts/ab/unsh_sgm/libavdm.so timeout 1200 python scripts/parity_report.py --cases tile12mp_corner,tile24mp_interior,tile24mp_corner,crop3,cfg1 --literal-cases none --ref-cases none --modes literal --out $OUT/parity_unshifted_sgm.json > $OUT/log.txt 2>&1 ) &
sleep 60
for V in unsh_sgm tree; do
  L=$ROOT/scripts/ab/$V/libavdm.so; [ $V = tree ] && L=$ROOT/alicevision_amd/csrc/libavdm.so
  AVDM_LIB=$L timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-10s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
wait
python - $OUT/parity_unshifted_sgm.json <<'PY'
import json,sys
for r in json.load(open(sys.argv[1])):
    fd=r['literal']['final_depth']
    print(r['case'], 'unshifted sums in the SGM sweep only vs literal oracle: rmse %.3e (best 99.5 %% %.2e, max %.3f), volume identical %.3f' % (fd['rmse_untrimmed'], fd['rmse_best_99.5pct'], fd['max_abs'], r['literal']['similarity_volume_levels']['0']))
PY
echo "== done"
