#!/bin/bash
# session r05_unshifted2: what the unshifted-sums variant build (scripts/ab/unshifted: -DAVDM_DEV_UNSHIFTED_SUMS=1, both similarity kernels) costs,
# and the third tile case at the bar (tile24mp_interior) with it
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_unshifted2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( AVDM_LIB=$ROOT/scripts/ab/unshifted/libavdm.so timeout 500 python scripts/parity_report.py --cases tile24mp_interior,crop3 --literal-cases none --ref-cases none --modes literal --out $OUT/parity_unshifted2.json > $OUT/log.txt 2>&1 ) &
sleep 50
for V in unshifted tree; do
  L=$ROOT/scripts/ab/$V/libavdm.so; [ $V = tree ] && L=$ROOT/alicevision_amd/csrc/libavdm.so
  AVDM_LIB=$L timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$V.json
  python - $OUT/bench_$V.json $V <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%-10s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity']))
PY
done
wait
python - $OUT/parity_unshifted2.json <<'PY'
import json,sys
for r in json.load(open(sys.argv[1])):
    fd=r['literal']['final_depth']
    print(r['case'], 'unshifted sums on the fast path vs literal oracle: rmse %.3e (best 99.5 %% %.2e, max %.3f), volume identical %.3f' % (fd['rmse_untrimmed'], fd['rmse_best_99.5pct'], fd['max_abs'], r['literal']['similarity_volume_levels']['0']))
PY
echo "== done"
