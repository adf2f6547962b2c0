#!/bin/bash
# session r05_unshifted: do the reference's UNSHIFTED fp32 sums in its order (variant build -DAVDM_DEV_UNSHIFTED_SUMS=1 of the fast path) bring the
# three tile cases that sit at BASELINE's bar against the literal oracle back under it?  Literal evaluation only.
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_unshifted}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
AVDM_LIB=$ROOT/scripts/ab/unshifted/libavdm.so timeout 700 python scripts/parity_report.py --cases tile24mp_corner,tile12mp_corner --literal-cases none --ref-cases none --modes literal --out $OUT/parity_unshifted.json > $OUT/log.txt 2>&1
python - $OUT/parity_unshifted.json <<'PY'
import json,sys
for r in json.load(open(sys.argv[1])):
    fd=r['literal']['final_depth']
    print(r['case'], 'unshifted sums on the fast path vs literal oracle: rmse %.3e (best 99.5 %% %.2e, max %.3f), volume identical %.3f, oracle %.0f s' % (fd['rmse_untrimmed'], fd['rmse_best_99.5pct'], fd['max_abs'], r['literal']['similarity_volume_levels']['0'], r['literal']['t_oracle_s']))
PY
tail -2 $OUT/log.txt | cut -c1-300
echo "== done"
