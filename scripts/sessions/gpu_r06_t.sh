#!/bin/bash
# session r06_t: the outlier-list test re-parametrised (factors that throw the patch farther than a window holds)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for F in 0.12 0.15 0.2 0.25; do for D in 0.02 0.04; do
  echo "factor $F density $D: $(AVDM_TEST_OUTLIER_FACTOR=$F AVDM_TEST_OUTLIER_DENSITY=$D timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k test_refine_outlier_list_equals_the_wave_fallback 2>&1 | grep -E "^outlier list|vs the ORACLE|^E  .*Assert|passed|failed" | cut -c1-330 | tr '\n' '|')"
done; done
echo "== done"
