#!/bin/bash
# A/B session: bench (cfg3, 2 steps) of every library variant under scripts/ab/<name>/libavdm.so named on the command line, then the
# similarity-related parity tests with each candidate library.   usage (through gpurun): bash scripts/gpu_ab.sh <tag> <variant> ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
for V in "$@"; do
  LIB=$ROOT/scripts/ab/$V/libavdm.so
  [ "$V" = "tree" ] && LIB=$ROOT/alicevision_amd/csrc/libavdm.so
  AVDM_LIB=$LIB timeout 300 python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_$V.json"))
    s=r["stages_ms"]
    print("%-22s value %.4f  sgm_sim %.1f  refine_sim %.1f  sgm_opt %.3f  color_opt %.2f  frac %.3f whole %.3f" % ("$V", r["value"], s["sgm_similarity"], s["refine_similarity"], s["sgm_optimize"], s["color_optimize"], r["roofline"]["frac"], r["roofline"]["frac_whole_call"]))
except Exception as e:
    print("$V", "FAILED", e, open("$OUT/bench_$V.err").read()[-500:])
PY
done 2>&1 | tee -a $OUT/ab.txt
if [ -n "$TESTLIBS" ]; then
  for V in $TESTLIBS; do
    echo "== parity tests with $V" | tee -a $OUT/ab.txt
    AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "${TESTK:-plane_pairs or similarity_volume or refine_volume or end_to_end or parity_table or chunk_window}" 2>&1 | tail -8 | tee -a $OUT/ab.txt
  done
fi
