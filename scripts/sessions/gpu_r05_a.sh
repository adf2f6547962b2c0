#!/bin/bash
# session r05_a: eight planes per pass as the DEFAULT of both similarity kernels (flipped before this session) —
#   1. the whole GPU suite in pytest-xdist workers (new: one worker per parity case; cfg5-shape tiles; tile-case platform spread), every parity
#      measurement dumped (-> profiles/r05_a_parity_*.json);
#   2. counter passes over the similarity kernels, eight planes (default) and four planes (AVDM_*_PLANES8=0) in the same session: instruction
#      counts, real clock (GRBM_GUI_ACTIVE / duration), VALU-busy, LDS waits, parked waves, vector-memory activity;
#   3. A/B bench: four / eight planes, the prepared variants of the eight-plane passes.
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== the whole GPU suite (xdist)"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 1500 python -m pytest tests -m gpu -q --no-header -rP > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   " $OUT/pytest.log | cut -c1-600 | tail -40
python - $OUT <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "parity_*.json"))):
    r = json.load(open(f))
    fd = lambda m: (m["final_depth"]["rmse_untrimmed"], m["final_depth"].get("rmse_untrimmed_relative"), m["final_depth"].get("rmse_untrimmed_in_pixsize"))
    line = "%-18s wp %.2e rel %.1e pix %.2f | lit %.2e rel %.1e pix %.2f | gpu-lit %.2e | vol0 wp %.3f lit %.3f" % ((r["case"],) + fd(r["well_posed"]) + fd(r["literal"]) + (r["gpu_literal_vs_oracle_literal"]["final_depth"]["rmse_untrimmed"], r["well_posed"]["similarity_volume_levels"]["0"], r["literal"]["similarity_volume_levels"]["0"]))
    sp = r.get("platform_spread")
    if sp:
        line += " | X %.2e" % sp["cuda_vs_literal"]["final_depth"]["rmse_untrimmed"]
        for k in ("cuda_vs_literal_interior", "default_vs_literal_interior", "default_vs_cuda_interior", "well_posed_vs_literal_interior", "literal_oracle_vs_reference_tile_interior"):
            if k in sp:
                line += " %s %.2e" % (k.replace("_interior", "_in").replace("_vs_", "/"), sp[k]["final_depth"]["rmse_untrimmed"])
    print(line)
PY
echo "== PMC: similarity kernels, eight planes (default)"
bash scripts/pmc_similarity.sh ${TAG}_p8 2>&1 | grep -v amdgpu.ids | tail -14
echo "== PMC: similarity kernels, four planes"
AVDM_SIM_PLANES8=0 AVDM_REFINE_PLANES8=0 bash scripts/pmc_similarity.sh ${TAG}_p4 2>&1 | grep -v amdgpu.ids | tail -14
echo "== A/B bench"
run() { # name, env...
  N=$1; shift
  env "$@" timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$N.json
  python - $OUT/bench_$N.json $N <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('%-18s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f  opt %.2f  frac %.3f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], s['color_optimize'], r['roofline']['frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run p4_a AVDM_SIM_PLANES8=0 AVDM_REFINE_PLANES8=0
run p8_a X=1
for V in p8_pipe3 p8_pipe4 r8_partial; do
  [ -f $ROOT/scripts/ab/$V/libavdm.so ] && run $V AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so
done
run p4_b AVDM_SIM_PLANES8=0 AVDM_REFINE_PLANES8=0
run p8_b X=1
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err > $OUT/bench_final.json; python - $OUT/bench_final.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_with_p2_map','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('cli_end_to_end')); print(r.get('cpu_baseline')); print(r.get('fixed_job'))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
