#!/bin/bash
# session r05_b: (1) the whole GPU suite in xdist workers once more, with the workers' OpenMP teams sized to the cores this time (r05_a: 15 x
# slower oracle under 2 x oversubscription, the suite ran into its time limit) — every parity measurement dumped; (2) A/B of the Refine outlier
# list (AVDM_REFINE_OUTLIER_LIST=0 = the round-4 wave fall-back) on the library with the eight-plane pass on partial chunks; (3) the driver's command.
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r05_b}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== the whole GPU suite (xdist, $(nproc) cores)"
T0=$(date +%s)
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 840 python -m pytest tests -m gpu -v --no-header -rP --durations=25 > $OUT/pytest.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - T0 )) s"
grep -E "passed|failed|^FAILED|^ERROR|^E   |outlier list:" $OUT/pytest.log | cut -c1-600 | tail -40
grep -E "^[0-9.]+s (call|setup)" $OUT/pytest.log | head -25
python - $OUT <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "parity_*.json"))):
    r = json.load(open(f))
    fd = lambda m: (m["final_depth"]["rmse_untrimmed"], m["final_depth"].get("rmse_untrimmed_relative"), m["final_depth"].get("rmse_untrimmed_in_pixsize"))
    line = "%-18s wp %.2e rel %.1e pix %.2f | lit %.2e rel %.1e pix %.2f | gpu-lit %.2e | vol0 wp %.3f lit %.3f | t %.0f %.0f" % ((r["case"],) + fd(r["well_posed"]) + fd(r["literal"]) + (r["gpu_literal_vs_oracle_literal"]["final_depth"]["rmse_untrimmed"], r["well_posed"]["similarity_volume_levels"]["0"], r["literal"]["similarity_volume_levels"]["0"], r["well_posed"]["t_oracle_s"], r["literal"]["t_oracle_s"]))
    sp = r.get("platform_spread")
    if sp:
        for k in ("cuda_vs_literal", "cuda_vs_literal_interior", "default_vs_literal_interior", "default_vs_cuda_interior", "well_posed_vs_literal_interior"):
            if k in sp:
                line += " %s %.2e" % (k.replace("_interior", "_in").replace("_vs_", "/"), sp[k]["final_depth"]["rmse_untrimmed"])
        line += " t %.0f" % sp["t_s"]
    print(line)
PY
echo "== A/B bench: Refine outlier list"
run() { # name, env...
  N=$1; shift
  env "$@" timeout 200 python bench.py --steps 11 --warmup 2 --no-cpu-baseline --cli-e2e 0 2>/dev/null > $OUT/bench_$N.json
  python - $OUT/bench_$N.json $N <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']
    print('%-12s %.4f maps/s  %.1f ms  sgm_sim %.1f  refine_sim %.1f  refine each %s' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], r['similarity_ms_each']['refine_similarity']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run list0_a AVDM_REFINE_OUTLIER_LIST=0
run list1_a X=1
run list0_b AVDM_REFINE_OUTLIER_LIST=0
run list1_b X=1
AVDM_REFINE_OUTLIER_STATS=1 timeout 100 python - <<'PY' 2>/dev/null
# how many units the list holds per T camera on the bench's scene (three reference cameras)
import ctypes, json, subprocess, sys, os
sys.argv = ["bench.py", "--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--cli-e2e", "0"]
import runpy
from alicevision_amd import abi
lib = abi.load(); lib.avdm_debug_refine_outlier_units.argtypes = [ctypes.POINTER(ctypes.c_uint)]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
u = (ctypes.c_uint * 2)(); lib.avdm_debug_refine_outlier_units(u)
print("outlier units over 3 depth maps x 10 T cameras: %d worked off, %d refused (capacity %d per launch)" % (u[0], u[1], 4000 * 3000 * 4 // 4))
PY
echo "== bench (the driver's command)"
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err > $OUT/bench_final.json; python - $OUT/bench_final.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','frac_with_p2_map','ms_per_launch_by_axis','box_copy_GBps')})
print(r.get('cli_end_to_end'))
PY
echo "== done"
