#!/bin/bash
# First GPU session of round 3 (prepared at the end of round 2, when the GPU budget was spent):
#   1. the full GPU test-suite — includes the two cases added after the last round-2 session (SGM aggregation over the tile buffer extent);
#   2. bench + rocprofv3 kernel trace of the library with the plane-pair SGM similarity kernel (round 2 closed without a trace of it);
#   3. the tiled CLI run with and without AVDM_SGM_BUFFER_EXTENT=1 (host/Sgm.cpp): maps of tiles that start at the image origin must be
#      byte-identical, the others may differ (last paragraph) — the first look at the switch on a GPU;
#   4. PMC passes over the similarity kernels (scripts/pmc_similarity.sh) if time remains: run that script separately.
TAG=${1:-r03_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 600 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=5 > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/log.txt
tail -14 $OUT/pytest.log
echo "== bench" | tee -a $OUT/log.txt
timeout 300 python bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/log.txt
python - <<PY
import json
r=json.load(open("$OUT/bench.json"))
print("value", r["value"], "ms/step", r["ms_per_step"], "frac", r["roofline"]["frac"])
print("stages", {k: round(v,2) for k,v in r["stages_ms"].items() if v > 1})
PY
echo "== rocprofv3 kernel trace (bench)" | tee -a $OUT/log.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace.log 2>&1)
python scripts/rocprof_csv_summary.py $OUT/trace $OUT/kernel_stats.csv >> $OUT/log.txt 2>&1
head -8 $OUT/kernel_stats.csv
echo "== tiled CLI: ROI extent vs buffer extent" | tee -a $OUT/log.txt
timeout 600 python - <<PY 2>&1 | tee -a $OUT/extent.txt
import json, os, subprocess, sys
sys.path.insert(0, "$ROOT")
import numpy as np
from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import make_scene
d = "/tmp/extent_scene"
sc = make_scene(5, 640, 480, seed=5, baseline=0.9, amp=0.6)
sfm, img = scene_io.write_scene(sc, d, n_landmarks=500, compression=0)
cli = os.path.join("$ROOT", "alicevision_amd", "bin", "aliceVision_depthMapEstimation")
base = [cli, "-i", sfm, "--imagesFolder", img, "--downscale", "1", "--rangeStart", "0", "--rangeSize", "1", "--sgmMaxDepths", "64",
        "--tileBufferWidth", "400", "--tileBufferHeight", "300", "--tilePadding", "32", "--exportIntermediateDepthSimMaps", "1", "-v", "warning"]
outs = {}
for flag in ("0", "1"):
    out = os.path.join(d, "out_" + flag)
    r = subprocess.run(base + ["-o", out], env=dict(os.environ, AVDM_SGM_BUFFER_EXTENT=flag), capture_output=True, text=True, timeout=300)
    print("extent", flag, "exit", r.returncode, r.stderr[-300:])
    outs[flag] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out)) if f.endswith(".exr")}
for f in outs["0"]:
    same = outs["0"][f] == outs["1"].get(f)
    print(f, "identical" if same else "DIFFERS")
PY
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
echo "== done" | tee -a $OUT/log.txt
echo "== PMC passes over the similarity kernels (shipped pair kernel)" | tee -a $OUT/log.txt
bash scripts/pmc_similarity.sh $TAG 2>&1 | tail -30
