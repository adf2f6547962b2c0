#!/bin/bash
# round 2: the two-worker CLI run (AVDM_FAKE_DEVICES=2 --nbGPUs 2) hung once in session r02_m.  Repeat it under a watchdog; a run that
# exceeds LIMIT seconds gets its thread backtraces dumped by rocgdb before it is killed (exact PID, never by pattern).
TAG=${1:-r02_hang}
RUNS=${2:-8}
LIMIT=${3:-45}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$(pwd)
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
D=/tmp/hang_scene
timeout 300 python - <<PY || exit 1
import json, os
from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import make_scene
d = "$D"
sc = make_scene(5, 640, 480, seed=5, baseline=0.9, amp=0.6)
lms = scene_io.sample_landmarks(sc, 500, amp=0.6)
os.makedirs(os.path.join(d, "images"), exist_ok=True)
with open(os.path.join(d, "scene.sfm"), "w") as f:
    json.dump(scene_io.sfm_dict(sc, lms, os.path.join(d, "images")), f)
for i in range(5):
    im = sc.images[i].numpy()
    exr_io.write_exr(os.path.join(d, "images", "%d.exr" % scene_io.view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]}, compression=0)
print("scene written")
PY
CLI=$ROOT/alicevision_amd/bin/aliceVision_depthMapEstimation
ARGS="-i $D/scene.sfm --imagesFolder $D/images --downscale 1 --rangeStart 0 --rangeSize 4 --sgmMaxDepths 64 --colorOptimizationNbIterations 5 --tileBufferWidth 400 --tileBufferHeight 300 --tilePadding 32 -v info"
hung=0
for i in $(seq 1 $RUNS); do
    t0=$(date +%s.%N)
    AVDM_FAKE_DEVICES=2 $EXTRA_ENV $CLI $ARGS -o $D/out_$i --nbGPUs 2 > $OUT/run_$i.log 2>&1 &
    pid=$!
    waited=0
    while kill -0 $pid 2>/dev/null && [ $waited -lt $LIMIT ]; do sleep 1; waited=$((waited + 1)); done
    if kill -0 $pid 2>/dev/null; then
        hung=$((hung + 1))
        echo "run $i: HUNG after $LIMIT s (pid $pid)" | tee -a $OUT/log.txt
        timeout 120 rocgdb -p $pid -batch -ex "set pagination off" -ex "thread apply all bt 40" > $OUT/bt_$i.txt 2>&1
        kill -9 $pid
        wait $pid 2>/dev/null
        [ $hung -ge 2 ] && break
    else
        wait $pid
        rc=$?
        t1=$(date +%s.%N)
        echo "run $i: exit $rc in $(python -c "print(round($t1 - $t0, 1))") s" | tee -a $OUT/log.txt
        [ $i -gt 2 ] && rm -f $OUT/run_$i.log
    fi
done
echo "hung $hung of $i runs" | tee -a $OUT/log.txt
for f in $OUT/bt_*.txt; do [ -f "$f" ] && grep -c "^Thread" $f && grep -n "^#[0-9 ]" $f | head -150; done
echo "== done"
