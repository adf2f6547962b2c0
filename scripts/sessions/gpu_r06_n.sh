#!/bin/bash
# session r06_n: where the outer cameras' extra Refine time goes — rocprofv3 kernel trace of the bench over ALL 11 cameras, per dispatch
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_n}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
(cd /tmp && AVDM_REFINE_OUTLIER_STATS=1 timeout 500 rocprofv3 --kernel-trace -f csv -d $ROOT/$OUT/trace -o kt -- python $ROOT/bench.py --steps 11 --warmup 0 --no-cpu-baseline --cli-e2e 0 --no-parity-mode-cost > $ROOT/$OUT/trace.log 2>&1)
tail -1 $OUT/trace.log | cut -c1-300
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - $F <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# split into steps at rgba_f32_to_f16x255 (the R image's pyramid opens a step)
steps=[];cur=None
for r in rows:
    n=r['Kernel_Name']
    if 'rgba_f32_to_f16x255' in n:
        cur=collections.defaultdict(lambda:[0,0.0]); steps.append(cur)
    if cur is None: continue
    short=n.split('(')[0].split('<')[0].replace('void ','').replace('avdm::','').replace('(anonymous namespace)::','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-6
    cur[short][0]+=1; cur[short][1]+=d
print(len(steps),'segments opened by the pyramid conversion')
keys=['similarity_kernel','refine_similarity_kernel','refine_outlier_kernel','optimize_step_points_kernel','refine_best_depth_kernel']
for i,s in enumerate(steps[-11:]):
    print(i,' '.join('%s %d x %.2f ms'%(k.replace('_kernel',''),s[k][0],s[k][1]) for k in keys if k in s))
PY
rm -rf $OUT/trace/*/*agent_info.csv
echo "== done"
