#!/bin/bash
# session r04_h: the counter passes of the similarity kernels and the driver's bench command once more, on the final source text (the knife-edge
# functions moved into csrc/avdm_knife.h after r04_g — same ISA, another sha256 — and the bench line only quotes counters stamped with the current one)
cd "$(dirname "$0")/../.."
TAG=${1:-r04_h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/pmc_similarity.sh $TAG 2>&1 | grep -v amdgpu.ids | tail -9
python scripts/collect_sim_pmc.py $TAG > /dev/null && cp profiles/r04_sim_pmc.json $OUT/
timeout 600 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err > $OUT/bench_final.json; python - $OUT/bench_final.json <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); s=r['stages_ms']
print('%.4f maps/s  %.1f ms' % (r['value'], r['ms_per_step'])); print({k: round(v, 3) for k, v in s.items()})
print({k: v for k, v in r['roofline'].items() if k in ('frac','frac_kernels_only','ms_per_launch_by_axis','box_copy_GBps','traffic')})
print(r.get('cli_end_to_end')); print(r['similarity'].get('valu_issue_frac'))
PY
timeout 300 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "parity_table_cfg1 or similarity_volume_parity or library_loaded" 2>&1 | tail -2
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
echo "== done"
