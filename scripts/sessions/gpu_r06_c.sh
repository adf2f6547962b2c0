#!/bin/bash
# session r06_c: after pack_h4 was fenced against v_fma_mixlo_f16 — the pyramid bit for bit, the fast parity-table cases with the parity-mode
# assertions, the CLI's flags; then the similarity tweaks A/B (fast sigmoid, outlier kernel's unroll) on one box
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()))" || { echo "GPU sanity check failed"; exit 1; }
echo "== pyramid + stage tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "pyramid_parity or reference_arithmetic or texture_unit or similarity_volume_parity or refine_volume_parity or refine_outlier or experiment_equals" 2>&1 | tail -12 | cut -c1-400
echo "== parity-table tests (fast cases)"
AVDM_PARITY_DUMP=$ROOT/$OUT timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "parity_table_cfg1 or (parity_of_default_tiles and corner) or real_shape" 2>&1 | grep -E "^E  |passed|failed|FAILED" | cut -c1-700 | tail -30
python - $OUT <<'PY'
import json,glob,sys
for p in sorted(glob.glob(sys.argv[1]+'/parity_*.json')):
    r=json.load(open(p))
    print('--',r['case'],r.get('pyramid_texels_differing'))
    for k in ('literal','reference_arithmetic_sgm_vs_oracle_literal','reference_arithmetic_all_vs_oracle_literal'):
        b=r.get(k)
        if not b: continue
        fd=b['final_depth']
        print('  %-44s rmse %.3e max %.4f | vol0 %.5f filt0 %.5f wta %.1e | refvol %s | refined max %s | sim %s' % (k, fd['rmse_untrimmed'], fd['max_abs'], b['similarity_volume_levels']['0'], b['sgm_filtered_volume_levels']['0'], b['sgm_wta_depth_differs'], {kk: round(v,5) for kk,v in b['refine_volume_abs'].items()}, (b.get('refined_depth') or {}).get('max_abs'), {kk: round(v,4) for kk,v in (b.get('final_sim') or {}).items()}))
PY
echo "== the program's flags"
timeout 600 python -m pytest tests/test_host_cli_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "reference_arithmetic" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | tail -8
show() { python - $1 $2 <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']; f=r['roofline']
    print('%-12s %.4f maps/s %.1f ms | sgm_sim %.1f refine_sim %.1f color %.2f rbd %.2f | frac %.4f kernels %.4f | refine each %s' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], s['color_optimize'], s.get('refine_best_depth',0), f['frac'], f['frac_kernels_only'], [round(x) for x in r['similarity_ms_each']['refine_similarity']]))
except Exception as e:
    print(sys.argv[2],'FAILED',e)
PY
}
echo "== bench A/B: tree = fast sigmoid + outlier kernel unrolled by 7 + R axis projected once"
for i in 1 2; do
  for V in tree perf_off outl4 sigm_off; do
    L=$ROOT/scripts/ab/$V/libavdm.so; [ $V = tree ] && L=$ROOT/alicevision_amd/csrc/libavdm.so
    AVDM_LIB=$L timeout 300 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_${V}_$i.err > $OUT/bench_${V}_$i.json; show $OUT/bench_${V}_$i.json ${V}_$i
  done
done
echo "== done"
