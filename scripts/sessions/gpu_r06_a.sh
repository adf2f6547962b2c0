#!/bin/bash
# session r06_a: (1) the Lab pyramid against the oracle's, texel for texel (glibc's cbrtf on the device, no contraction); (2) the product's
# reference-arithmetic mode (avdm_sgm_params_t / avdm_refine_params_t::referenceArithmetic) against the literal oracle on cfg1, crop3 and the
# three tile cases that sat at / over BASELINE's bar, from the GPU's own pyramids; (3) what the mode costs on the bench; (4) the SGM
# aggregation call: per-launch events on / off, the call's span on the device clock, 2 / 3 / 4 columns per workgroup
cd "$(dirname "$0")/../.."
ROOT=$(pwd)
TAG=${1:-r06_a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 24, device='cuda'); print('gpu sanity', float(x.sum()), torch.cuda.get_device_name(0))" || { echo "GPU sanity check failed"; exit 1; }
nproc
echo "== pyramid / texture / literal tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "pyramid_parity or texture_unit or camera_fill or fractional" 2>&1 | tail -15 | cut -c1-300
echo "== parity: cfg1, crop3 (default, literal kernel on the oracle's pyramids, reference-arithmetic modes on the GPU's own pyramids)"
timeout 900 python scripts/parity_report.py --cases cfg1,crop3 --literal-cases cfg1,crop3 --ref-cases none --strict sgm,all --out $OUT/parity_small.json > $OUT/parity_small.log 2>&1
echo "== parity: the three tile cases + the interior 12 MP tile, literal oracle only"
timeout 1500 python scripts/parity_report.py --cases tile12mp_corner,tile24mp_interior,tile24mp_corner,tile12mp_interior --literal-cases none --ref-cases none --modes literal --strict sgm,all --out $OUT/parity_tiles.json > $OUT/parity_tiles.log 2>&1
python - $OUT/parity_small.json $OUT/parity_tiles.json <<'PY'
import json,sys
for p in sys.argv[1:]:
    try:
        rs=json.load(open(p))
    except Exception as e:
        print(p,'MISSING',e); continue
    for r in rs:
        print('--',r['case'],'pyramid texels differing',r.get('pyramid_texels_differing'))
        for k in ('well_posed','literal','reference_arithmetic_sgm_vs_oracle_literal','reference_arithmetic_all_vs_oracle_literal','gpu_literal_vs_oracle_literal'):
            if k in r and r[k]:
                b=r[k]; fd=b['final_depth']
                print('   %-46s rmse %.3e  best99.5 %.2e  max %.3f | volume identical %.4f  filtered identical %.4f  wta differs %.2e | refvol %s | sim %s | t %.0f s' % (
                    k, fd['rmse_untrimmed'], fd['rmse_best_99.5pct'], fd['max_abs'], b['similarity_volume_levels']['0'], b['sgm_filtered_volume_levels']['0'], b['sgm_wta_depth_differs'],
                    b.get('refine_volume_abs'), (b.get('final_sim') or {}).get('identical_halfs'), b.get('t_s', b.get('t_oracle_s', 0))))
PY
tail -3 $OUT/parity_small.log | cut -c1-300; tail -3 $OUT/parity_tiles.log | cut -c1-300
show() { python - $1 $2 <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); s=r['stages_ms']; f=r['roofline']
    print('%-14s %.4f maps/s %.1f ms | sgm_sim %.1f refine_sim %.1f sgm_opt %.4f color %.2f rbd %.2f | frac %.4f kernels %.4f span %s by_axis %s copy %.0f' % (sys.argv[2], r['value'], r['ms_per_step'], s['sgm_similarity'], s['refine_similarity'], s['sgm_optimize'], s['color_optimize'], s.get('refine_best_depth',0), f['frac'], f['frac_kernels_only'], f.get('frac_call_span'), f.get('ms_per_launch_by_axis'), f.get('box_copy_GBps',0)))
except Exception as e:
    print(sys.argv[2],'FAILED',e)
PY
}
echo "== bench: default, ABAB with the per-launch events off"
for i in 1 2; do
  timeout 300 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_default_$i.err > $OUT/bench_default_$i.json; show $OUT/bench_default_$i.json default_$i
  AVDM_BENCH_KERNEL_EVENTS=0 timeout 300 python bench.py --steps 11 --warmup 3 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_noev_$i.err > $OUT/bench_noev_$i.json; show $OUT/bench_noev_$i.json noevents_$i
done
echo "== bench: SGM columns per workgroup (fast builds: 256 planes only)"
for V in sgm_wpb4 sgm_wpb3 sgm_wpb2; do
  AVDM_LIB=$ROOT/scripts/ab/$V/libavdm.so timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --cli-e2e 0 2> $OUT/bench_$V.err > $OUT/bench_$V.json; show $OUT/bench_$V.json $V
done
echo "== bench: the reference-arithmetic mode's cost"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cli-e2e 0 --reference-arithmetic sgm 2> $OUT/bench_strict_sgm.err > $OUT/bench_strict_sgm.json; show $OUT/bench_strict_sgm.json strict_sgm
AVDM_STRICT_WINDOWS=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --cli-e2e 0 --reference-arithmetic sgm 2> $OUT/bench_strict_sgm_nowin.err > $OUT/bench_strict_sgm_nowin.json; show $OUT/bench_strict_sgm_nowin.json strict_sgm_global_taps
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --cli-e2e 0 --reference-arithmetic all 2> $OUT/bench_strict_all.err > $OUT/bench_strict_all.json; show $OUT/bench_strict_all.json strict_all
echo "== done"
