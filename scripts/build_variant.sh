#!/bin/bash
# A/B build of libavdm.so with extra -D flags for ONE source (default avdm_similarity.hip): scripts/ab/<name>/libavdm.so (git-ignored, travels
# to the GPU box; select it with AVDM_LIB).  Prints the register / scratch figures of the default instantiations.
#   usage: scripts/build_variant.sh <name> [-DFOO=1 ...]         (SRC=avdm_sgm.hip scripts/build_variant.sh ... for another source)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-avdm_similarity.hip}
OUT=$ROOT/scripts/ab/$NAME
mkdir -p $OUT
CS=$ROOT/alicevision_amd/csrc
EXTRA=""
case $SRC in avdm_sgm.hip|avdm_maps.hip|avdm_fuse.hip|avdm_literal.hip|avdm_image.hip|avdm_jpeg.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I $ROOT/include $EXTRA "$@" \
   -Rpass-analysis=kernel-resource-usage -c $CS/$SRC -o $OUT/${SRC%.hip}.o 2> $OUT/remarks.txt
OBJS=""
for f in avdm_image avdm_similarity avdm_sgm avdm_maps avdm_fuse avdm_jpeg avdm_literal; do
  if [ -f $OUT/$f.o ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS $CS/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libavdm.so $OBJS
python3 - $OUT/remarks.txt <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for blk in re.split(r'(?=remark: [^\n]*Function Name:)', txt):
    m=re.search(r'Function Name: (\S+)', blk)
    if not m: continue
    n=m.group(1)
    if not any(k in n for k in ('similarity_kernelILb1ELi4ELb0ELi56E','refine_similarity_kernelILb1ELi3ELb1ELi40E','sgm_pair_kernel','optimize_step')): continue
    g=lambda k: re.search(k+r': (\d+)', blk).group(1)
    print(n[:70], 'VGPR',g('VGPRs'),'AGPR',g('AGPRs'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'),'sspill',g('SGPRs Spill'),'vspill',g('VGPRs Spill'))
PY
