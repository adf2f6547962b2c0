#!/bin/bash
# What this GPU box delivers for the SGM access shape (the path kernel's duration varies box to box, profiles/README.md): micro-benchmark,
# HBM probe, and the device state rocm-smi / rocminfo report.
cd "$(dirname "$0")/.."
timeout 120 python scripts/sgm_microbench.py 1 8 2>&1 | grep tiles
(cd scripts/probes && [ -x hbm_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o hbm_probe hbm_probe.hip 2>/dev/null; timeout 60 ./hbm_probe 2>&1 | grep -E "float4 copy|column walk 2R 1W nt, 1000|column walk 2R 1W nt, 4000")
rocm-smi --showclocks --showpower --showtemp --showmemuse --showperflevel --showmaxpower --showmemorypartition --showcomputepartition 2>&1 | grep -E "GPU\[0\]" | cut -c1-120
rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Marketing Name|Chip ID|Internal Node ID" | head -8
cat /sys/class/drm/card*/device/current_link_speed 2>/dev/null | head -2
cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1
cat /sys/class/drm/card*/device/vbios_version 2>/dev/null | head -1
