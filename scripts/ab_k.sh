#!/bin/bash
# Developer tool (GPU box): stage times of several library variants on the headline and the non-saturating scene.
#   scripts/ab_k.sh default oldk17 r0b0 ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  for wl in 1m_1080p 1m_1080p_lowopac; do
    BRUSH_HIP_LIB=$lib WORKLOAD=$wl STEPS=${STEPS:-30} timeout 120 python scripts/stage_times.py 2>/dev/null | sed "s/^/$v $wl /" | python -c "
import sys,re
for l in sys.stdin:
    head, _, rest = l.partition('|')
    d = dict(zip(rest.split()[0::2], rest.split()[1::2]))
    print(head.strip(), 'K16', d.get('Rasterize'), 'K17', d.get('RasterizeBackw'))"
  done
done
