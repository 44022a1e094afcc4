#!/bin/bash
# Developer tool (GPU box): one stage's time for several library variants.   STAGE=MapGaussiansTo scripts/ab_stage.sh default k5nz ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  BRUSH_HIP_LIB=$lib STEPS=${STEPS:-30} timeout 100 python scripts/stage_times.py 2>/dev/null | python -c "
import sys,os
for l in sys.stdin:
    head, _, rest = l.partition('|')
    d = dict(zip(rest.split()[0::2], rest.split()[1::2]))
    print('$v', head.split()[1:5], os.environ.get('STAGE'), d.get(os.environ.get('STAGE')))"
done
