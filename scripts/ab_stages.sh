#!/bin/bash
# Developer tool: per-stage times of the headline bench for several library variants.  scripts/ab_stages.sh base nowalk ...
cd "$(dirname "$0")/.."
for v in default "$@"; do
  if [ "$v" = default ]; then lib=""; else lib="brush_amd/variants/libbrush_hip_$v.so"; fi
  BRUSH_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 50 | python -c "
import json,sys
d=json.load(sys.stdin)
print('$v', d['ms_per_step'], ' '.join('%s=%.1f'%(k[:12],v['ms']*1e3) for k,v in d['stages'].items()))"
done
