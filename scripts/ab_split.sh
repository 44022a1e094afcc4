#!/bin/bash
# Developer tool (GPU box): split tiles (option k16_split) against no split and against brush_amd/variants/libbrush_hip_head.so, inside ONE gpurun call:
# the headline's step, the object-centric frame's K16 / K17 (scripts/k17_critical_path.py), the bench's object_centric step.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "head:" "new:k16_split=0" "new:k16_split=250" "$@"; do
  lib=${cfg%%:*}; opt=${cfg#*:}
  if [ $lib = head ]; then E="BRUSH_HIP_LIB=$GRAFT_REPO_ROOT/brush_amd/variants/libbrush_hip_head.so"; else E="X_AB=1"; fi
  env $E BH_OPTIONS="$opt" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$cfg headline', d['ms_per_step'], 'K16', d['stages']['Rasterize']['ms'], 'K17', d['stages']['RasterizeBackwards']['ms'])"
  env $E BH_OPTIONS="$opt" python bench.py --workload 1m_1080p_centered --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$cfg centered', d['ms_per_step'], 'K16', d['stages']['Rasterize']['ms'], 'K17', d['stages']['RasterizeBackwards']['ms'], 'K1', d['stages']['ProjectSplats']['ms'])"
  env $E BH_OPTIONS="$opt" python scripts/k17_critical_path.py 2>&1 | grep "1m_1080p" | cut -c1-20,100-260
done; done
