#!/bin/bash
# Developer tool (GPU box): libraries against each other inside ONE gpurun call — the headline, the object-centric frame, the converging loop:
#   scripts/ab_libs.sh <variant|default> ...      (brush_amd/variants/libbrush_hip_<variant>.so; `default` = the in-tree build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = default ]; then E="X_AB=1"; else E="BRUSH_HIP_LIB=$GRAFT_REPO_ROOT/brush_amd/variants/libbrush_hip_$lib.so"; fi
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$lib headline', d['ms_per_step'], 'K16', d['stages']['Rasterize']['ms'], 'K17', d['stages']['RasterizeBackwards']['ms'])"
  env $E python bench.py --workload 1m_1080p_centered --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$lib centered', d['ms_per_step'], 'K16', d['stages']['Rasterize']['ms'], 'K17', d['stages']['RasterizeBackwards']['ms'])"
  env $E python bench.py --loop-only exact_lists --loop-steps 1500 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['train_loop']['exact_lists']; print('$lib loop', d['ms_per_step'], 'k16', [s['k16_ms'] for s in d['segments']], 'k17', [s['k17_ms'] for s in d['segments']])"
done; done
