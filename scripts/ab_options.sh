#!/bin/bash
# Developer tool (GPU box): A/B of bh_set_option settings inside ONE gpurun call (the boxes of the pool differ by +-5 %).
#   scripts/ab_options.sh "<BH_OPTIONS a>" "<BH_OPTIONS b>" ...      each: the headline's step + K16 / K17, the converging training loop
#   (complete lists, 1500 steps) and the object-centric frame
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for opt in "$@"; do
  BH_OPTIONS=$opt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$opt headline', d['ms_per_step'], 'K16', d['stages']['Rasterize']['ms'], 'K17', d['stages']['RasterizeBackwards']['ms'])"
  BH_OPTIONS=$opt python bench.py --loop-only exact_lists --loop-steps 1500 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['train_loop']['exact_lists']; print('$opt loop', d['ms_per_step'], [s['ms_per_step'] for s in d['segments']], 'k16', [s['k16_ms'] for s in d['segments']], 'k17', [s['k17_ms'] for s in d['segments']])"
  BH_OPTIONS=$opt python scripts/k17_critical_path.py 2>&1 | grep "1m_1080p" | cut -c1-20,100-260
done; done
