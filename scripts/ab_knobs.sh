#!/bin/bash
# Developer tool: per-stage times of the headline bench under several env settings.  scripts/ab_knobs.sh "BH_SORT_KPT=4" "BH_SORT_KPT=8"
cd "$(dirname "$0")/.."
for v in "X_DEFAULT=1" "$@"; do
  env $v python bench.py --no-cpu-baseline --steps 50 | python -c "
import json,sys
d=json.load(sys.stdin)
print('$v', d['ms_per_step'], ' '.join('%s=%.1f'%(k[:12],v['ms']*1e3) for k,v in d['stages'].items()))"
done
