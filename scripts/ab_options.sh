cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for opt in "lpt_classes=linear" "lpt_classes=log"; do
  BH_OPTIONS=$opt python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$opt headline', d['ms_per_step'], 'K17', d['stages']['RasterizeBackwards']['ms'])"
  BH_OPTIONS=$opt python bench.py --loop-only exact_lists 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['train_loop']['exact_lists']; print('$opt loop', d['ms_per_step'], [s['ms_per_step'] for s in d['segments']], [s['k17_ms'] for s in d['segments']])"
done; done
