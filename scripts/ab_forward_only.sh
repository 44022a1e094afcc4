cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in head default; do
  if [ $lib = default ]; then E="X_AB=1"; else E="BRUSH_HIP_LIB=$GRAFT_REPO_ROOT/brush_amd/variants/libbrush_hip_$lib.so"; fi
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --loop-steps 0 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$lib', d['ms_per_step'], d['forward_only'])"
done; done
