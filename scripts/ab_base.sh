#!/bin/bash
# Developer tool: alternate the headline bench between a variant library and the in-tree build.  scripts/ab_base.sh <variant> [steps] [reps]
cd "$(dirname "$0")/.."
V=${1:-base}; STEPS=${2:-100}; REPS=${3:-3}
for i in $(seq $REPS); do
  a=$(BRUSH_HIP_LIB=brush_amd/variants/libbrush_hip_$V.so python bench.py --no-cpu-baseline --steps $STEPS | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  b=$(python bench.py --no-cpu-baseline --steps $STEPS | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  echo "$V $a   tree $b"
done
