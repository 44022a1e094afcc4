#!/bin/bash
# Developer tool: A/B one environment knob on the headline bench, alternating runs.   scripts/ab_env.sh BH_NO_LPT [steps] [reps]
KNOB=$1; STEPS=${2:-100}; REPS=${3:-3}
for i in $(seq $REPS); do
  a=$(python bench.py --no-cpu-baseline --steps $STEPS | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  b=$(env $KNOB=1 python bench.py --no-cpu-baseline --steps $STEPS | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
  echo "default $a   $KNOB=1 $b"
done
