#!/bin/bash
# GPU box: per-kernel averages (rocprofv3 --kernel-trace --stats) of the headline bench for the in-tree build and variants.
#   scripts/kstat_variants.sh <grep pattern> default <variant | ENV=VALUE> ...
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  envv="X_KS_DEFAULT=1"
  case "$v" in default) lib="" ;; *=*) lib=""; envv="$v" ;; *) lib="$GRAFT_REPO_ROOT/brush_amd/variants/libbrush_hip_$v.so" ;; esac
  rm -rf /tmp/ks_$v
  env $envv BRUSH_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-extra --no-pmc ${BENCH_ARGS} > /tmp/ks_$v.json 2>/dev/null
  echo "== $v  $(python -c "import json; d=json.load(open('/tmp/ks_$v.json')); print(d['ms_per_step'])")"
  python $GRAFT_REPO_ROOT/scripts/kstats.py $(find /tmp/ks_$v -name "*kernel_stats.csv") | grep -E "$PAT"
done
