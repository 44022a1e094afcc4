#!/bin/bash
# Developer tool (build container + GPU box): A/B kernel variants of libbrush_hip.so.  One script, four verbs:
#
#   scripts/ab.sh build <name> "<extra hipcc flags>"            every translation unit with the flags  (e.g. -DBH_BWD_WAVES=5)
#   scripts/ab.sh build-one <name> <unit> "<extra hipcc flags>"  only <unit>.hip differs from the in-tree build (e.g. rasterize)
#   scripts/ab.sh build-probe <name> "<extra hipcc flags>"       a MEASUREMENT build: the probe sites (-DBH_K16_PROBE=.., -DBH_NO_ATOMIC, ...; wrong
#        results by design) are NOT in the shipping sources — probes/probe_sites.patch re-inserts them into a scratch copy, which is built
#        -> brush_amd/variants/libbrush_hip_<name>.so  (git-ignored; travels to the GPU box; selected with BRUSH_HIP_LIB=...)
#   scripts/ab.sh stages <variant|default|ENV=VALUE> ...          per-stage times of the headline bench (50 steps) for each argument:
#        a variant name, `default` (the in-tree build) or an environment setting such as BH_SORT_KPT=8 / BH_NO_LPT=1
#   scripts/ab.sh alternate <variant|ENV=VALUE> [steps] [reps]    ms_per_step, alternating the argument with the in-tree default
#        (boxes of the pool differ by +-10 %: always compare inside ONE gpurun call)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function"
UNITS="api project sort depth_sort scan rasterize loss loss_fused optim refine filter3d ply upload comm exchange"
OUT=$ROOT/brush_amd/variants
verb=$1; shift || true
setup() {   # $1 = variant | default | ENV=VALUE  -> ENVV (an `env` argument)
  case "$1" in
    default) ENVV="X_AB_DEFAULT=1" ;;
    *=*) ENVV="$1" ;;
    *) ENVV="BRUSH_HIP_LIB=$OUT/libbrush_hip_$1.so" ;;
  esac
}
case "$verb" in
  build)
    NAME=$1; EXTRA=$2; mkdir -p $OUT/obj_$NAME; cd $ROOT/brush_amd/csrc
    for f in $UNITS; do /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o $OUT/obj_$NAME/$f.o & done; wait
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$NAME/*.o -ldl -o $OUT/libbrush_hip_$NAME.so
    rm -rf $OUT/obj_$NAME; echo built $OUT/libbrush_hip_$NAME.so ;;
  build-probe)
    NAME=$1; EXTRA=$2; TMP=$(mktemp -d); mkdir -p $TMP/brush_amd $TMP/include $OUT
    cp -r $ROOT/brush_amd/csrc $TMP/brush_amd/; cp $ROOT/include/*.h $TMP/include/; rm -f $TMP/brush_amd/csrc/*.o
    (cd $TMP && patch -p1 -s -F3 < $ROOT/probes/probe_sites.patch) || { echo "probes/probe_sites.patch no longer applies: rebase it on the current kernels"; exit 1; }
    cd $TMP/brush_amd/csrc
    for f in $UNITS; do /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o $f.o & done; wait
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -ldl -o $OUT/libbrush_hip_$NAME.so
    rm -rf $TMP; echo built $OUT/libbrush_hip_$NAME.so ;;
  build-one)
    NAME=$1; UNIT=$2; EXTRA=$3; mkdir -p $OUT; cd $ROOT/brush_amd/csrc; make -s -j8 >/dev/null
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $UNIT.hip -o $OUT/${UNIT}_$NAME.o
    OBJS=$(ls *.o | grep -v "^$UNIT.o$" | grep -v "\.th\.o$")   # (the test-hook objects belong to the other library)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/${UNIT}_$NAME.o -ldl -o $OUT/libbrush_hip_$NAME.so
    rm -f $OUT/${UNIT}_$NAME.o; echo built $OUT/libbrush_hip_$NAME.so ;;
  stages)
    cd $ROOT
    for v in "$@"; do setup "$v"
      env $ENVV python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-50} ${BENCH_ARGS} | python -c "
import json,sys
d=json.load(sys.stdin)
print('$v', d['ms_per_step'], ' '.join('%s=%.1f'%(k[:12],v['ms']*1e3) for k,v in d['stages'].items()))"
    done ;;
  alternate)
    cd $ROOT; setup "$1"; STEPS=${2:-100}; REPS=${3:-3}
    for i in $(seq $REPS); do
      a=$(env $ENVV python bench.py --no-cpu-baseline --no-extra --steps $STEPS ${BENCH_ARGS} | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
      b=$(python bench.py --no-cpu-baseline --no-extra --steps $STEPS ${BENCH_ARGS} | python -c "import json,sys; print(json.load(sys.stdin)['ms_per_step'])")
      echo "$1 $a   default $b"
    done ;;
  *) sed -n 2,12p "$0"; exit 1 ;;
esac
