#!/bin/bash
# Developer tool: variant of libbrush_hip.so that differs from the in-tree build in ONE translation unit.
#   scripts/ab_build_one.sh <name> <unit (e.g. rasterize)> "<extra hipcc flags>"   -> brush_amd/variants/libbrush_hip_<name>.so
set -e
NAME=$1; UNIT=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/brush_amd/variants; mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function $EXTRA"
cd $ROOT/brush_amd/csrc
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc $FLAGS -c $UNIT.hip -o $OUT/${UNIT}_$NAME.o
OBJS=$(ls *.o | grep -v "^$UNIT.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/${UNIT}_$NAME.o -ldl -o $OUT/libbrush_hip_$NAME.so
rm -f $OUT/${UNIT}_$NAME.o
echo built $OUT/libbrush_hip_$NAME.so
