#!/bin/bash
# Developer tool: build a kernel variant of libbrush_hip.so for A/B timing on the GPU box.
#   scripts/ab_build.sh <name> "<extra hipcc flags, e.g. -DBH_BWD_WAVES=5>"
# -> brush_amd/variants/libbrush_hip_<name>.so   (select with BRUSH_HIP_LIB=...)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/brush_amd/variants; mkdir -p $OUT/obj_$NAME
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function $EXTRA"
cd $ROOT/brush_amd/csrc
for f in api project sort depth_sort scan rasterize loss loss_fused optim refine filter3d ply upload comm exchange; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$NAME/*.o -ldl -o $OUT/libbrush_hip_$NAME.so
rm -rf $OUT/obj_$NAME
echo built $OUT/libbrush_hip_$NAME.so
