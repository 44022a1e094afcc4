#!/bin/bash
# Developer tool (build container): brush_amd/variants/libbrush_hip_trc.so = the in-tree sources + a per-block trace of the forward blend
# (wall-clock start / end, quadrant, tile, listed and walked entries of every block of the LAST launch, read back by bh_debug_k16_trace).
# A measurement build: never the product.   Then on the GPU box:  BRUSH_HIP_LIB=.../libbrush_hip_trc.so python scripts/k16_trace2.py [workload]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d); mkdir -p $TMP/brush_amd $TMP/include $ROOT/brush_amd/variants
cp -r $ROOT/brush_amd/csrc $TMP/brush_amd/; cp $ROOT/include/*.h $TMP/include/; rm -f $TMP/brush_amd/csrc/*.o
python3 - $TMP/brush_amd/csrc/rasterize.hip <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
rep("namespace bh {\n", "namespace bh {\n__device__ unsigned long long g_k16_trace[16384 * 4];\n")
rep("    constexpr int NK = NQ == 4 ? 2 : 1;\n    const uint32_t bidx = blockIdx.x;\n",
    "    constexpr int NK = NQ == 4 ? 2 : 1;\n    const uint32_t bidx = blockIdx.x;\n    const unsigned long long trc_t0 = wall_clock64();\n")
rep("    const bool live_end = any_live();\n    bool saturated = __ballot(live_end) == 0ull;",
    "    if (threadIdx.x == 0 && bidx < 16384u) { g_k16_trace[bidx * 4 + 0] = trc_t0; g_k16_trace[bidx * 4 + 1] = wall_clock64(); "
    "g_k16_trace[bidx * 4 + 2] = ((unsigned long long)(NQ == 1 ? qsel + 1u : 0u) << 32) | local_tile; "
    "g_k16_trace[bidx * 4 + 3] = ((unsigned long long)(range_hi - range_lo) << 32) | (batch_start - range_lo); }\n"
    "    const bool live_end = any_live();\n    bool saturated = __ballot(live_end) == 0ull;")
# K17: per block (start, time of its last look for a job, jobs taken)
rep("namespace bh {\n", "namespace bh {\n__device__ unsigned long long g_k17_trace[32768 * 4];\n__device__ unsigned long long g_k17_hw[32768];\n")
rep("  for (uint32_t jidx = blockIdx.x >> 3;; jidx += gridDim.x >> 3) {",
    "  if (threadIdx.x == 0 && blockIdx.x < 32768u) { uint32_t hwid, xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hwid)); asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc)); g_k17_hw[blockIdx.x] = ((unsigned long long)xcc << 32) | hwid; g_k17_trace[blockIdx.x * 4] = wall_clock64(); g_k17_trace[blockIdx.x * 4 + 2] = 0ull; g_k17_trace[blockIdx.x * 4 + 3] = 0ull; }\n"
    "  for (uint32_t jidx = blockIdx.x >> 3;; jidx += gridDim.x >> 3) {\n"
    "    if (threadIdx.x == 0 && blockIdx.x < 32768u) g_k17_trace[blockIdx.x * 4 + 1] = wall_clock64();")
rep("    int lane = threadIdx.x;\n    if (JOBS) asm volatile",
    "    if (threadIdx.x == 0 && blockIdx.x < 32768u) { g_k17_trace[blockIdx.x * 4 + 2] = 1ull; g_k17_trace[blockIdx.x * 4 + 3] = (unsigned long long)(seg_hi0 - seg_lo0) | ((unsigned long long)seg << 32); }\n"
    "    int lane = threadIdx.x;\n    if (JOBS) asm volatile")
s = s.rstrip() + "\nextern \"C\" int bh_debug_k17_hw(void* out, unsigned long long bytes) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bh::g_k17_hw), bytes < sizeof(bh::g_k17_hw) ? bytes : sizeof(bh::g_k17_hw)); }\n"
s = s.rstrip() + "\nextern \"C\" int bh_debug_k17_trace(void* out, unsigned long long bytes) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bh::g_k17_trace), bytes < sizeof(bh::g_k17_trace) ? bytes : sizeof(bh::g_k17_trace)); }\n"
s = s.rstrip() + "\nextern \"C\" int bh_debug_k16_trace(void* out, unsigned long long bytes) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bh::g_k16_trace), bytes < sizeof(bh::g_k16_trace) ? bytes : sizeof(bh::g_k16_trace)); }\n"
open(p, "w").write(s)
PY
cd $TMP/brush_amd/csrc && make -j8 ../libbrush_hip.so > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -S --cuda-device-only rasterize.hip -o /tmp/rasterize_trc.s 2>/dev/null || true
cp $TMP/brush_amd/libbrush_hip.so $ROOT/brush_amd/variants/libbrush_hip_trc.so; rm -rf $TMP
echo built $ROOT/brush_amd/variants/libbrush_hip_trc.so
