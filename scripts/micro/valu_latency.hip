// Microbenchmark (dev tool): how many waves per SIMD / independent chains per wave does it take to saturate the gfx950
// VALU with DEPENDENT f32 ops?  One-wave workgroups, W waves per SIMD (grid = 1024 * W), C independent fma chains per lane.
// Prints cycles per wave-instruction per SIMD at the 2.4 GHz nominal clock.  Always run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ __launch_bounds__(64) void chain(float* out, float a, float b, int iters) {
    float x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = threadIdx.x + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(x[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int C>
static void run(float* d, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 1024 * waves_per_simd, iters = 4000;
    hipLaunchKernelGGL(chain<C>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<C>, dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * 16 * C;
    const double per_simd_per_s = winst / (ms * 1e-3) / 1024.0;
    printf("waves/SIMD %d chains %d: %.2f cycles/inst/SIMD @2.4GHz   (per wave: one inst every %.2f cycles)\n", waves_per_simd, C, 2.4e9 / per_simd_per_s,
           2.4e9 / per_simd_per_s * waves_per_simd);
}
int main() {
    float* d; (void)hipMalloc(&d, 1024 * 8 * 64 * 4);
    for (int w : {1, 2, 3, 4, 6, 8}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); }
    return 0;
}
