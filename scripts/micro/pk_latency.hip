// Microbenchmark (dev tool): dependent-issue interval of v_pk_fma_f32 against v_fma_f32 for ONE wave per SIMD (a lone wave at the end of a
// launch) and for several.  One-wave workgroups, W waves per SIMD (grid = 1024 * W), C independent chains per lane.  Always run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int C, bool PK>
__global__ __launch_bounds__(64) void chain(float* out, float a, float b, int iters) {
    float2v x[C];
    const float2v av = {a, a}, bv = {b, b};
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = float2v{(float)threadIdx.x + c, (float)threadIdx.x - c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n" : "+v"(x[c]) : "v"(av), "v"(bv));
                else { asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(x[c].x) : "v"(a), "v"(b)); }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += x[c].x + x[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int C, bool PK>
static void run(float* d, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 1024 * waves_per_simd, iters = 4000;
    hipLaunchKernelGGL((chain<C, PK>), dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<C, PK>), dim3(blocks), dim3(64), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * iters * 16 * C;
    const double per_simd_per_s = winst / (ms * 1e-3) / 1024.0;
    printf("%s waves/SIMD %d chains %d: per wave one inst every %.2f cycles\n", PK ? "v_pk_fma_f32" : "v_fma_f32   ", waves_per_simd, C, 2.4e9 / per_simd_per_s * waves_per_simd);
}
int main() {
    float* d; (void)hipMalloc(&d, 1024 * 8 * 64 * 4);
    for (int w : {1, 2}) { run<1, false>(d, w); run<1, true>(d, w); run<2, false>(d, w); run<2, true>(d, w); run<4, false>(d, w); run<4, true>(d, w); }
    return 0;
}
