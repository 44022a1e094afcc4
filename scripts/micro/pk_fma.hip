// Microbenchmark: is v_pk_fma_f32 faster per flop than v_fma_f32 on gfx950?  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x[8]; f2 y[4];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) y[i] = f2{x[2*i], x[2*i+1]};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
        } else if (MODE == 2) {  // v_exp_f32
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
        } else {  // v_pk_mul_f32
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = y[i] * a2;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += y[i][0] + y[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* d, double flops_per_elem_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * 64;  // 64 scalar element-ops per iter per lane
    printf("%-14s %.3f ms  %.2f T elem-ops/s  (%.1f TFLOP/s)\n", name, ms, lane_ops / ms / 1e9, lane_ops * flops_per_elem_iter / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d, 2); run<1>("v_pk_fma_f32", d, 2); run<2>("v_exp_f32", d, 1); run<3>("v_pk_mul_f32", d, 1);
    return 0;
}
