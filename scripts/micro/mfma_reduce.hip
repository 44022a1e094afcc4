// Dev tool: checks the lane layout assumptions of the MFMA wave reduction used by rasterize_backward_kernel:
// ten per-lane values -> ten wave-wide sums with v_mfma_f32_16x16x4_f32 (A = values, B = column selector; then A = 1, B = row sums).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* in /*[10][64]*/, float* out /*[64]*/) {
    const int lane = threadIdx.x;
    v4f d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        const float sel = (lane & 15) == c ? 1.0f : 0.0f;
        const float g = in[c * 64 + lane];
        if (c & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(g, sel, d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(g, sel, d0, 0, 0, 0);
    }
    const v4f d = d0 + d1;
    const float t = (d.x + d.y) + (d.z + d.w);
    const v4f z = {0, 0, 0, 0};
    const v4f r = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, t, z, 0, 0, 0);
    out[lane] = r.x;
}
int main() {
    float h[640], *di, *dout, ho[64];
    for (int i = 0; i < 640; ++i) h[i] = sinf(i * 0.37f) * (1 + i % 7);
    hipMalloc(&di, sizeof h); hipMalloc(&dout, 256);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 10; ++c) {
        double s = 0; for (int l = 0; l < 64; ++l) s += h[c * 64 + l];
        for (int g = 0; g < 4; ++g) if (fabs(ho[c + 16 * g] - s) > 1e-4 * (1 + fabs(s))) bad++;
        printf("comp %d: want %.5f got %.5f %.5f %.5f %.5f\n", c, s, ho[c], ho[c + 16], ho[c + 32], ho[c + 48]);
    }
    printf(bad ? "LAYOUT WRONG (%d)\n" : "layout ok\n", bad);
    return bad != 0;
}
