// Developer microbenchmark (not part of the library): how fast does rocPRIM sort the tile-sort workload
// (9.6 M (u32 key, u32 value) pairs on the low 13 key bits) on this GPU?  A yardstick for sort.hip.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/rocprim_sort.hip -o scripts/micro/rocprim_sort && scripts/micro/rocprim_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::atol(argv[1]) : 9624687;
    const unsigned bits = argc > 2 ? std::atoi(argv[2]) : 13;
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (uint32_t)(s % 8160u); hv[i] = (uint32_t)i; }
    uint32_t *k0, *k1, *v0, *v1;
    CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    CK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
    size_t tmp_bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
    void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::printf("rocprim radix_sort_pairs n=%zu bits=%u: %.1f us per sort (temp %zu KB)\n", n, bits, ms * 1000.0f / reps, tmp_bytes >> 10);
    // keys only, for the depth sort comparison
    size_t tb2 = 0; CK(rocprim::radix_sort_pairs(nullptr, tb2, k0, k1, v0, v1, (size_t)1000000, 0, 32));
    void* tmp2; CK(hipMalloc(&tmp2, tb2));
    for (int w = 0; w < 3; ++w) CK(rocprim::radix_sort_pairs(tmp2, tb2, k0, k1, v0, v1, (size_t)1000000, 0, 32));
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) CK(rocprim::radix_sort_pairs(tmp2, tb2, k0, k1, v0, v1, (size_t)1000000, 0, 32));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    std::printf("rocprim radix_sort_pairs n=1000000 bits=32: %.1f us per sort\n", ms * 1000.0f / reps);
    return 0;
}
