// How fast can 50 MB be streamed from HBM at all?  A read-and-sum kernel over 8 distinct 50.56 MB buffers (the size of
// a padded 2500^2 fp64 distortion matrix) round-robin, so every launch misses the Infinity Cache: the ceiling the B = 1
// distortion product (k_gemv1) can be compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ p, size_t n2, double* out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0.0;
    for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
        v2d v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { const v2d v = p[i]; s += v.x + v.y; }
    if (s == 1.2345e-300) out[0] = s;
}

int main()
{
    const size_t bytes = (size_t)2500 * 2528 * 8, n2 = bytes / 16;
    std::vector<v2d*> bufs(8);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    double* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {512, 1024, 2048, 4096}) {
        for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, bufs[rep], n2, out);
        CK(hipDeviceSynchronize());
        float total = 0;
        const int reps = 40;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, bufs[rep % 8], n2, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total += ms;
        }
        printf("read 50.56 MB, %4d blocks: %.2f us per launch  %.0f GB/s\n", blocks, total / reps * 1e3, bytes / (total / reps * 1e-3) / 1e9);
    }
    return 0;
}
