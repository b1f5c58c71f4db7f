// Issue rate of v_mfma_f64_4x4x4_4b_f64 (512 flop / wave instruction) next to the 16x16x4 form (2048 flop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_4x4(double* out, int iters, double a0)
{
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void k_16x16(double* out, int iters, double a0)
{
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

int main()
{
    double* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which)
        for (int blocks : {256, 512, 1024, 2048}) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k_4x4, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-9);
                else hipLaunchKernelGGL(k_16x16, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-9);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)blocks * 4 * iters * (which == 0 ? 8 * 512.0 : 4 * 2048.0);
            printf("%s blocks=%d: %.3f ms  %.2f TFLOP/s\n", which == 0 ? "mfma_f64_4x4x4 " : "mfma_f64_16x16x4", blocks, ms, flop / ms / 1e9);
        }
    return 0;
}
