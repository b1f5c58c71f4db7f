// Achievable fp64 rates on gfx950 and the shader clock they run at:
//   v_mfma_f64_16x16x4_f64 (2048 flop / wave instruction) and v_fma_f64 (128 flop / wave instruction).
// Shader clock = s_memtime ticks / wall_clock64 ticks (100 MHz) measured inside the kernel by block 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return; } } while (0)

__global__ __launch_bounds__(256) void k_mfma(double* out, long long* clk, int iters, double a0)
{
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void k_fma(double* out, long long* clk, int iters, double a0)
{
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = a0 * (i + threadIdx.x);
    const double m = 1.0 + a0, c = a0;
    long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], m, c);
    }
    long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <typename F>
void run(const char* name, F kernel, int blocks, int iters, double flop_per_wave_iter)
{
    double* out; long long* clk; long long h[2];
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1e-9);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flops = (double)blocks * 4 * iters * flop_per_wave_iter;
    printf("%-6s blocks=%5d: %8.3f ms  %6.2f TFLOP/s   shader clock %.0f MHz (%lld ticks / %lld x 10 ns)\n", name, blocks, ms,
           flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10e-9) / 1e6, h[0], h[1]);
    CK(hipFree(out)); CK(hipFree(clk));
}

// both pipes at once: MFMA waves and FMA waves resident on the same SIMDs (two streams)
void coexec(int mfma_blocks, int mfma_iters, int fma_blocks, int fma_iters)
{
    double* out; long long* clk;
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&clk, 16));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float ms[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {      // 0: mfma alone, 1: fma alone, 2: together
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(hipStreamWaitEvent(s1, e0, 0)); CK(hipStreamWaitEvent(s2, e0, 0));
            if (mode != 1) hipLaunchKernelGGL(k_mfma, dim3(mfma_blocks), dim3(256), 0, s1, out, clk, mfma_iters, 1e-9);
            if (mode != 0) hipLaunchKernelGGL(k_fma, dim3(fma_blocks), dim3(256), 0, s2, out, clk, fma_iters, 1e-9);
            CK(hipEventRecord(e1, s1)); CK(hipEventRecord(e2, s2));
            CK(hipStreamWaitEvent(0, e1, 0)); CK(hipStreamWaitEvent(0, e2, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
    }
    const double tf_m = (double)mfma_blocks * 4 * mfma_iters * 4 * 2048.0, tf_f = (double)fma_blocks * 4 * fma_iters * 8 * 128.0;
    printf("coexec mfma %d blocks + fma %d blocks: mfma alone %.3f ms (%.1f TF), fma alone %.3f ms (%.1f TF), together %.3f ms (%.1f TF total)\n",
           mfma_blocks, fma_blocks, ms[0], tf_m / ms[0] / 1e9, ms[1], tf_f / ms[1] / 1e9, ms[2], (tf_m + tf_f) / ms[2] / 1e9);
}

int main()
{
    for (int bpc : {1, 2, 4, 8}) run("mfma", k_mfma, 256 * bpc, 40000 / bpc, 4 * 2048.0);
    for (int bpc : {1, 2, 4, 8}) run("fma", k_fma, 256 * bpc, 400000 / bpc, 8 * 128.0);
    run("mfma", k_mfma, 512, 400000, 4 * 2048.0);
    run("fma", k_fma, 1024, 2000000, 8 * 128.0);
    coexec(512, 40000, 1024, 200000);
    coexec(512, 40000, 512, 400000);
    coexec(256, 80000, 1024, 200000);
    return 0;
}
