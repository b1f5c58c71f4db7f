// Achievable issue rate of v_mfma_f64_16x16x4_f64 on gfx950: blocks of 256 threads, each wave runs ITER x NACC
// independent MFMAs.  Prints TFLOP/s for 1, 2 and 4 blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters)
{
    double* out; hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.2f TFLOP/s\n", NACC, blocks, iters, ms, flops / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int bpc : {1, 2, 4}) {
        run<1>(256 * bpc, 20000 / bpc);
        run<2>(256 * bpc, 10000 / bpc);
        run<4>(256 * bpc, 5000 / bpc);
    }
    // long run: sustained clocks
    run<4>(512, 100000);
    return 0;
}
