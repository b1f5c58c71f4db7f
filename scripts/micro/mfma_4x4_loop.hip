// Does the 4x4x4 fp64 MFMA keep its rate when fed from LDS the way a GEMM inner loop would?
// Per k-step of 4 a wave reads TI x 4 rotated X fragments + TJ matrix fragments (ds_read_b64) and issues TI x TJ x 4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int TI, int TJ, int BKK>
__global__ __launch_bounds__(256) void k_loop(double* out, int iters)
{
    constexpr int LD = BKK + 2;
    __shared__ double sX[2 * 16 * TI * LD + 8];
    __shared__ double sA[2 * 16 * TJ * LD + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * TI * LD; i += 256) sX[i] = 1e-3 * i;
    for (int i = tid; i < 2 * 16 * TJ * LD; i += 256) sA[i] = 1e-3 * i;
    __syncthreads();
    const int fk = lane >> 4;
    const int rowA = lane & 15;
    int rowX[4];
    for (int q = 0; q < 4; ++q) rowX[q] = 4 * ((((lane >> 2) & 3) + q) & 3) + (lane & 3);
    const double* x = sX + (wave >> 1) * 16 * TI * LD;
    const double* a = sA + (wave & 1) * 16 * TJ * LD;
    double acc[TI][TJ][4];
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.0;
    for (int it = 0; it < iters; ++it) {
        // alternate between the two halves of the buffers (as a double-buffered GEMM does): the reads cannot be hoisted
        const double* xb = x + (it & 1) * 16 * TI * LD * 0 + ((it & 1) ? 2 : 0);
        const double* ab = a + ((it & 1) ? 2 : 0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < BKK; ks += 4) {
            double xv[TI][4], av[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[i][q] = xb[(16 * i + rowX[q]) * LD + ks + fk];
#pragma unroll
            for (int j = 0; j < TJ; ++j) av[j] = ab[(16 * j + rowA) * LD + ks + fk];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[i][j][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(xv[i][q], av[j], acc[i][j][q], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    if (s == 12345.678) out[0] = s;
}

template <int CTRL>
__device__ __forceinline__ double row_rotate(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    int lo = (int)bits, hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// the three rotated copies of an X fragment come from DPP row rotations instead of three more LDS reads
template <int TI, int TJ, int BKK>
__global__ __launch_bounds__(256) void k_loop_dpp(double* out, int iters)
{
    constexpr int LD = BKK + 2;
    __shared__ double sX[2 * 16 * TI * LD + 8];
    __shared__ double sA[2 * 16 * TJ * LD + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * TI * LD; i += 256) sX[i] = 1e-3 * i;
    for (int i = tid; i < 2 * 16 * TJ * LD; i += 256) sA[i] = 1e-3 * i;
    __syncthreads();
    const int fk = lane >> 4, row = lane & 15;
    const double* x = sX + (wave >> 1) * 16 * TI * LD;
    const double* a = sA + (wave & 1) * 16 * TJ * LD;
    double acc[TI][TJ][4];
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.0;
    for (int it = 0; it < iters; ++it) {
        const double* xb = x + ((it & 1) ? 2 : 0);
        const double* ab = a + ((it & 1) ? 2 : 0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < BKK; ks += 4) {
            double xv[TI][4], av[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                xv[i][0] = xb[(16 * i + row) * LD + ks + fk];
                xv[i][1] = row_rotate<0x120 + 12>(xv[i][0]);      // row_ror:12 = rotate left by 4 lanes
                xv[i][2] = row_rotate<0x120 + 8>(xv[i][0]);
                xv[i][3] = row_rotate<0x120 + 4>(xv[i][0]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) av[j] = ab[(16 * j + row) * LD + ks + fk];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[i][j][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(xv[i][q], av[j], acc[i][j][q], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    if (s == 12345.678) out[0] = s;
}

template <int TI, int TJ, int BKK, bool DPP>
int run(int blocks)
{
    double* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (DPP) hipLaunchKernelGGL((k_loop_dpp<TI, TJ, BKK>), dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL((k_loop<TI, TJ, BKK>), dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flop = (double)blocks * 4 * iters * (BKK / 4) * TI * TJ * 4 * 512.0;
    printf("%s TI=%d TJ=%d BK=%d blocks=%d: %.3f ms  %.2f TFLOP/s\n", DPP ? "dpp" : "lds", TI, TJ, BKK, blocks, ms, flop / ms / 1e9);
    CK(hipFree(out));
    return 0;
}

int main()
{
    for (int blocks : {256, 512}) {
        if (run<2, 4, 16, false>(blocks)) return 1;
        if (run<4, 4, 16, false>(blocks)) return 1;
        if (run<2, 8, 16, false>(blocks)) return 1;
    }
    return 0;
}
