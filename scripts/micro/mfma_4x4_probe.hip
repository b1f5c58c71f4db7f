// Lane layout of v_mfma_f64_4x4x4_4b_f64 (A, B, C operands; CBSZ / ABID broadcast), found by one-hot probing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int CBSZ, int ABID>
__global__ void k_probe(double* out)      // block = (la, lb): A one-hot at lane la, B one-hot at lane lb
{
    const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    out[((size_t)la * 64 + lb) * 64 + lane] = c;
}

template <int CBSZ, int ABID>
int run(const char* label)
{
    double* d; CK(hipMalloc(&d, sizeof(double) * 64 * 64 * 64));
    hipLaunchKernelGGL((k_probe<CBSZ, ABID>), dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    printf("== %s\n", label);
    // for each la: which lb give a non-zero, and where
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int lc = 0; lc < 64; ++lc)
                if (h[((size_t)la * 64 + lb) * 64 + lc] != 0.0) printf(" (B%d->C%d)", lb, lc);
        printf("\n");
    }
    CK(hipFree(d));
    return 0;
}

int main()
{
    if (run<0, 0>("cbsz=0 abid=0")) return 1;
    if (run<2, 1>("cbsz=2 abid=1")) return 1;
    return 0;
}
