// Fit kernels: the MIGRAD state machines of vmx_migrad.h advanced where the walkers live.
//
// A round of the fit driver (vmx_fit_migrad, vegamx.hip) is
//     k_fit_advance   one thread per fit: consume the chi2 values of the fit's last request, run Minuit's bookkeeping up to the
//                     next request, leave the number of points it asks for
//     k_fit_scan      one block: exclusive scan of those counts -> the fits' row offsets in the round's batch, the total for the host
//     k_fit_emit      one block per fit: its requested points as parameter rows (the fit's base row with the free columns set to
//                     the external values of the internal point, Minuit's transforms in the kernel) + the walker's mock row
// followed by the engine's own chain over the rows; chi2 stays in HBM and is read by the next k_fit_advance.  The host sees one
// integer per round.  (Reference semantics: vega/minimizer.py:66-97 per fit, vega/analysis.py:224-308 per mock - one MIGRAD
// after the other there.)
//
// A thread that walks its fit's state in global memory waits ~0.5 us for every dependent access - Minuit's bookkeeping is a
// chain of them (230 us per round measured, most of it the Jacobi sweeps of MnPosDef).  The state is therefore a template on
// its capacity N (free parameters per Minuit object: 4, 8, 16 or 32, the smallest that holds the stages), a block copies its
// fits' states into LDS with all its threads (coalesced), every thread advances its fit there, and the block copies them back.
#pragma once
#include <hip/hip_runtime.h>

#include "vmx_migrad.h"

struct FitDev {
    void* state;                        // [F] vmx_migrad::FitStateT<N>
    const vmx_migrad::Spec* spec;
    double* base;                       // [F][P] the fits' parameter rows (fixed columns, start values, a finished stage's result)
    const int32_t* mock_row;            // [F] pool row a fit's walkers are compared with (nullptr: the items' own data)
    vmx_migrad::StageOut out[vmx_migrad::MAX_STAGES];
    int32_t* count;                     // [F] points of the current request
    int32_t* offset;                    // [F + 1] row offsets of the current round (exclusive scan of count)
    int32_t* done;                      // [F]
    double* theta;                      // [cap][P] the round's rows
    int32_t* mock;                      // [cap]
    const double* chi2;                 // [cap] values of the previous round's rows
    volatile int32_t* host_word;        // mapped host memory: [0] rows of the round, [1] fits still running
    int32_t F, P, admitted;             // fits, parameter columns, fits [0, admitted) may run
};

// stride (in doubles) of a fit's state in LDS: odd, so that the threads' accesses to the same member fall on different banks
template <int N>
constexpr int fit_lds_stride() { return (int)((sizeof(vmx_migrad::FitStateT<N>) + 7) / 8) | 1; }

// T fits per block of FIT_THREADS threads: all threads copy the states, thread t < T advances fit blockIdx.x * T + t.
// T = 1 unless the states are tiny: the threads of a wave are in different phases of their fits, a wave executes the UNION of its
// threads' paths, and this kernel is 78 KB of code fetched cold (2.5 ns per instruction) - 8 fits per wave
// took 174 us per round, one fit per wave takes what its own path costs.
constexpr int FIT_THREADS = 64;
template <int N, int T>
__global__ __launch_bounds__(FIT_THREADS) void k_fit_advance(FitDev D)
{
    extern __shared__ double fit_state_lds[];
    using State = vmx_migrad::FitStateT<N>;
    constexpr int WORDS = (int)(sizeof(State) / 8), STRIDE = fit_lds_stride<N>();
    static_assert(sizeof(State) % 8 == 0, "fit state: whole doubles");
    const int f0 = blockIdx.x * T, t = threadIdx.x;
    const int live = min(T, D.F - f0);
    // (a fit that is done, or not yet admitted, needs no copy: its thread leaves 0 points)
    bool any = false;
    for (int q = 0; q < live; ++q) any = any || (f0 + q < D.admitted && !D.done[f0 + q]);
    if (!any) { if (t < live) D.count[f0 + t] = 0; return; }
    double* g = (double*)D.state + (size_t)f0 * WORDS;
    for (int w = t; w < live * WORDS; w += FIT_THREADS) fit_state_lds[(w / WORDS) * STRIDE + (w % WORDS)] = g[w];
    __syncthreads();
    const int f = f0 + t;
    if (t < live) {
        State& s = *(State*)(fit_state_lds + t * STRIDE);
        int c = 0;
        if (f < D.admitted && !s.done)
            c = vmx_migrad::advance<N>(s, *D.spec, D.chi2 + D.offset[f], D.base + (size_t)f * D.P, D.out, f);
        D.count[f] = c;
        D.done[f] = s.done;
    }
    __syncthreads();
    for (int w = t; w < live * WORDS; w += FIT_THREADS) g[w] = fit_state_lds[(w / WORDS) * STRIDE + (w % WORDS)];
}

__global__ __launch_bounds__(1024) void k_fit_scan(FitDev D)
{
    __shared__ int part[1024];
    __shared__ int running[1024];
    const int t = threadIdx.x;
    const int per = (D.F + 1023) / 1024;
    const int lo = t * per, hi = min(D.F, lo + per);
    int sum = 0, alive = 0;
    for (int f = lo; f < hi; ++f) { sum += D.count[f]; alive += (f >= D.admitted || !D.done[f]) ? 1 : 0; }
    part[t] = sum; running[t] = alive;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0, w = t >= d ? running[t - d] : 0;
        __syncthreads();
        part[t] += v; running[t] += w;
        __syncthreads();
    }
    int off = part[t] - sum;
    for (int f = lo; f < hi; ++f) { D.offset[f] = off; off += D.count[f]; }
    if (t == 1023) {
        D.offset[D.F] = part[t];
        D.host_word[0] = part[t];
        D.host_word[1] = running[t];
    }
}

template <int N>
__global__ __launch_bounds__(64) void k_fit_emit(FitDev D)
{
    extern __shared__ int32_t fit_lds[];            // [P] column -> free parameter (or -1), then N doubles (8-byte aligned)
    const int f = blockIdx.x, t = threadIdx.x;
    const int c = D.count[f];
    if (c == 0) return;
    const vmx_migrad::FitStateT<N>& s = ((const vmx_migrad::FitStateT<N>*)D.state)[f];
    const vmx_migrad::StageSpec& st = D.spec->stage[s.stage];
    const int n = st.n, P = D.P;
    int32_t* inv = fit_lds;
    double* ext = (double*)(fit_lds + ((P + 1) & ~1));
    for (int col = t; col < P; col += 64) inv[col] = -1;
    __syncthreads();
    if (t < n) inv[st.col[t]] = t;
    __syncthreads();
    const double* base = D.base + (size_t)f * P;
    const int64_t row0 = D.offset[f];
    const int32_t mock = D.mock_row ? D.mock_row[f] : -1;
    for (int q = 0; q < c; ++q) {
        if (t < n) ext[t] = vmx_migrad::int2ext(st, t, vmx_migrad::request_coord<N>(s, n, q, t));
        __syncthreads();
        double* row = D.theta + (row0 + q) * P;
        for (int col = t; col < P; col += 64) row[col] = inv[col] >= 0 ? ext[inv[col]] : base[col];
        if (t == 0) D.mock[row0 + q] = mock;
        __syncthreads();
    }
}
