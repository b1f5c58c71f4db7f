// MIGRAD as a resumable state machine: one fit = one FitState that is advanced from request to request.
//
// What the reference runs per fit (vega/minimizer.py:66-97: iminuit.Minuit(chi2, ...).migrad(ncall) - a bias-only pre-fit, then
// the full fit - and vega/analysis.py:224-308 for every Monte-Carlo mock) is restated in vega_amd/migrad.py (`_Fit`: a coroutine
// that yields the points of its next stage).  This header is the same algorithm, decision for decision, written so that it can
// run where the walkers live: `advance()` consumes the chi2 values of the fit's last request, runs Minuit's bookkeeping up to the
// next request (the 2 m points of a gradient cycle, one line-search trial, HESSE's steps, its off-diagonal points) and returns
// the number of points it asks for; `request_point()` expands the request.  No heap, no recursion, no HIP type: the library
// compiles it into a kernel (one thread per fit, vmx_fit.h) and tests/test_migrad_machine.py compiles it with g++ under
// AddressSanitizer / UBSan and holds it against `_Fit` point for point.
//
// Semantics kept from `_Fit` (vega_amd/migrad.py), i.e. Minuit2 strategy 1: internal coordinates with Minuit's guards at the
// limits; the step-derived seed refined by the two-point gradient (3 cycles, step tolerance 0.3, gradient tolerance 0.05);
// NegativeG2LineSearch; Newton step, MnLineSearch's parabolic search (12 evaluations at most), EDM with the old metric, Davidon's
// rank-two update (+ the rank-one term when delgam > gvg), dcovar; HESSE (MnHesse: diagonal steps with adaptation, the refined
// gradient, off-diagonal elements, MnPosDef) when dcovar > 0.05; iminuit's `iterate` re-runs; a chain of up to two Minuit
// objects per fit (the reference's bias pre-fit, then the full fit from its result).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define VMX_HD __host__ __device__
#else
#define VMX_HD
#endif

namespace vmx_migrad {

constexpr int MAXN = 32;                            // free parameters of one Minuit object
constexpr int MAX_STAGES = 2;
constexpr double EPS = 8.881784197001252e-16;       // MnMachinePrecision::Eps = 4 DBL_EPSILON = 2^-50
constexpr double EPS2 = 5.9604644775390625e-08;     // ... Eps2 = 2 sqrt(Eps) = 2^-24
constexpr double SQRT_EPS2 = 0.000244140625;        // 2^-12
constexpr double HALF_PI = 1.5707963267948966;
constexpr double SENTINEL = 1e99;                   // chi2 >= this: the model could not be evaluated there (the engine returns 1e100)

// strategy 1
constexpr int GRAD_NCYCLES = 3, HESS_NCYCLES = 5, HESS_GRAD_NCYCLES = 2;
constexpr double GRAD_STEP_TOL = 0.3, GRAD_TOL = 0.05, HESS_STEP_TOL = 0.3, HESS_G2_TOL = 0.05;

enum Flags : int32_t { F_VALID = 1, F_HESSE_FAILED = 2, F_ACCURATE = 4, F_AT_LIMIT = 8, F_MADE_POSDEF = 16 };
enum ReqKind : int32_t { REQ_NONE = 0, REQ_POINT = 1, REQ_PAIRS = 2, REQ_OFFDIAG = 3 };

struct StageSpec {
    int32_t n;                       // free parameters of this Minuit object
    int32_t col[MAXN];               // their columns of the parameter row
    int32_t has_lo[MAXN], has_hi[MAXN];
    double lo[MAXN], hi[MAXN];
    double err[MAXN];                // configured step sizes (external coordinates)
};

struct Spec {
    int32_t n_stages;                // 1, or 2: the first object's result is the second one's starting point
    int32_t n_params;                // columns of a parameter row
    int32_t iterate;                 // iminuit's migrad(iterate=5): runs per object while the minimum is neither valid nor at the call limit
    int32_t maxfcn;
    double up, tol;
    StageSpec stage[MAX_STAGES];
};

// results of stage s for fit f: x [f][n], ext [f][n], V [f][n][n] (n = the stage's), the scalars [f]
struct StageOut {
    double* x; double* ext; double* V; double* fval; double* edm;
    int32_t* flags; int64_t* nfcn; int32_t* n_iter;
};

enum Phase : int32_t {
    PH_INIT = 0, PH_SEED_F, PH_GRAD_BEGIN, PH_GRAD_VALS, PH_SEEDED, PH_NEGG2, PH_NEGG2_LS_DONE, PH_NEGG2_GRAD_DONE,
    PH_OUTER_BEGIN, PH_OUTER_LOOP, PH_IT_BEGIN, PH_IT_LS_DONE, PH_IT_GRAD_DONE, PH_IT_END,
    PH_LS_BEGIN, PH_LS_Y1, PH_LS_LOOP1, PH_LS_Y2, PH_LS_LOOP2, PH_LS_INNER, PH_LS_Y3,
    PH_HESSE_BEGIN, PH_H_AMIN, PH_H_DIAG_EMIT, PH_H_DIAG_VALS, PH_H_DIAG_DONE, PH_H_HG_EMIT, PH_H_HG_VALS, PH_H_OFF_EMIT,
    PH_H_OFF_VALS, PH_H_FINAL, PH_AFTER_HESSE, PH_OUTER_TAIL, PH_FINISH, PH_DONE
};

template <int N>
struct FitStateT {
    // control
    int32_t phase, stage, rerun, done;
    int32_t seeded, ipass, iterate_again, reached_limit;
    int32_t made_posdef, hesse_failed, accurate, run_failed;
    int32_t n_iter, pad0;
    int64_t nfcn, maxfcn_eff;
    // the request being served
    int32_t req_kind, req_m;
    int32_t req_idx[N];
    double req_step[N], req_x[N];
    // the minimisation state (MinimumState): internal point, value, gradient, metric
    double x[N], grd[N], g2[N], gstep[N], err[N];
    double fval, dcovar, edm;
    // two-point gradient (Numerical2PGradientCalculator)
    int32_t g_ret, g_cycle;
    uint32_t g_active, pad1;
    double g_fval, g_dfmin;
    double g_x[N], g_grd[N], g_g2[N], g_gstep[N], g_stepb4[N];
    // line search (MnLineSearch)
    int32_t ls_ret, ls_niter;
    double ls_x[N], ls_step[N];
    double ls_f0, ls_gdel, ls_overal, ls_undral, ls_toler8, ls_slamax, ls_slamin, ls_flast, ls_slam, ls_f2;
    double ls_p0x, ls_p0y, ls_p1x, ls_p1y, ls_p2x, ls_p2y, ls_fvmin, ls_xvmin;
    double ls_lam, ls_fmin;
    // NegativeG2LineSearch
    int32_t ng_iter, pad2;
    // HESSE (MnHesse)
    int32_t h_ret, h_j, h_failed;
    uint32_t h_open;
    double h_amin, h_aimsag, h_dfmin;
    double h_g2[N], h_gst[N], h_grd[N], h_dirin[N], h_yy[N], h_d[N], h_dmin[N], h_chgold[N];
    int32_t h_cyc[N], h_mult[N];
    // metric and two work matrices (row stride N)
    double V[N * N], M1[N * N], M2[N * N];
};
using FitState = FitStateT<MAXN>;


VMX_HD inline bool finite_(double v) { return __builtin_isfinite(v); }
VMX_HD inline double value_(const double* vals, int k)
{
    const double v = vals[k];
    return (finite_(v) && v < SENTINEL) ? v : __builtin_inf();
}
VMX_HD inline bool has_limits(const StageSpec& st, int i) { return st.has_lo[i] || st.has_hi[i]; }

// Minuit's parameter transformations (SinParameterTransformation, SqrtLow / SqrtUpParameterTransformation)
VMX_HD inline double ext2int(const StageSpec& st, int i, double value)
{
    const double distnn = 8. * SQRT_EPS2;
    if (st.has_lo[i] && st.has_hi[i]) {
        const double yy = 2. * (value - st.lo[i]) / (st.hi[i] - st.lo[i]) - 1.;
        if (yy * yy > 1. - EPS2) return yy < 0. ? -HALF_PI + distnn : HALF_PI - distnn;
        return asin(yy);
    }
    if (st.has_lo[i]) {
        const double yy = value - st.lo[i] + 1.;
        return yy * yy < 1. + EPS2 ? distnn : sqrt(yy * yy - 1.);
    }
    if (st.has_hi[i]) {
        const double yy = st.hi[i] - value + 1.;
        return yy * yy < 1. + EPS2 ? distnn : sqrt(yy * yy - 1.);
    }
    return value;
}
VMX_HD inline double int2ext(const StageSpec& st, int i, double value)
{
    if (st.has_lo[i] && st.has_hi[i]) return st.lo[i] + 0.5 * (st.hi[i] - st.lo[i]) * (sin(value) + 1.);
    if (st.has_lo[i]) return st.lo[i] - 1. + sqrt(value * value + 1.);
    if (st.has_hi[i]) return st.hi[i] + 1. - sqrt(value * value + 1.);
    return value;
}
VMX_HD inline double dint2ext(const StageSpec& st, int i, double value)
{
    if (st.has_lo[i] && st.has_hi[i]) return 0.5 * fabs((st.hi[i] - st.lo[i]) * cos(value));
    if (st.has_lo[i]) return value / sqrt(value * value + 1.);
    if (st.has_hi[i]) return -value / sqrt(value * value + 1.);
    return 1.;
}

template <int N>
VMX_HD inline int request_count(const FitStateT<N>& s, int n)
{
    switch (s.req_kind) {
    case REQ_POINT: return 1;
    case REQ_PAIRS: return 2 * s.req_m;
    case REQ_OFFDIAG: return n * (n - 1) / 2;
    default: return 0;
    }
}
// largest request a Minuit object over n parameters can make
VMX_HD inline int max_request(int n) { const int a = 2 * n, b = n * (n - 1) / 2; return a > b ? (a > 1 ? a : 1) : b; }

// internal coordinates of point q of the current request
template <int N>
VMX_HD inline void request_point(const FitStateT<N>& s, int n, int q, double* pt)
{
    for (int i = 0; i < n; ++i) pt[i] = s.req_x[i];
    if (s.req_kind == REQ_PAIRS) {
        const int i = s.req_idx[q >> 1];
        if (q & 1) pt[i] -= s.req_step[q >> 1]; else pt[i] += s.req_step[q >> 1];
    } else if (s.req_kind == REQ_OFFDIAG) {
        int i = 0, left = q;
        while (i < n - 2 && left >= n - 1 - i) { left -= n - 1 - i; ++i; }      // (bounded whatever q is)
        const int j = i + 1 + left;
        pt[i] += s.req_step[i];
        pt[j] += s.req_step[j];
    }
}

// coordinate i of point q of the current request (what request_point leaves in pt[i]): a thread per coordinate on the device
template <int N>
VMX_HD inline double request_coord(const FitStateT<N>& s, int n, int q, int i)
{
    double v = s.req_x[i];
    if (s.req_kind == REQ_PAIRS) {
        if (s.req_idx[q >> 1] == i) { if (q & 1) v -= s.req_step[q >> 1]; else v += s.req_step[q >> 1]; }
    } else if (s.req_kind == REQ_OFFDIAG) {
        int a = 0, left = q;
        while (a < n - 2 && left >= n - 1 - a) { left -= n - 1 - a; ++a; }
        if (i == a || i == a + 1 + left) v += s.req_step[i];
    }
    return v;
}

// ---- small dense linear algebra on row-stride-N matrices
template <int N>
VMX_HD inline double quad_form(const double* V, const double* g, int n)      // g^T V g
{
    double t = 0.;
    for (int i = 0; i < n; ++i) {
        double r = 0.;
        for (int j = 0; j < n; ++j) r += V[i * N + j] * g[j];
        t += g[i] * r;
    }
    return t;
}

// extreme eigenvalues of the symmetric matrix A (destroyed): cyclic Jacobi rotations
template <int N>
VMX_HD inline void eig_extremes(double* A, int n, double* emin, double* emax)
{
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0., diag = 0.;
        for (int i = 0; i < n; ++i) {
            diag += A[i * N + i] * A[i * N + i];
            for (int j = i + 1; j < n; ++j) off += A[i * N + j] * A[i * N + j];
        }
        if (!(off > 1e-32 * diag)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * N + q];
                if (apq == 0.) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                const double c = 1. / sqrt(t * t + 1.), sn = t * c;
                for (int k = 0; k < n; ++k) {       // columns p, q
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - sn * akq;
                    A[k * N + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {       // rows p, q
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - sn * aqk;
                    A[q * N + k] = sn * apk + c * aqk;
                }
            }
    }
    double lo = A[0], hi = A[0];
    for (int i = 1; i < n; ++i) { const double d = A[i * N + i]; if (d < lo) lo = d; if (d > hi) hi = d; }
    *emin = lo; *emax = hi;
}

// MnPosDef on M (in place; W: work matrix).  Returns whether the matrix had to be changed to become positive definite.
template <int N>
VMX_HD inline bool make_posdef(double* M, int n, double* W)
{
    if (n == 1) {
        if (M[0] < EPS) { M[0] = 1.0; return true; }
        return false;
    }
    const double epspdf = 1e-6;          // max(1e-6, Eps2)
    double dgmin = M[0];
    for (int i = 1; i < n; ++i) if (M[i * N + i] < dgmin) dgmin = M[i * N + i];
    const double dg = dgmin <= 0 ? 0.5 + epspdf - dgmin : 0.;
    for (int i = 0; i < n; ++i) {
        double d = M[i * N + i] + dg;
        if (d < 0.) d = 1.;
        M[i * N + i] = d;
    }
    for (int i = 0; i < n; ++i) {
        const double si = 1. / sqrt(M[i * N + i]);
        for (int j = 0; j < n; ++j) W[i * N + j] = M[i * N + j] * si * (1. / sqrt(M[j * N + j]));
    }
    // MnPosDef asks whether pmin > epspdf * max(|pmax|, 1) for the extreme eigenvalues of the scaled matrix W (unit diagonal).
    // Almost always it is, by orders of magnitude, and that can be SHOWN without eigenvalues: |pmax| <= R, the largest absolute row
    // sum (Gershgorin), and pmin > tau whenever W - tau I has a Cholesky factor.  With tau = 2 epspdf max(R, 1) + 1e-9 and every
    // pivot > 1e-9 the answer stands whatever the rounding (a Jacobi sweep costs a thread thousands of dependent instructions:
    // 80 us of the 290 us a round's bookkeeping took).  Anything closer goes through the eigenvalues.
    {
        double R = 0.;
        for (int i = 0; i < n; ++i) {
            double r = 0.;
            for (int j = 0; j < n; ++j) r += fabs(W[i * N + j]);
            if (r > R) R = r;
        }
        const double tau = 2. * epspdf * (R > 1. ? R : 1.) + 1e-9;
        bool clear = finite_(R);
        // in-place Cholesky of the lower triangle of a copy is not needed: the factor's entries go to the strict upper triangle of W
        // (W is symmetric and is destroyed by the eigenvalue path anyway, which reads it only if this test fails - so work on the
        // upper triangle and restore it from the lower one afterwards)
        for (int j = 0; j < n && clear; ++j) {
            double d = W[j * N + j] - tau;
            for (int k = 0; k < j; ++k) d -= W[k * N + j] * W[k * N + j];
            if (!(d > 1e-9)) { clear = false; break; }
            const double l = sqrt(d);
            for (int i = j + 1; i < n; ++i) {
                double v = W[i * N + j];                        // (lower triangle: untouched original)
                for (int k = 0; k < j; ++k) v -= W[k * N + i] * W[k * N + j];
                W[j * N + i] = v / l;                           // L[i][j] kept at the mirrored position
            }
        }
        if (clear) return false;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) W[i * N + j] = W[j * N + i];
    }
    double pmin, pmax;
    eig_extremes<N>(W, n, &pmin, &pmax);
    pmax = fabs(pmax) > 1. ? fabs(pmax) : 1.;
    if (pmin > epspdf * pmax) return false;
    const double padd = 0.001 * pmax - pmin;
    for (int i = 0; i < n; ++i) M[i * N + i] *= (1. + padd);
    return true;
}

// in-place inverse (Gauss-Jordan, partial pivoting); false for an exactly singular matrix
template <int N>
VMX_HD inline bool invert(double* A, int n)
{
    int piv[N];
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(A[c * N + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[r * N + c]) > best) { best = fabs(A[r * N + c]); p = r; }
        if (!(best > 0.)) return false;
        piv[c] = p;
        if (p != c) for (int k = 0; k < n; ++k) { const double t = A[c * N + k]; A[c * N + k] = A[p * N + k]; A[p * N + k] = t; }
        const double d = 1. / A[c * N + c];
        A[c * N + c] = 1.;
        for (int k = 0; k < n; ++k) A[c * N + k] *= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[r * N + c];
            if (f == 0.) continue;
            A[r * N + c] = 0.;
            for (int k = 0; k < n; ++k) A[r * N + k] -= f * A[c * N + k];
        }
    }
    for (int c = n - 1; c >= 0; --c)
        if (piv[c] != c) for (int r = 0; r < n; ++r) { const double t = A[r * N + c]; A[r * N + c] = A[r * N + piv[c]]; A[r * N + piv[c]] = t; }
    return true;
}

template <int N>
VMX_HD inline void set_diag_metric(FitStateT<N>& s, int n)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) s.V[i * N + j] = 0.;
    for (int i = 0; i < n; ++i) s.V[i * N + i] = fabs(s.g2[i]) > EPS2 ? 1. / s.g2[i] : 1.;
}

// sub-machine entries
template <int N>
VMX_HD inline void begin_gradient(FitStateT<N>& s, int n, const double* x, double fval, const double* grd, const double* g2,
                                  const double* gstep, double up, int32_t ret)
{
    for (int i = 0; i < n; ++i) { s.g_x[i] = x[i]; s.g_grd[i] = grd[i]; s.g_g2[i] = g2[i]; s.g_gstep[i] = gstep[i]; s.g_stepb4[i] = 0.; }
    s.g_fval = fval;
    s.g_dfmin = 8. * EPS2 * (fabs(fval) + up);
    s.g_active = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    s.g_cycle = 0;
    s.g_ret = ret;
    s.phase = PH_GRAD_BEGIN;
}
template <int N>
VMX_HD inline void begin_line_search(FitStateT<N>& s, int n, const double* x, double f0, double gdel, int32_t ret)
{
    // (s.ls_step holds the direction)
    for (int i = 0; i < n; ++i) s.ls_x[i] = x[i];
    s.ls_f0 = f0; s.ls_gdel = gdel; s.ls_ret = ret;
    s.phase = PH_LS_BEGIN;
}
template <int N>
VMX_HD inline void emit_line_point(FitStateT<N>& s, int n, double slam)
{
    for (int i = 0; i < n; ++i) s.req_x[i] = s.ls_x[i] + slam * s.ls_step[i];
    s.req_kind = REQ_POINT;
}

// Advance fit `fit` until its next request.  `vals`: the chi2 values of the points of its previous request (in request
// order); `base`: the fit's parameter row (read for start values, written with a stage's result before the next stage starts);
// `outs`: result arrays of the stages.  Returns the number of points requested; 0: the fit is done.
template <int N>
VMX_HD inline int advance(FitStateT<N>& s, const Spec& sp, const double* vals, double* base, const StageOut* outs, int64_t fit)
{
    const double up = sp.up;
    const double edmval = 0.002 * (sp.tol * up > EPS2 ? sp.tol * up : EPS2);
    for (;;) {
        const StageSpec& st = sp.stage[s.stage];
        const int n = st.n;
        switch (s.phase) {
        case PH_INIT: {
            if (s.rerun == 0) {
                for (int i = 0; i < n; ++i) { s.err[i] = st.err[i]; s.x[i] = ext2int(st, i, base[st.col[i]]); }
                s.seeded = 0;
            } else {
                // iminuit's `iterate`: from the previous run's values, with its parameter errors as steps and its error matrix as
                // the first metric (MnSeedGenerator on a state that carries a covariance: dcovar = 0)
                const StageOut& o = outs[s.stage];
                const double* xs = o.x + fit * n;
                const double* Vs = o.V + fit * n * n;
                for (int i = 0; i < n; ++i) {
                    const double ext = int2ext(st, i, xs[i]);
                    const double var = 2. * up * Vs[i * n + i];
                    const double dxs = sqrt(var > 0. ? var : 0.);
                    double e = has_limits(st, i) ? dxs * fabs(dint2ext(st, i, xs[i])) : dxs;
                    if (!(e > 0.)) e = st.err[i];
                    s.err[i] = e;
                    s.x[i] = ext2int(st, i, ext);
                    for (int j = 0; j < n; ++j) s.V[i * N + j] = Vs[i * n + j];
                }
                s.seeded = 1;
            }
            s.nfcn = 0; s.n_iter = 0; s.run_failed = 0; s.reached_limit = 0;
            s.made_posdef = 0; s.hesse_failed = 0; s.accurate = 0;
            for (int i = 0; i < n; ++i) s.req_x[i] = s.x[i];
            s.req_kind = REQ_POINT;
            s.phase = PH_SEED_F;
            return 1;
        }
        case PH_SEED_F: {
            s.fval = value_(vals, 0);
            s.nfcn += 1;
            if (!finite_(s.fval)) { s.run_failed = 1; s.phase = PH_FINISH; break; }
            // InitialGradientCalculator: first / second derivatives from the configured step sizes
            for (int i = 0; i < n; ++i) {
                const double var = s.x[i], werr = s.err[i];
                const double sav = int2ext(st, i, var);
                double sav2 = sav + werr;
                if (st.has_hi[i] && sav2 > st.hi[i]) sav2 = st.hi[i];
                const double vplu = ext2int(st, i, sav2) - var;
                sav2 = sav - werr;
                if (st.has_lo[i] && sav2 < st.lo[i]) sav2 = st.lo[i];
                const double vmin = ext2int(st, i, sav2) - var;
                const double gsmin = 8. * EPS2 * (fabs(var) + EPS2);
                double dirin = 0.5 * (fabs(vplu) + fabs(vmin));
                if (!(dirin > gsmin)) dirin = gsmin;
                s.g2[i] = 2.0 * up / (dirin * dirin);
                double gs = 0.1 * dirin;
                if (!(gs > gsmin)) gs = gsmin;
                s.grd[i] = s.g2[i] * dirin;
                if (has_limits(st, i) && gs > 0.5) gs = 0.5;
                s.gstep[i] = gs;
            }
            begin_gradient(s, n, s.x, s.fval, s.grd, s.g2, s.gstep, up, PH_SEEDED);
            break;
        }
        case PH_GRAD_BEGIN: {
            if (s.g_cycle >= GRAD_NCYCLES) { s.phase = s.g_ret; break; }
            const double vrysml = 8. * EPS * EPS;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                if (!((s.g_active >> i) & 1u)) continue;
                const double epspri = EPS2 + fabs(s.g_grd[i] * EPS2);
                const double optstp = sqrt(s.g_dfmin / (fabs(s.g_g2[i]) + epspri));
                double step = fabs(0.1 * s.g_gstep[i]);
                if (optstp > step) step = optstp;                           // max(optstp, |0.1 gstep|)
                if (has_limits(st, i) && step > 0.5) step = 0.5;
                if (10. * fabs(s.g_gstep[i]) < step) step = 10. * fabs(s.g_gstep[i]);     // min(step, 10 |gstep|)
                double floor_ = 8. * fabs(EPS2 * s.g_x[i]);
                if (vrysml > floor_) floor_ = vrysml;
                if (floor_ > step) step = floor_;                           // max(step, max(vrysml, 8 |eps2 x|))
                if (fabs((step - s.g_stepb4[i]) / step) < GRAD_STEP_TOL) { s.g_active &= ~(1u << i); continue; }
                s.g_gstep[i] = step;
                s.g_stepb4[i] = step;
                s.req_idx[m] = i;
                s.req_step[m] = step;
                ++m;
            }
            if (m == 0) { s.phase = s.g_ret; break; }
            for (int i = 0; i < n; ++i) s.req_x[i] = s.g_x[i];
            s.req_kind = REQ_PAIRS; s.req_m = m;
            s.phase = PH_GRAD_VALS;
            return 2 * m;
        }
        case PH_GRAD_VALS: {
            const int m = s.req_m;
            s.nfcn += 2 * m;
            for (int q = 0; q < m; ++q) {
                const int i = s.req_idx[q];
                const double step = s.req_step[q];
                const double fs1 = value_(vals, 2 * q), fs2 = value_(vals, 2 * q + 1);
                const double grdb4 = s.g_grd[i];
                s.g_grd[i] = 0.5 * (fs1 - fs2) / step;
                s.g_g2[i] = (fs1 + fs2 - 2. * s.g_fval) / step / step;
                if (fabs(grdb4 - s.g_grd[i]) / (fabs(s.g_grd[i]) + s.g_dfmin / step) < GRAD_TOL) s.g_active &= ~(1u << i);
            }
            s.g_cycle += 1;
            s.phase = PH_GRAD_BEGIN;
            break;
        }
        case PH_SEEDED: {
            for (int i = 0; i < n; ++i) { s.grd[i] = s.g_grd[i]; s.g2[i] = s.g_g2[i]; s.gstep[i] = s.g_gstep[i]; }
            if (s.seeded) s.dcovar = 0.;
            else { set_diag_metric(s, n); s.dcovar = 1.; }
            s.edm = 0.5 * quad_form<N>(s.V, s.grd, n);
            bool negative = false;
            for (int i = 0; i < n; ++i) if (s.g2[i] <= 0.) negative = true;
            s.ng_iter = 0;
            s.phase = negative ? PH_NEGG2 : PH_OUTER_BEGIN;
            break;
        }
        case PH_NEGG2: {
            // NegativeG2LineSearch: walk along a direction with a negative second derivative until it turns positive
            int pick = -1;
            bool any = false;
            if (s.ng_iter < 2 * n)
                for (int i = 0; i < n && pick < 0; ++i) {
                    if (!(s.g2[i] <= 0.)) continue;
                    any = true;
                    if (fabs(s.gstep[i]) < EPS2) continue;
                    pick = i;
                }
            (void)any;
            if (pick < 0) {
                set_diag_metric(s, n);
                s.dcovar = 1.;
                s.edm = 0.5 * quad_form<N>(s.V, s.grd, n);
                s.phase = PH_OUTER_BEGIN;
                break;
            }
            for (int i = 0; i < n; ++i) s.ls_step[i] = 0.;
            s.ls_step[pick] = s.gstep[pick] * (s.grd[pick] > 0. ? -1. : 1.);
            begin_line_search(s, n, s.x, s.fval, s.ls_step[pick] * s.grd[pick], PH_NEGG2_LS_DONE);
            break;
        }
        case PH_NEGG2_LS_DONE: {
            for (int i = 0; i < n; ++i) s.x[i] = s.ls_x[i] + s.ls_lam * s.ls_step[i];
            s.fval = s.ls_fmin;
            begin_gradient(s, n, s.x, s.fval, s.grd, s.g2, s.gstep, up, PH_NEGG2_GRAD_DONE);
            break;
        }
        case PH_NEGG2_GRAD_DONE: {
            for (int i = 0; i < n; ++i) { s.grd[i] = s.g_grd[i]; s.g2[i] = s.g_g2[i]; s.gstep[i] = s.g_gstep[i]; }
            s.ng_iter += 1;
            s.phase = PH_NEGG2;
            break;
        }
        case PH_OUTER_BEGIN: {
            s.maxfcn_eff = sp.maxfcn;
            s.ipass = 0;
            s.phase = PH_OUTER_LOOP;
            break;
        }
        case PH_OUTER_LOOP: {
            s.iterate_again = 0;
            s.phase = PH_IT_BEGIN;
            break;
        }
        case PH_IT_BEGIN: {
            // VariableMetricBuilder: Newton step with the current metric
            double gdel = 0.;
            for (int i = 0; i < n; ++i) {
                double r = 0.;
                for (int j = 0; j < n; ++j) r += s.V[i * N + j] * s.grd[j];
                s.ls_step[i] = -r;
                gdel += s.ls_step[i] * s.grd[i];
            }
            if (gdel > 0.) {
                for (int i = 0; i < n * N; ++i) s.M1[i] = s.V[i];
                make_posdef<N>(s.M1, n, s.M2);
                for (int i = 0; i < n * N; ++i) s.V[i] = s.M1[i];
                gdel = 0.;
                for (int i = 0; i < n; ++i) {
                    double r = 0.;
                    for (int j = 0; j < n; ++j) r += s.V[i * N + j] * s.grd[j];
                    s.ls_step[i] = -r;
                    gdel += s.ls_step[i] * s.grd[i];
                }
                if (gdel > 0.) { s.reached_limit = 0; s.phase = PH_IT_END; break; }
            }
            begin_line_search(s, n, s.x, s.fval, gdel, PH_IT_LS_DONE);
            break;
        }
        case PH_IT_LS_DONE: {
            const double fnew = s.ls_fmin;
            if (fabs(fnew - s.fval) <= fabs(s.fval) * EPS) {        // no improvement
                s.reached_limit = s.nfcn >= s.maxfcn_eff;
                s.phase = PH_IT_END;
                break;
            }
            for (int i = 0; i < n; ++i) s.req_x[i] = s.ls_x[i] + s.ls_lam * s.ls_step[i];     // (scratch: the new point)
            begin_gradient(s, n, s.req_x, fnew, s.grd, s.g2, s.gstep, up, PH_IT_GRAD_DONE);
            break;
        }
        case PH_IT_GRAD_DONE: {
            double edm = 0.5 * quad_form<N>(s.V, s.g_grd, n);
            if (edm != edm) { s.reached_limit = 0; s.phase = PH_IT_END; break; }
            if (edm < 0.) {
                for (int i = 0; i < n * N; ++i) s.M1[i] = s.V[i];
                make_posdef<N>(s.M1, n, s.M2);
                for (int i = 0; i < n * N; ++i) s.V[i] = s.M1[i];
                edm = 0.5 * quad_form<N>(s.V, s.g_grd, n);
                if (edm < 0.) { s.reached_limit = 0; s.phase = PH_IT_END; break; }
            }
            // Davidon's update (DavidonErrorUpdator): dx -> M1 row 0, dg -> row 1, vg -> row 2 (scratch)
            double* dx = s.M1; double* dg = s.M1 + N; double* vg = s.M1 + 2 * N;
            double delgam = 0., gvg = 0.;
            for (int i = 0; i < n; ++i) { dx[i] = s.g_x[i] - s.x[i]; dg[i] = s.g_grd[i] - s.grd[i]; delgam += dx[i] * dg[i]; }
            for (int i = 0; i < n; ++i) {
                double r = 0.;
                for (int j = 0; j < n; ++j) r += s.V[i * N + j] * dg[j];
                vg[i] = r;
            }
            for (int i = 0; i < n; ++i) gvg += dg[i] * vg[i];
            double dcov = s.dcovar;
            if (!(delgam == 0. || gvg <= 0.)) {
                const bool rank1 = delgam > gvg;
                double sum_upd = 0., sum_v = 0.;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        double u = dx[i] * dx[j] / delgam - vg[i] * vg[j] / gvg;
                        if (rank1) u += gvg * ((dx[i] / delgam - vg[i] / gvg) * (dx[j] / delgam - vg[j] / gvg));
                        const double v = s.V[i * N + j] + u;
                        s.M2[i * N + j] = v;
                        if (j >= i) { sum_upd += fabs(u); sum_v += fabs(v); }
                    }
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) s.V[i * N + j] = s.M2[i * N + j];
                dcov = 0.5 * (s.dcovar + sum_upd / sum_v);
            }
            for (int i = 0; i < n; ++i) { s.x[i] = s.g_x[i]; s.grd[i] = s.g_grd[i]; s.g2[i] = s.g_g2[i]; s.gstep[i] = s.g_gstep[i]; }
            s.fval = s.g_fval; s.dcovar = dcov; s.edm = edm;
            s.made_posdef = 0; s.hesse_failed = 0; s.accurate = 0;
            s.n_iter += 1;
            if (edm * (1. + 3. * dcov) > edmval && s.nfcn < s.maxfcn_eff) { s.phase = PH_IT_BEGIN; break; }
            s.reached_limit = s.nfcn >= s.maxfcn_eff;
            s.phase = PH_IT_END;
            break;
        }
        case PH_IT_END: {
            if (s.reached_limit) { s.phase = PH_FINISH; break; }
            if (s.dcovar > 0.05) { s.h_ret = PH_AFTER_HESSE; s.phase = PH_HESSE_BEGIN; break; }
            s.phase = PH_OUTER_TAIL;
            break;
        }
        // ---- MnLineSearch along s.ls_step from s.ls_x (value ls_f0, slope ls_gdel): (ls_lam, ls_fmin)
        case PH_LS_BEGIN: {
            s.ls_overal = 1000.; s.ls_undral = -100.;
            s.ls_niter = 1;
            double slamin = 0.;
            for (int i = 0; i < n; ++i) {
                if (s.ls_step[i] == 0.) continue;
                const double ratio = fabs(s.ls_x[i] / s.ls_step[i]);
                if (slamin == 0. || ratio < slamin) slamin = ratio;
            }
            if (fabs(slamin) < EPS) slamin = EPS;
            s.ls_slamin = slamin * EPS2;
            emit_line_point(s, n, 1.);
            s.phase = PH_LS_Y1;
            return 1;
        }
        case PH_LS_Y1: {
            const double f1 = value_(vals, 0);
            s.nfcn += 1;
            s.ls_niter += 1;
            s.ls_fvmin = s.ls_f0; s.ls_xvmin = 0.;
            if (f1 < s.ls_f0) { s.ls_fvmin = f1; s.ls_xvmin = 1.; }
            s.ls_toler8 = 0.05; s.ls_slamax = 5.; s.ls_flast = f1; s.ls_slam = 1.;
            s.ls_p0x = 0.; s.ls_p0y = s.ls_f0; s.ls_p1x = 1.; s.ls_p1y = f1;
            s.ls_f2 = 0.;
            s.phase = PH_LS_LOOP1;
            break;
        }
        case PH_LS_LOOP1: {
            const double denom = 2. * (s.ls_flast - s.ls_f0 - s.ls_gdel * s.ls_slam) / (s.ls_slam * s.ls_slam);
            double slam = denom != 0. ? -s.ls_gdel / denom : 1.;
            if (slam < 0.) slam = s.ls_slamax;
            if (slam > s.ls_slamax) slam = s.ls_slamax;
            if (slam < s.ls_toler8) slam = s.ls_toler8;
            if (slam < s.ls_slamin || (fabs(slam - 1.) < s.ls_toler8 && s.ls_p1y < s.ls_p0y)) {
                s.ls_lam = s.ls_xvmin; s.ls_fmin = s.ls_fvmin; s.phase = s.ls_ret; break;
            }
            if (fabs(slam - 1.) < s.ls_toler8) slam = 1. + s.ls_toler8;
            s.ls_slam = slam;
            emit_line_point(s, n, slam);
            s.phase = PH_LS_Y2;
            return 1;
        }
        case PH_LS_Y2: {
            const double f2 = value_(vals, 0);
            s.nfcn += 1;
            s.ls_f2 = f2;
            s.ls_niter += 1;
            bool iterate = false;
            if (f2 < s.ls_fvmin) { s.ls_fvmin = f2; s.ls_xvmin = s.ls_slam; }
            if (fabs(s.ls_p0y - s.ls_fvmin) < fabs(s.ls_fvmin) * EPS) {
                iterate = true;
                s.ls_flast = f2;
                s.ls_toler8 = 0.05 * s.ls_slam;
                s.ls_overal = s.ls_slam - s.ls_toler8;
                s.ls_slamax = s.ls_overal;
                s.ls_p1x = s.ls_slam; s.ls_p1y = s.ls_flast;
            }
            if (iterate && s.ls_niter < 12) { s.phase = PH_LS_LOOP1; break; }
            if (s.ls_niter >= 12) { s.ls_lam = s.ls_xvmin; s.ls_fmin = s.ls_fvmin; s.phase = s.ls_ret; break; }
            s.ls_p2x = s.ls_slam; s.ls_p2y = f2;
            s.phase = PH_LS_LOOP2;
            break;
        }
        case PH_LS_LOOP2: {
            const double alpha = 2.;
            if (alpha * fabs(s.ls_xvmin) > s.ls_slamax) s.ls_slamax = alpha * fabs(s.ls_xvmin);
            // parabola through p0, p1, p2: y = a x^2 + b x + c (MnParabolaFactory)
            const double x1 = s.ls_p0x, y1 = s.ls_p0y, x2 = s.ls_p1x, y2 = s.ls_p1y, x3 = s.ls_p2x, y3 = s.ls_p2y;
            const double xm = (x1 + x2 + x3) / 3.;
            const double dx1 = x1 - xm, dx2 = x2 - xm, dx3 = x3 - xm;
            const double dx12 = dx1 - dx2, dx13 = dx1 - dx3, dx23 = dx2 - dx3;
            const double a = y1 / (dx12 * dx13) - y2 / (dx12 * dx23) + y3 / (dx13 * dx23);
            double b = -y1 * (dx2 + dx3) / (dx12 * dx13) + y2 * (dx1 + dx3) / (dx12 * dx23) - y3 * (dx1 + dx2) / (dx13 * dx23);
            b -= 2. * xm * a;
            double slam;
            if (a < EPS2) {
                const double slopem = 2. * a * s.ls_xvmin + b;
                slam = slopem < 0. ? s.ls_xvmin + s.ls_slamax : s.ls_xvmin - s.ls_slamax;
            } else {
                slam = -b / (2. * a);
                if (slam > s.ls_xvmin + s.ls_slamax) slam = s.ls_xvmin + s.ls_slamax;
                if (slam < s.ls_xvmin - s.ls_slamax) slam = s.ls_xvmin - s.ls_slamax;
            }
            if (slam > 0.) { if (slam > s.ls_overal) slam = s.ls_overal; }
            else if (slam < s.ls_undral) slam = s.ls_undral;
            s.ls_slam = slam;
            s.phase = PH_LS_INNER;
            break;
        }
        case PH_LS_INNER: {
            const double slam = s.ls_slam;
            double toler9 = fabs(s.ls_toler8 * slam);
            if (s.ls_toler8 > toler9) toler9 = s.ls_toler8;
            if (fabs(s.ls_p0x - slam) < toler9 || fabs(s.ls_p1x - slam) < toler9 || fabs(s.ls_p2x - slam) < toler9) {
                s.ls_lam = s.ls_xvmin; s.ls_fmin = s.ls_fvmin; s.phase = s.ls_ret; break;
            }
            emit_line_point(s, n, slam);
            s.phase = PH_LS_Y3;
            return 1;
        }
        case PH_LS_Y3: {
            const double f3 = value_(vals, 0);
            s.nfcn += 1;
            double slam = s.ls_slam;
            bool iterate = false;
            if (f3 > s.ls_p0y && f3 > s.ls_p1y && f3 > s.ls_p2y) {
                if (slam > s.ls_xvmin && slam - s.ls_toler8 < s.ls_overal) s.ls_overal = slam - s.ls_toler8;
                if (slam < s.ls_xvmin && slam + s.ls_toler8 > s.ls_undral) s.ls_undral = slam + s.ls_toler8;
                slam = 0.5 * (slam + s.ls_xvmin);
                s.ls_slam = slam;
                iterate = true;
                s.ls_niter += 1;
            }
            if (iterate && s.ls_niter < 12) { s.phase = PH_LS_INNER; break; }
            if (s.ls_niter >= 12) { s.ls_lam = s.ls_xvmin; s.ls_fmin = s.ls_fvmin; s.phase = s.ls_ret; break; }
            // the new point replaces the worst of the three
            if (s.ls_p0y > s.ls_p1y && s.ls_p0y > s.ls_p2y) { s.ls_p0x = slam; s.ls_p0y = f3; }
            else if (s.ls_p1y > s.ls_p0y && s.ls_p1y > s.ls_p2y) { s.ls_p1x = slam; s.ls_p1y = f3; }
            else { s.ls_p2x = slam; s.ls_p2y = f3; }
            if (f3 < s.ls_fvmin) { s.ls_fvmin = f3; s.ls_xvmin = slam; }
            else {
                if (slam > s.ls_xvmin && slam - s.ls_toler8 < s.ls_overal) s.ls_overal = slam - s.ls_toler8;
                if (slam < s.ls_xvmin && slam + s.ls_toler8 > s.ls_undral) s.ls_undral = slam + s.ls_toler8;
            }
            s.ls_niter += 1;
            if (s.ls_niter >= 12) { s.ls_lam = s.ls_xvmin; s.ls_fmin = s.ls_fvmin; s.phase = s.ls_ret; break; }
            s.phase = PH_LS_LOOP2;
            break;
        }
        // ---- MnHesse (strategy 1) at s.x
        case PH_HESSE_BEGIN: {
            for (int i = 0; i < n; ++i) s.req_x[i] = s.x[i];
            s.req_kind = REQ_POINT;
            s.phase = PH_H_AMIN;
            return 1;
        }
        case PH_H_AMIN: {
            s.h_amin = value_(vals, 0);
            s.nfcn += 1;
            s.h_aimsag = SQRT_EPS2 * (fabs(s.h_amin) + up);
            for (int i = 0; i < n; ++i) {
                s.h_g2[i] = s.g2[i]; s.h_gst[i] = s.gstep[i]; s.h_grd[i] = s.grd[i];
                s.h_dirin[i] = s.gstep[i]; s.h_yy[i] = 0.;
                s.h_dmin[i] = 8. * EPS2 * (fabs(s.x[i]) + EPS2);
                s.h_d[i] = fabs(s.gstep[i]) > s.h_dmin[i] ? fabs(s.gstep[i]) : s.h_dmin[i];
                s.h_cyc[i] = 0; s.h_mult[i] = 0;
                for (int j = 0; j < n; ++j) s.M1[i * N + j] = 0.;
            }
            s.h_open = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
            s.h_failed = 0;
            s.phase = PH_H_DIAG_EMIT;
            break;
        }
        case PH_H_DIAG_EMIT: {
            if (s.h_failed || s.h_open == 0u) { s.phase = PH_H_DIAG_DONE; break; }
            int m = 0;
            for (int i = 0; i < n; ++i)
                if ((s.h_open >> i) & 1u) { s.req_idx[m] = i; s.req_step[m] = s.h_d[i]; ++m; }
            for (int i = 0; i < n; ++i) s.req_x[i] = s.x[i];
            s.req_kind = REQ_PAIRS; s.req_m = m;
            s.phase = PH_H_DIAG_VALS;
            return 2 * m;
        }
        case PH_H_DIAG_VALS: {
            const int m = s.req_m;
            s.nfcn += 2 * m;
            for (int q = 0; q < m && !s.h_failed; ++q) {
                const int i = s.req_idx[q];
                const double d = s.h_d[i];
                const double fs1 = value_(vals, 2 * q), fs2 = value_(vals, 2 * q + 1);
                const double sag = 0.5 * (fs1 + fs2 - 2. * s.h_amin);
                if (!(sag > EPS2)) {
                    // flat or negative curvature at this step: widen it (at most five times per cycle)
                    s.h_mult[i] += 1;
                    if (has_limits(st, i)) {
                        if (d > 0.5 || s.h_mult[i] >= 5) { s.h_failed = 1; break; }
                        s.h_d[i] = d * 10.;
                        if (s.h_d[i] > 0.5) s.h_d[i] = 0.51;
                    } else {
                        if (s.h_mult[i] >= 5) { s.h_failed = 1; break; }
                        s.h_d[i] = d * 10.;
                    }
                    continue;
                }
                s.h_mult[i] = 0;
                const double g2bfor = s.h_g2[i];
                s.h_g2[i] = 2. * sag / (d * d);
                s.h_grd[i] = (fs1 - fs2) / (2. * d);
                s.h_gst[i] = d;
                s.h_dirin[i] = d;
                s.h_yy[i] = fs1;
                double dn = sqrt(2. * s.h_aimsag / fabs(s.h_g2[i]));
                if (has_limits(st, i) && dn > 0.5) dn = 0.5;
                if (dn < s.h_dmin[i]) dn = s.h_dmin[i];
                s.h_cyc[i] += 1;
                if (fabs((dn - d) / dn) < HESS_STEP_TOL || fabs((s.h_g2[i] - g2bfor) / s.h_g2[i]) < HESS_G2_TOL || s.h_cyc[i] >= HESS_NCYCLES) {
                    s.M1[i * N + i] = s.h_g2[i];
                    s.h_open &= ~(1u << i);
                    continue;
                }
                if (dn > 10. * d) dn = 10. * d;
                if (dn < 0.1 * d) dn = 0.1 * d;
                s.h_d[i] = dn;
            }
            s.phase = PH_H_DIAG_EMIT;
            break;
        }
        case PH_H_DIAG_DONE: {
            if (s.h_failed) { s.hesse_failed = 1; s.fval = s.h_amin; s.phase = s.h_ret; break; }
            // refined first derivatives (HessianGradientCalculator): h_d = the steps, h_dmin = their lower bounds
            s.h_dfmin = 4. * EPS2 * (fabs(s.h_amin) + up);
            for (int i = 0; i < n; ++i) {
                const double dmin = 4. * EPS2 * (s.x[i] + EPS2);
                const double epspri = EPS2 + fabs(s.h_grd[i] * EPS2);
                const double optstp = sqrt(s.h_dfmin / (fabs(s.h_g2[i]) + epspri));
                double d = 0.2 * fabs(s.h_gst[i]);
                if (d > optstp) d = optstp;
                if (d < dmin) d = dmin;
                s.h_d[i] = d; s.h_dmin[i] = dmin; s.h_chgold[i] = 10000.;
            }
            s.h_open = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
            s.h_j = 0;
            s.phase = PH_H_HG_EMIT;
            break;
        }
        case PH_H_HG_EMIT: {
            if (s.h_j >= HESS_GRAD_NCYCLES || s.h_open == 0u) { s.phase = PH_H_OFF_EMIT; break; }
            int m = 0;
            for (int i = 0; i < n; ++i)
                if ((s.h_open >> i) & 1u) { s.req_idx[m] = i; s.req_step[m] = s.h_d[i]; ++m; }
            for (int i = 0; i < n; ++i) s.req_x[i] = s.x[i];
            s.req_kind = REQ_PAIRS; s.req_m = m;
            s.phase = PH_H_HG_VALS;
            return 2 * m;
        }
        case PH_H_HG_VALS: {
            const int m = s.req_m;
            s.nfcn += 2 * m;
            for (int q = 0; q < m; ++q) {
                const int i = s.req_idx[q];
                const double d = s.h_d[i];
                const double fs1 = value_(vals, 2 * q), fs2 = value_(vals, 2 * q + 1);
                const double grdold = s.h_grd[i];
                const double grdnew = (fs1 - fs2) / (2. * d);
                const double dgmin = EPS * (fabs(fs1) + fabs(fs2)) / d;
                if (fabs(grdnew) < EPS) { s.h_open &= ~(1u << i); continue; }
                const double change = fabs((grdold - grdnew) / grdnew);
                if (change > s.h_chgold[i] && s.h_j > 1) { s.h_open &= ~(1u << i); continue; }
                s.h_chgold[i] = change;
                s.h_grd[i] = grdnew;
                s.h_gst[i] = d;
                if (change < 0.05 || fabs(grdold - grdnew) < dgmin || d < s.h_dmin[i]) { s.h_open &= ~(1u << i); continue; }
                s.h_d[i] = d * 0.2;
            }
            s.h_j += 1;
            s.phase = PH_H_HG_EMIT;
            break;
        }
        case PH_H_OFF_EMIT: {
            if (n < 2) { s.phase = PH_H_FINAL; break; }
            for (int i = 0; i < n; ++i) { s.req_x[i] = s.x[i]; s.req_step[i] = s.h_dirin[i]; }
            s.req_kind = REQ_OFFDIAG;
            s.phase = PH_H_OFF_VALS;
            return n * (n - 1) / 2;
        }
        case PH_H_OFF_VALS: {
            s.nfcn += n * (n - 1) / 2;
            int q = 0;
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j, ++q) {
                    const double el = (value_(vals, q) + s.h_amin - s.h_yy[i] - s.h_yy[j]) / (s.h_dirin[i] * s.h_dirin[j]);
                    s.M1[i * N + j] = el;
                    s.M1[j * N + i] = el;
                }
            s.phase = PH_H_FINAL;
            break;
        }
        case PH_H_FINAL: {
            const bool made = make_posdef<N>(s.M1, n, s.M2);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) s.M2[i * N + j] = s.M1[i * N + j];
            if (!invert<N>(s.M2, n)) { s.hesse_failed = 1; s.fval = s.h_amin; s.phase = s.h_ret; break; }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) s.V[i * N + j] = s.M2[i * N + j];
            for (int i = 0; i < n; ++i) { s.grd[i] = s.h_grd[i]; s.g2[i] = s.h_g2[i]; s.gstep[i] = s.h_gst[i]; }
            s.fval = s.h_amin;
            s.dcovar = 0.;
            s.edm = 0.5 * quad_form<N>(s.V, s.grd, n);
            s.made_posdef = made ? 1 : 0; s.hesse_failed = 0; s.accurate = made ? 0 : 1;
            s.phase = s.h_ret;
            break;
        }
        case PH_AFTER_HESSE: {
            if (s.hesse_failed) { s.phase = PH_FINISH; break; }
            if (s.edm > edmval && s.edm >= fabs(EPS2 * s.fval)) s.iterate_again = 1;
            s.phase = PH_OUTER_TAIL;
            break;
        }
        case PH_OUTER_TAIL: {
            if (s.ipass == 0) s.maxfcn_eff = (int64_t)(sp.maxfcn * 1.3);
            s.ipass += 1;
            s.phase = s.iterate_again ? PH_OUTER_LOOP : PH_FINISH;
            break;
        }
        case PH_FINISH: {
            const StageOut& o = outs[s.stage];
            const bool bad = s.run_failed || !finite_(s.fval);
            int32_t flags = 0;
            if (!bad) {
                if (!s.reached_limit && s.edm <= 10. * edmval && !s.hesse_failed) flags |= F_VALID;
                if (s.hesse_failed) flags |= F_HESSE_FAILED;
                if (s.accurate) flags |= F_ACCURATE;
                if (s.reached_limit) flags |= F_AT_LIMIT;
                if (s.made_posdef) flags |= F_MADE_POSDEF;
            }
            if (s.rerun == 0) {
                o.nfcn[fit] = s.nfcn;
                o.n_iter[fit] = s.n_iter;
            } else {
                o.nfcn[fit] += s.nfcn;
                o.n_iter[fit] += s.n_iter;
            }
            if (s.rerun == 0 || !bad) {
                for (int i = 0; i < n; ++i) {
                    o.x[fit * n + i] = s.x[i];
                    for (int j = 0; j < n; ++j) o.V[(fit * n + i) * n + j] = bad ? 0. : s.V[i * N + j];
                }
                o.fval[fit] = bad ? __builtin_inf() : s.fval;
                o.edm[fit] = bad ? __builtin_inf() : s.edm;
                o.flags[fit] = bad ? F_HESSE_FAILED : flags;
            } else
                o.flags[fit] |= F_AT_LIMIT;      // (nothing to restart from: the previous state stays, no further run)
            const int32_t fl = o.flags[fit];
            if (!(fl & F_VALID) && !(fl & F_AT_LIMIT) && finite_(o.fval[fit]) && s.rerun + 1 < sp.iterate) {
                s.rerun += 1;
                s.phase = PH_INIT;
                break;
            }
            // this Minuit object is done: its external values; the next object starts from them
            for (int i = 0; i < n; ++i) o.ext[fit * n + i] = int2ext(st, i, o.x[fit * n + i]);
            if (s.stage + 1 < sp.n_stages) {
                if (finite_(o.fval[fit])) for (int i = 0; i < n; ++i) base[st.col[i]] = o.ext[fit * n + i];
                s.stage += 1;
                s.rerun = 0;
                s.phase = PH_INIT;
                break;
            }
            s.done = 1;
            s.req_kind = REQ_NONE;
            s.phase = PH_DONE;
            return 0;
        }
        default:
            s.done = 1;
            s.req_kind = REQ_NONE;
            return 0;
        }
    }
}

template <int N>
VMX_HD inline void reset(FitStateT<N>& s)
{
    s.phase = PH_INIT; s.stage = 0; s.rerun = 0; s.done = 0; s.req_kind = REQ_NONE; s.req_m = 0;
    s.nfcn = 0; s.n_iter = 0;
}

}  // namespace vmx_migrad
