// Device-side data layout and kernels of the vegamx engine (gfx950 / CDNA4 only).
//
// Stage map (reference file:line each kernel replaces):
//   k_theta_affine    parameter-level blinding of the walkers               vega_interface.py:389-421, utils.py:375-393
//   k_prologue        parameters -> per-(walker, pipeline) scalars        utils.py:45-108, scale_parameters.py:38-230
//   k_gk_table        G(k,mu) binning table (static)                       power_spectrum.py:481-502
//   k_pk_multipoles   P(k,mu) and its Legendre projection, fused           power_spectrum.py:87-196 + pktoxi.py:138
//   k_xtab            per-batch tables: D_NL(k,mu)^p G(k,mu) of a batch that shares its non-linear parameters (level 1),
//                     times the Gaussian smoothing / broadening factors, + a table for the peak component (level 2)
//                                                                          power_spectrum.py:435-502, :382-417, :526-556
//   k_pk_tab2         P(k,mu) and its projection for the core groups against level-2 tables   power_spectrum.py:163-380
//   k_pk_poly         pipelines whose mu dependence is the Kaiser polynomial x static G: closed form
//   k_pk_w            shared-W groups (pipelines that differ in their Kaiser polynomials only) of large batches
//   k_gemm_nt44 / k_gemm_nt / k_gemv / k_gemv1   D = A . X for a static matrix and a batch of walker vectors
//                     (4x4x4 four-block MFMA / 16x16x4 MFMA / streaming for <= 8 walkers / one walker):
//                     FFTLog+spline operator (pktoxi.py:141-144), metal matrices (metals.py:338-367),
//                     distortion matrix (model.py:143-144), inverse covariance (vega_interface.py:316)
//   k_metal_kron      metal matrix in Kronecker form A (x) B (new_metals)  metals.py:338-367, :501-655
//   k_xi_bins (+ _static; k_poly_bins at set-up)   spline evaluation on rescaled bins, Legendre sum, bias evolution, growth,
//                     QSO radiation                                         pktoxi.py:144-162, correlation_func.py:117-236,276-349,446-489
//   k_assemble        peak/smooth/metals combination + pre-distortion broadband   model.py:119-140,186, metals.py:331-334
//   k_post            post-distortion broadband, model output, masked residual   model.py:147-149, vega_interface.py:310-315
//   k_chi2            diff^T C^-1 diff + priors + error sentinel                  vega_interface.py:268-279,304,316-319
//   chi2-only evaluations (static quadratic form, include/vegamx.h: vmx_set_quadratic_form):
//   k_xi_assemble_quad / k_assemble_quad   bins of both components + the entry x' - x0'   (k_xi_bins + k_assemble)
//   k_gemm_nt44<12> / k_gemv1<., 2>        Q' (x' - x0'), contracted with x' - x0' in the epilogue    model.py:143-144 + vega_interface.py:316
//   k_chi2_parts / k_chi2_quad             the contraction's partial sums + constants + priors + sentinel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vegamx.h"
#include "vmx_plan.h"

#define VMX_NS 40            // scalars per (walker, pipeline)
#define VMX_PAD 32           // leading dimensions are multiples of 32 doubles
#define VMX_MAX_BB 32        // broadband terms per item and position
#define VMX_MAX_METALS 64
#define VMX_MAX_QUAD_COEF 64 // additive post-distortion broadband coefficients the quadratic form of chi2 can carry
#define VMX_MAX_GROUP 8      // independent products in one grouped launch (GemmGroup)

enum {
    S_BIAS1 = 0, S_BB1, S_BIAS2, S_BB2,
    S_UV_BG, S_UV_BP, S_UV_LAM, S_HE_BG, S_HE_LAM,
    S_HCD_B, S_HCD_BB, S_HCD_L0,
    S_GA, S_GB, S_EA, S_EB,
    S_AQ1, S_AQ2, S_AKV, S_AAV, S_ABV, S_AKP,
    S_VD1, S_VD2,
    S_AP, S_AT, S_DRP, S_EV1A, S_EV1B, S_EV2A, S_EV2B,
    S_RAD_S, S_RAD_A, S_RAD_L, S_RAD_D, S_RAD_IL, S_RAD_ID,       // (IL, ID: reciprocals of the lifetime and decrease lengths)
    S_NO_RULE                   // != 0: this walker lies outside the box the mu node rule was validated on -> plain midpoint loop
};

static inline int vmx_pad(int n) { return (n + VMX_PAD - 1) / VMX_PAD * VMX_PAD; }

struct PipeDev {
    vmx_pipe_desc d;
    int32_t n;            // bins
    int32_t n_pad;        // per-walker stride of the xi buffer (zero tail)
    int64_t coord_off;    // offset into the coordinate arrays
    int64_t xi_off;       // offset into the xi buffer
    double rp_absmin, rp_absmax, rt_min, rt_max;    // over the bins with r != 0 (bounds of the rescaled separations)
    int32_t col;          // column of this pipeline in the P_ell / spline-coefficient buffers (-1: it has none, see poly_basis)
    int32_t poly_basis;   // >= 0: its P(k,mu) is the Kaiser polynomial times static factors only, so its spline
                          // coefficients are a0 C0 + a1 C1 + a2 C2 with the static vectors C (EngineDev::poly_coef): no
                          // P(k,mu), no FFTLog column per walker
    int64_t poly_bins_off; // >= 0: ... and its coordinates are static too: offset of Y[3][n_pad] in EngineDev::poly_bins
    int32_t split_evol;   // new-bias-evolution: clnrelz holds tracer 1's ln(rel z), clnrelz2 tracer 2's
    int32_t tracers_swapped;   // vmx_add_pipeline put the caller's second tracer first (canonical order)
    // odd-multipole (relativistic / asymmetry) terms: static spline coefficients + amplitude slots
    int32_t odd_rel, odd_asy, odd_ncoef;
    int32_t odd_slot[5];
    int64_t odd_off;
    int64_t odd_dyn_off;  // >= 0: with a caller's spectrum (pk_direct) the walkers' own coefficient rows, EngineDev::odd_dyn
    int32_t odd_dyn_ld;   // ... [B][odd_dyn_ld] from this offset
    double odd_x0, odd_h;
};

struct BBTermDev {
    int32_t func, n_coef;
    int32_t slot[16];
    int64_t basis_off;    // [n_coef][n] doubles in the bb basis pool
};

struct MetalDev {
    vmx_metal_desc d;
    int64_t mat_off;      // offset of the dense metal matrix (-1: identity)
    int32_t mat_ld;
    int64_t xim_off;      // offset into the metal-product buffer (per-walker stride n_model_pad)
    const double* svec;   // static correlation [n_model] (fast_metals: frozen metal x metal term) instead of a pipeline
    const double* basis;  // static Kaiser basis [3][n_model]: xi = Y0 + (beta1 + beta2) Y1 + beta1 beta2 Y2
    const double* kron_a; // Kronecker-form metal matrix A (x) B (mat_off = 0 then marks "has a matrix"): A [n_rp][n_rp],
    const double* kron_b; //   B [n_rt][n_rt] or nullptr (identity)
    int32_t kron_nrp, kron_nrt;
};

struct ItemDev {
    vmx_item_desc d;
    int32_t n_model_pad, n_dist_pad, n_masked, n_masked_pad;
    int32_t n_metals;
    int32_t metal_begin;              // first index in the metal table
    int32_t n_bb[4];
    BBTermDev bb[4][VMX_MAX_BB];
    int64_t model_off;                // offset of this item in the per-walker model output
    const double* add_vec; int32_t add_slot; double add_default;   // additive template of the smooth component
    int64_t masked_off;               // offset in the concatenated masked vector (global-cov mode)
    // device arrays
    const double* dm;  int32_t dm_ld;         // distortion matrix or null
    // ... or in CSR form (scipy.sparse.csr_array of the reference, data.py:342-346): row pointers, column indices, values
    const int64_t* dm_ptr; const int32_t* dm_idx; const double* dm_val;
    const double* cinv; int32_t cinv_ld;      // inverse covariance or null (identity)
    const int32_t* inv_mask;                  // [n_dist] -> masked index or -1
    const double* data;                       // [n_masked]
    const double* mock_pool;                  // [n_mocks][n_masked] per-walker data vectors, or null
    double* vec;                              // [B][n_model_pad]   pre-distortion model
    double* dist;                             // [S][B][n_dist_pad] distortion product slabs
    double* res;                              // [B][n_masked_pad]  residual
    double* z;                                // [S][B][n_masked_pad] C^-1 residual slabs
    // static quadratic form of chi2 (k_assemble_quad / k_chi2_quad): x' = [pre-distortion vector ; additive
    // post-distortion broadband coefficients (1 + bao) c_j], expanded around the reference point x0'
    int32_t plain_pair;                       // peak and smooth component are plain spline sums (k_xi_assemble_quad)
    int32_t nq, nq_pad, q_na;                 // n_model + q_na entries, padded stride
    int32_t q_slot[VMX_MAX_QUAD_COEF];        // theta column of coefficient j
    const double* q_x0;                       // [nq_pad]  reference vector
    const double* q_lin;                      // [1 + n_mocks][nq_pad]  DM'^T S^T C^-1 r0 (row 0: data, row 1 + k: mock k)
    const double* q_c0;                       // [1 + n_mocks]  r0^T C^-1 r0
    double* q_x;                              // [B][nq_pad]  x' - x0'
    double* q_z;                              // [S][B][nq_pad]  slabs of L' (x' - x0')
    // factored form of the same chi2 (vmx_set_quadratic_form_kind): chi2 = || u0 - F dx ||^2 with C^-1 = U^T U,
    // F = U S DM' [n_masked][nq_pad], u0 = U r0 - the cheaper one when the model grid is much finer than the data grid
    const double* q_u0;                       // [1 + n_mocks][n_masked_pad]  U (d - S DM' x0')
    double* q_y;                              // [S][B][n_masked_pad]  slabs of F (x' - x0')
};


// Bins that arrive pre-summed: a group kernel adds the contributions of up to four (static-coordinate pipelines: sixteen)
// pipelines of an item - each times its walker's factor: 1, bao_amp or the metal pair's bias product - and stores ONE array,
// in the bins slot of the group's first pipeline.  assemble_bin adds those arrays and skips the members.
struct ItemSums {
    int32_t n_arrays, n_pad;
    int32_t core_peak, core_smooth;     // 1: the item's peak / smooth component is inside one of the arrays
    uint64_t metal_mask;                // bit m: metal pair m of the item is
    int64_t off[8];                     // offsets of the arrays in the bins buffer (per-walker stride n_pad)
};

struct EngineDev {
    // template
    int32_t nk, nkp, n_mu, n_ell;
    int32_t pad_l, pad_r;       // fht_extrap: power-law pads of the FFTLog input behind the nk samples of a P_ell row
    const int32_t* pipe_active; // [n_active] the pipeline of a P_ell / coefficient column
    // mu-quadrature nodes appended to every mu-indexed table (mu, lnmu, sq1mmu2, gk, xtab): rows [n_mu, n_rows).  With
    // weights node_w they reproduce the n_mu-point midpoint sums of the reference (power_spectrum.py:76-77, pktoxi.py:138)
    // from the first mu_lo and the last mu_hi midpoints plus n_extra nodes - see vmx_set_mu_quadrature in vegamx.h
    int32_t n_rows, n_extra, mu_lo, mu_hi;
    const double* node_w;       // [n_extra]
    double k_node_max;          // k tiles up to this wavenumber use the node rule (0: the midpoint sums everywhere)
    // applicability guard of the node rule (vmx_set_mu_rule_box): a walker with theta[rule_slot[i]] outside
    // [rule_lo[i], rule_hi[i]] - the box the rule is validated on, tests/test_mu_quadrature.py - takes the plain loop
    const int32_t* rule_slot; const double* rule_lo; const double* rule_hi; int32_t n_rule;
    const double* k;            // [nkp]
    const double* pklin;        // [3][nkp]
    const double* delta2;       // [nkp]
    const double* mu;           // [n_mu]
    const double* mu_img;       // [n_mu] {mu^2, mu^4}, [n_extra] {mu, mu^2, mu^4, w}: the LDS image of k_pk_tab2's node tables
    const double* mu_img_w;     // ... of k_pk_w's: [n_mu] {mu^2, mu^4}, [n_extra] {mu^2, mu^4, mu^6, w}
    const double* sq1mmu2;      // [n_mu] sqrt(1 - mu^2)
    const double* lnmu;         // [n_mu] ln(mu)
    const double* wl;           // [4][n_mu] L_ell(mu) (2 ell + 1) / n_mu
    const double* fv_x; const double* fv_f; int32_t fv_n;   // Voigt-profile HCD table
    const double* gk;           // [tables][n_mu][nkp]
    double* xtab;               // [arinyo groups][2][n_rows][nkp]  tables of the batch's first walker (see xtab_level)
    const int32_t* const_slots; int32_t n_const_slots;   // parameters asserted constant across the batch
    // Table level of this evaluation (0: none).  1: D_NL^power * G - the batch shares its Arinyo parameters.  2: the batch
    // also shares every Gaussian factor (smoothing, Gaussian velocity dispersion, peak broadening), so the table holds
    // D_NL^power * G * exp(-k^2 (gb + (ga - gb) mu^2)) and a second one the same for the peak partner: the HCD factor is
    // then the only exponential left per walker and (k, mu).
    // A table only changes with those parameters: k_xtab compares them with the ones the table was built from (xtab_key
    // [tables][VMX_XTAB_KEY], written by the chi2 kernel of the evaluation that built it), recomputes a stale table and
    // leaves a current one alone.  xtab_k [tables][4][nkp]: per wavenumber the Arinyo part of e0, e2, the error flag, the (level-2) bound of the exponents.
    const int32_t* xtab_pipe; const int32_t* xtab_partner; int32_t n_xtab; int32_t xtab_level; double* xtab_key; double* xtab_k;
    int32_t xtab_block0;        // k_prologue: first of the table blocks behind the walkers' (0: k_xtab is its own launch)
    const double* gk_mom;       // [tables + 1][6][nkp]  sum_j mu_j^(2n) G(k, mu_j); last table: G = 1
    int32_t n_gk;
    // fftlog / spline
    int32_t n_coef, ncp;        // coefficients per ell, padded
    int32_t extrapolate;        // splines extend beyond their knots (legacy transform) instead of flagging
    double x0[VMX_MAX_ELL], h[VMX_MAX_ELL], xlast[VMX_MAX_ELL], inv_h[VMX_MAX_ELL];
    int same_grid;              // the multipoles' ln r grids coincide (x0, h equal over ell)
    // pipelines
    int32_t n_pipe;
    int32_t n_active;           // pipelines with a column in pl / coef (the others carry a static coefficient basis)
    const double* poly_coef;    // [n_ell][n_static][3][ncp]  FFTLog o spline of the Kaiser-basis spectra of those
    const double* poly_bins;    // per static-coordinate pipeline [3][n_pad]: the basis evaluated on its bins
    int32_t n_static;
    const PipeDev* pipes;
    const double* crp; const double* crt;       // r mu and r sqrt(1 - mu^2) of every bin (static)
    const double* cr; const double* cmu; const double* cz; const double* crelz; const double* clnrelz; const double* cgrowth;
    const double* clnrelz2;     // second tracer's ln(rel z) (same layout; equals clnrelz unless split_evol)
    // items
    int32_t n_items;
    const ItemDev* items;
    const MetalDev* metals;
    int32_t n_metals_total;
    const double* bb_basis;
    const double* odd_coef;
    const double* odd_dyn;      // direct_pk: per-walker odd-multipole spline coefficients (PipeDev::odd_dyn_off)
    const double* sn_a; int32_t sn_n; double sn_tau0, sn_dtau;     // UV shot-noise A(tau) table
    // priors
    int32_t n_priors;
    const int32_t* prior_slot; const double* prior_mean; const double* prior_sigma;
    // batch buffers
    int32_t n_params;
    const double* theta;        // [B][n_params]
    // theta_host: the walkers are elsewhere - mapped pinned host memory (small host batches skip the staging copies) or the
    // caller's device buffer, read in place (large batches, eager launches): k_prologue reads them there and leaves a device
    // copy in `theta_copy` (= theta) for the later kernels; k_chi2 stores its results to chi2_host / status_host too
    const double* theta_host; double* theta_copy; double* chi2_host; int32_t* status_host;
    int32_t src_host;           // theta_host is mapped host memory (a batch of at most 8 walkers)
    double* scal;               // [B][n_pipe][VMX_NS]
    double* metal_bias;         // [B][3][n_metals_total]: bias product x multiplicity, beta1 + beta2, beta1 * beta2
    double beta_override; int32_t beta_override_on;      // set-up hook: betas of the bias-free metal pipelines
    double* pl;                 // [n_ell][n_active][B][nkp]   (pipeline-major columns: column = PipeDev::col * B + walker)
    double* coef;               // [n_ell][n_active][B][ncp]
    double* xi;                 // per pipeline [B][n]
    // pre-summed bins (k_xi_bins_group / k_xi_bins_static_group; null / 0: every contribution of an item in its own array)
    const struct ItemSums* sums; int32_t sums_on;
    double* xim;                // metal matrix products
    double* model;              // [B][model_size]
    double* chi2;               // [B]
    int32_t* status;            // [B]
    // direct_pk mode (model.py:188-207): the model is the smooth pipeline of every item evaluated with a linear spectrum
    // supplied per walker (e.g. by a Boltzmann code); no peak component, and additive terms enter once
    const double* pk_direct;    // [B][nkp] or null
    int32_t* k_live;            // [0] wavenumbers >= *k_live have P_ell = 0 for every walker and pipeline of the batch;
                                // [1] wavenumbers < k_live[1] sat in tiles that took the mu node rule (a statistic);
                                // [2], [3] the spline-coefficient window of the last evaluation (a statistic)
    unsigned long long* pk_trace;   // debugging aid (VMX_PK_TRACE): per block of k_pk_tab2 {start, end (100 MHz ticks), hw id, xcc id}
    int32_t* coef_win;          // [2] first / last spline coefficient any bin of the batch reads (k_prologue; reset by k_chi2)
    const int32_t* mock_index;  // [B] row of the mock pool used as data by walker b, -1: the item's data vector
    int32_t model_size;
    // global covariance mode
    const double* gcinv; int32_t g_n, g_ld; double* gres; double* gz;
    // zero-copy host evaluations: the last kernel publishes done_seq in mapped host memory after its results, so the
    // host can wait on that word instead of on the stream (null: no such wait)
    volatile int64_t* done_host; int64_t done_seq;
};

// split-K slab counts of a product per item (+ the global one)
struct SlabInfo { int32_t z[16]; int32_t g; };
#define CHI2_THREADS 1024                        // k_chi2 / k_chi2_quad: one block per walker

// ------------------------------------------------------------------------------------------------
// prologue: parameters -> scalars
// ------------------------------------------------------------------------------------------------
#define VMX_CAS __attribute__((address_space(4)))      // constant address space: uniform reads become scalar loads
__device__ inline double th(const double* t, int slot, double dflt) { return slot >= 0 ? t[slot] : dflt; }

// (TR: vmx_tracer in any address space)
template <class TR>
__device__ inline void tracer_bias_beta(const double* t, TR& tr, double gr, double& bias, double& beta)
{
    // reference vega/utils.py:45-82
    const bool has_bias = tr.bias_slot >= 0, has_eta = tr.bias_eta_slot >= 0, has_beta = tr.beta_slot >= 0;
    bias = has_bias ? t[tr.bias_slot] : 0.0;
    beta = has_beta ? t[tr.beta_slot] : 0.0;
    const double eta = has_eta ? t[tr.bias_eta_slot] : 0.0;
    if (!has_bias) bias = eta * gr / beta;
    if (!has_beta) beta = eta * gr / bias;
}

// parameter-level blinding (vega_interface.py:389-421, utils.py:375-393): theta -> scale * theta + shift, in place;
// tr = [scale[n_params], shift[n_params]].  A unit scale is a plain addition, as the reference's `+=`.
__global__ void k_theta_affine(double* theta, const double* tr, int n_params, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = i % n_params;
    const double s = tr[p], sh = tr[n_params + p], t = theta[i];
    theta[i] = s == 1.0 ? t + sh : s * t + sh;
}

// A single walker through the host entry point travels in the kernel arguments (no read of mapped host memory over
// PCIe at the head of a latency-bound chain); up to this many parameters.
#define VMX_THETA_ARG_MAX 160
#define PRO_T 64           // threads of a k_prologue block = walkers of one slot
struct ThetaArg { double v[VMX_THETA_ARG_MAX]; };

// exponents of the Gaussian factors exp(-k^2 (gb + (ga - gb) mu^2)) of a pipeline: peak broadening (reference
// power_spectrum.py:395-402), Gaussian smoothing (:526-556) and Gaussian velocity dispersion.  One function for k_prologue's
// scalars and for the key of the level-2 tables (the table blocks form walker 0's values themselves).
template <class PD>
__device__ __forceinline__ void gauss_exponents(const double* t, PD& d, double gr, double& ga, double& gb)
{
    ga = 0.0; gb = 0.0;
    if (d.peak_nl) {
        double sp, st;
        if (d.sigma_nl_par_slot >= 0 && d.sigma_nl_per_slot >= 0) { sp = t[d.sigma_nl_par_slot]; st = t[d.sigma_nl_per_slot]; }
        else if (d.sigma_nl_par_slot >= 0) { sp = t[d.sigma_nl_par_slot]; st = sp / (1.0 + gr); }
        else { st = t[d.sigma_nl_per_slot]; sp = st * (1.0 + gr); }
        ga += 0.5 * sp * sp; gb += 0.5 * st * st;
    }
    for (int i = 0; i < d.n_smooth; ++i) {
        const double sp = t[d.smooth_par_slot[i]], st = t[d.smooth_per_slot[i]];
        ga += d.smooth_weight[i] * sp * sp; gb += d.smooth_weight[i] * st * st;
    }
    if (d.vd_kind == VMX_VD_GAUSS)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (d.tracer[q].discrete) { const double sg = t[d.tracer[q].vd_sigma_slot]; ga += 0.25 * sg * sg; }
}

#define XTAB_ROWS 8         // table rows per 256 threads of a table block
__device__ __forceinline__ double vmx_log(double x);
__device__ void xtab_body(const EngineDev& D, int xtab, int bx, int by, int rows_per_block, const double* t0);

template <bool BYVAL>
__device__ __forceinline__ void prologue_body(const EngineDev& D, int B)
{
    // where the walkers are: the kernel arguments (single evaluation), mapped host memory (small host batches: one PCIe round
    // trip per block, all blocks at once), the caller's device buffer (read in place) or the engine's own copy
    // (BYVAL: ThetaArg is the kernel's FIRST argument and is read where it lies, at the start of the kernel-argument segment -
    // global memory; taken by address as a parameter the compiler copies it to scratch, in every wave of the grid)
    const double* src_all = BYVAL ? (const double*)__builtin_amdgcn_kernarg_segment_ptr() : D.theta_host ? D.theta_host : D.theta;
    if (D.xtab_block0 > 0 && (int)blockIdx.x >= D.xtab_block0) {
        // The blocks behind the walkers' own check the level-2 / level-1 tables against walker 0 (k_xtab's work, in this launch:
        // a launch of its own costs ~5 us of an evaluation whose tables are current nearly always).  They read walker 0 from
        // the source the walker blocks read it from; nothing of this launch's output.
        const int rows_per_block = XTAB_ROWS * (256 / PRO_T);
        const int nbx = (D.nkp + PRO_T - 1) / PRO_T, nby = (D.n_rows + rows_per_block - 1) / rows_per_block;
        const int q = (int)blockIdx.x - D.xtab_block0;
        xtab_body(D, q / (nbx * nby), q % nbx, (q / nbx) % nby, rows_per_block, src_all);
        return;
    }
    // block = (slot, 64 walkers).  Slots 0 .. n_pipe-1 fill the P(k,mu) half of one pipeline's scalars (amplitudes, UV, HCD,
    // Gaussian exponents, Arinyo, velocity dispersion: S_BIAS1 .. S_VD2), slots n_pipe .. 2 n_pipe-1 its xi half (scale
    // parameters, the coefficient window, evolution, radiation, the mu rule's guard: S_AP .. S_NO_RULE), slot 2 n_pipe the
    // walker-level values.  A wave serves ONE slot: every branch on the pipeline's descriptor is uniform (with the walker's
    // slots side by side in a wave it walked every pipeline's path in turn), the descriptor reads are broadcasts, and a block
    // fetches the code of its half only (the kernel's time is its instruction fetch).
#ifdef VMX_EXP_PRO_TRACE
#define PRO_STAMP(I) do { if (threadIdx.x == 0 && D.pk_trace) D.pk_trace[8 * blockIdx.x + (I)] = wall_clock64(); } while (0)
#else
#define PRO_STAMP(I) do {} while (0)
#endif
    PRO_STAMP(0);
    const int n_chunk = (B + PRO_T - 1) / PRO_T;
    const int slot = (int)blockIdx.x / n_chunk, b_first = ((int)blockIdx.x % n_chunk) * PRO_T;
    const int b = b_first + (int)threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { D.k_live[0] = 0; D.k_live[1] = 0; }
    // The slot's pipeline descriptor is read dozens of times ahead of data-dependent branches.  It is static and the slot is
    // the block's: through the constant address space those reads are scalar loads (batched by the compiler, cached) and the
    // branches scalar branches - from LDS or global memory every one of them was a vector round trip the next read waited for.
    const bool half_b = slot >= D.n_pipe && slot < 2 * D.n_pipe, walker_slot = slot == 2 * D.n_pipe;
    const int p = min(half_b ? slot - D.n_pipe : slot, D.n_pipe - 1);
    const VMX_CAS PipeDev& P = *(const VMX_CAS PipeDev*)(D.pipes + p);
    // LDS: [constant-slot list with walker 0's values][rule box][the block's walkers] - the lists the walker loops below walk
    // (from global memory every element of theirs is a dependent round trip inside the loop) and the parameters
    extern __shared__ double s_dyn[];
    const int n_cs = D.n_const_slots, n_ru = D.n_rule;
    double* s_c0 = s_dyn;                       // [n_cs] walker 0's values at the constant slots
    double* s_cs = s_c0 + n_cs;                 // [n_cs] the slots
    double* s_ru = s_cs + n_cs;                 // [3][n_ru] slot, lower and upper bound of the rule's box
    double* s_theta = s_ru + 3 * n_ru;          // [PRO_T][n_params | 1] the block's walkers (odd stride: the lanes of a wave read
    const int t_ld = D.n_params | 1;            // the same parameter of 64 walkers)
    // This kernel's blocks are single waves that run their code once: its time is the instruction FETCH (every 64 bytes of
    // straight-line code is an instruction-cache miss, ~2.5 ns per instruction against 0.5 in a hot loop - block time stamps,
    // round 4), so the loops below stay rolled and short.
    const int rows = min(B, b_first + PRO_T) - b_first;
    {
        // the walkers' rows.  From device memory: direct global -> LDS copies, one wave instruction per row and 64 dwords of it,
        // all in flight together (no staging registers: a rolled loop).  The few walkers of a batch in the kernel arguments or in
        // mapped host memory: ordinary loads, four per thread in flight (one trip over PCIe for up to 256 parameters).
        if (BYVAL || D.src_host) {
            const int count = rows * D.n_params;
            const double* src = src_all + (size_t)b_first * D.n_params;
            for (int i0 = threadIdx.x; i0 < count; i0 += 4 * PRO_T) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * PRO_T; v[u] = i < count ? src[i] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * PRO_T; if (i < count) s_theta[(i / D.n_params) * t_ld + i % D.n_params] = v[u]; }
            }
        } else {
            const int nd = 2 * D.n_params;              // dwords of a row
            const char* src = (const char*)(src_all + (size_t)b_first * D.n_params);
            for (int c0 = 0; c0 < nd; c0 += PRO_T)
                if (c0 + (int)threadIdx.x < nd)
                    for (int r = 0; r < rows; ++r)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((size_t)r * nd + c0 + threadIdx.x) * 4),
                                                         (__attribute__((address_space(3))) void*)((char*)(s_theta + r * t_ld) + c0 * 4), 4, 0, 0);
        }
        // the lists (slot n_pipe: the constant slots with walker 0's values; the others: the rule's box)
        if (walker_slot)
            for (int i = threadIdx.x; i < n_cs; i += PRO_T) { const int q = D.const_slots[i]; s_cs[i] = (double)q; s_c0[i] = src_all[q]; }
        else if (half_b)
            for (int i = threadIdx.x; i < n_ru; i += PRO_T) { s_ru[i] = (double)D.rule_slot[i]; s_ru[n_ru + i] = D.rule_lo[i]; s_ru[2 * n_ru + i] = D.rule_hi[i]; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PRO_STAMP(7);
    // (a surplus lane of the last chunk shadows the chunk's last walker and stores nothing: the wave reductions below see 64 lanes)
    const bool live = b < B;
    const double* t = s_theta + (size_t)min((int)threadIdx.x, B - 1 - b_first) * t_ld;
    __syncthreads();
    PRO_STAMP(1);
    if (slot == 0 && (BYVAL || D.theta_host)) {
        // the copy the later kernels use leaves from LDS
        double* dst = D.theta_copy + (size_t)b_first * D.n_params;
        for (int r = 0; r < rows; ++r)
            for (int c = threadIdx.x; c < D.n_params; c += PRO_T) dst[r * D.n_params + c] = s_theta[r * t_ld + c];
    }

    if (!walker_slot) {
        const auto& d = P.d;
        // the scalars are collected in registers and stored at the end: with stores in between, the compiler has to keep
        // every parameter load behind the previous store (the pointers may alias) - ~30 dependent L2 round trips
        double s[VMX_NS];
#pragma unroll
        for (int i = 0; i < VMX_NS; ++i) s[i] = 0.0;
        double* out = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;

      if (!half_b) {
        const double gr = th(t, d.growth_rate_slot, d.growth_rate_default);
        double b1, be1, b2, be2;
        tracer_bias_beta(t, d.tracer[0], gr, b1, be1);
        if (d.same_tracer) { b2 = b1; be2 = be1; }
        else tracer_bias_beta(t, d.tracer[1], gr, b2, be2);

        if (d.fast_metals && D.beta_override_on) { be1 = D.beta_override; be2 = D.beta_override; }
        const bool eff1 = d.tracer[0].is_lya && (d.uvb || d.heii || d.hcd_model != VMX_HCD_NONE);
        const bool eff2 = d.tracer[1].is_lya && (d.uvb || d.heii || d.hcd_model != VMX_HCD_NONE);
        if (d.fast_metals && !eff1) { s[S_BIAS1] = 1.0; s[S_BB1] = be1; }
        else { s[S_BIAS1] = b1; s[S_BB1] = b1 * be1; }
        if (d.fast_metals && !eff2) { s[S_BIAS2] = 1.0; s[S_BB2] = be2; }
        else { s[S_BIAS2] = b2; s[S_BB2] = b2 * be2; }

        if (d.uvb) { s[S_UV_BG] = t[d.bias_gamma_slot]; s[S_UV_BP] = t[d.bias_prim_slot]; s[S_UV_LAM] = t[d.lambda_uv_slot]; }
        if (d.heii) { s[S_HE_BG] = t[d.bias_gamma_e_slot]; s[S_UV_BP] = t[d.bias_prim_slot]; s[S_HE_LAM] = t[d.lambda_heii_slot]; }
        if (d.hcd_model != VMX_HCD_NONE) {
            const double bh = t[d.bias_hcd_slot], beh = t[d.beta_hcd_slot];
            s[S_HCD_B] = bh; s[S_HCD_BB] = bh * beh; s[S_HCD_L0] = th(t, d.l0_hcd_slot, d.l0_default);
        }

        double ga, gb;
        gauss_exponents(t, d, gr, ga, gb);
        if (d.exp_par_slot >= 0) { const double a = t[d.exp_par_slot], c = t[d.exp_per_slot]; s[S_EA] = a * a; s[S_EB] = c * c; }
        if (d.vd_kind != VMX_VD_NONE && d.vd_kind != VMX_VD_GAUSS)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const auto& tr = d.tracer[q];
                if (!tr.discrete) continue;
                const double sg = t[tr.vd_sigma_slot];
                s[q == 0 ? S_VD1 : S_VD2] = sg * sg;
            }
        s[S_GA] = ga; s[S_GB] = gb;

        if (d.nl_model == VMX_NL_ARINYO) {
            s[S_AQ1] = t[d.arinyo_slot[0]]; s[S_AQ2] = th(t, d.arinyo_slot[1], 0.0);
            s[S_AKV] = t[d.arinyo_slot[2]]; s[S_AAV] = t[d.arinyo_slot[3]];
            s[S_ABV] = t[d.arinyo_slot[4]]; s[S_AKP] = t[d.arinyo_slot[5]];
        }
        PRO_STAMP(2);
        if (live)
#pragma unroll
            for (int i = 0; i < S_AP; ++i) out[i] = s[i];
      } else {
        // scale parameters (reference scale_parameters.py:162-230)
        double ap = 1.0, at = 1.0;
        if (d.scale_mode == VMX_SCALE_AP_AT) { ap = t[d.scale_slot[0]]; at = t[d.scale_slot[1]]; }
        else if (d.scale_mode == VMX_SCALE_AISO_EPS) {
            const double aiso = t[d.scale_slot[0]], eps = t[d.scale_slot[1]];
            ap = aiso * (1.0 + eps) * (1.0 + eps); at = aiso / (1.0 + eps);
        } else if (d.scale_mode == VMX_SCALE_PHI_ALPHA) {
            const double phi = t[d.scale_slot[0]], alpha = t[d.scale_slot[1]];
            ap = alpha / sqrt(phi); at = alpha * sqrt(phi);
        }
        s[S_AP] = ap; s[S_AT] = at;
        s[S_DRP] = th(t, d.drp_slot, 0.0);
        {
            // Spline coefficients this (walker, pipeline) can read: r'^2 = ap^2 (rp + drp)^2 + at^2 rt^2 over its bins is
            // bounded by the extremes of |rp|, rt (|rp + drp| lies in [max(0, |rp| - |drp|), |rp| + |drp|]); the FFTLog
            // product computes the rows inside the batch's window only.  NaN / out-of-range inputs open the window fully.
            const double adrp = fabs(s[S_DRP]);
            const double lo_rp = fmax(P.rp_absmin - adrp, 0.0), hi_rp = P.rp_absmax + adrp;
            const double r2lo = ap * ap * lo_rp * lo_rp + at * at * P.rt_min * P.rt_min;
            const double r2hi = ap * ap * hi_rp * hi_rp + at * at * P.rt_max * P.rt_max;
            int jlo = 0, jhi = D.n_coef - 1;
            if (r2lo > 0.0 && r2hi >= r2lo && r2hi < 1e300) {
                const double xlo = 0.5 * vmx_log(r2lo), xhi = 0.5 * vmx_log(r2hi);
                double ulo = 1e300, uhi = -1e300;
                for (int e = 0; e < d.n_ell; ++e) {
                    ulo = fmin(ulo, (xlo - D.x0[e]) * D.inv_h[e]);
                    uhi = fmax(uhi, (xhi - D.x0[e]) * D.inv_h[e]);
                }
                if (ulo > -1e9 && uhi < 1e9) {
                    jlo = max(0, (int)floor(ulo) - 2);
                    jhi = min(D.n_coef - 1, (int)floor(uhi) + 5);
                }
            }
            if (P.odd_rel || P.odd_asy || D.extrapolate) { jlo = 0; jhi = D.n_coef - 1; }
            // one pair of global atomics per wave, not per thread (2 x 1280 atomics on two addresses took ~20 us): the lanes'
            // windows meet in a butterfly first (as LDS atomics the compiler reduces them in a 64-step scalar loop)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { jlo = min(jlo, __shfl_xor(jlo, m)); jhi = max(jhi, __shfl_xor(jhi, m)); }
            if (threadIdx.x == 0) {
                atomicMin(&D.coef_win[0], jlo);
                atomicMax(&D.coef_win[1], jhi);
            }
        }
        PRO_STAMP(3);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const auto& tr = d.tracer[q];
            double a, c = 0.0;
            if (tr.evol_kind == VMX_EVOL_CROOM) { a = t[d.croom_slot[0]]; c = t[d.croom_slot[1]]; }
            else a = t[tr.alpha_slot];
            s[q == 0 ? S_EV1A : S_EV2A] = a; s[q == 0 ? S_EV1B : S_EV2B] = c;
        }
        if (d.radiation) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s[S_RAD_S + i] = t[d.rad_slot[i]];
            s[S_RAD_IL] = 1.0 / s[S_RAD_L]; s[S_RAD_ID] = 1.0 / s[S_RAD_D];
        }
        {
            // the node rule of the mu sums is trusted inside the parameter box it was validated on; outside (or NaN) the
            // walker's P(k,mu) blocks run the reference's midpoint loop itself
            bool outside = false;
            for (int q = 0; q < n_ru; ++q) {
                const double v = t[(int)s_ru[q]];
                outside = outside || !(v >= s_ru[n_ru + q] && v <= s_ru[2 * n_ru + q]);
            }
            s[S_NO_RULE] = outside ? 1.0 : 0.0;
            if (outside && p == 0 && live) atomicAdd(D.k_live + 4, 1);          // (a statistic: walkers that left the box, cumulative)
        }
        PRO_STAMP(4);
        if (live)
#pragma unroll
            for (int i = S_AP; i < VMX_NS; ++i) out[i] = s[i];
        PRO_STAMP(5);
      }
    }

    // metal bias products (reference metals.py:295-313, :331-332) and the Kaiser coefficients of the static-basis metals
    // (the walker's 2 n_pipe + 1 threads share the metals: one serial chain of 19 bias / beta look-ups per walker was a third of
    // this kernel at 23 pipelines)
    if (!live) return;
    for (int m = slot; m < D.n_metals_total; m += 2 * D.n_pipe + 1) {
        const vmx_metal_desc& d = D.metals[m].d;
        double f = d.multiplicity;
        const double gr = th(t, d.growth_rate_slot, d.growth_rate_default);
        double b1, be1, b2, be2;
        tracer_bias_beta(t, d.tracer[0], gr, b1, be1);
        if (d.same_tracer) { b2 = b1; be2 = be1; } else tracer_bias_beta(t, d.tracer[1], gr, b2, be2);
        if (d.apply_bias) f *= b1 * b2 * th(t, d.extra_bias_slot, 1.0);
        if (d.amplitude_slot >= 0) f *= t[d.amplitude_slot];
        double* mb = D.metal_bias + (size_t)b * 3 * D.n_metals_total;
        mb[m] = f;
        mb[D.n_metals_total + m] = be1 + be2;
        mb[2 * D.n_metals_total + m] = be1 * be2;
    }
    if (!walker_slot) return;

    int st = 0;
    for (int q = 0; q < n_cs; ++q)
        if (t[(int)s_cs[q]] != s_c0[q]) st = VMX_STATUS_NOT_CONSTANT;
    D.status[b] = st;
    D.chi2[b] = 0.0;
    PRO_STAMP(6);
}

__global__ __launch_bounds__(PRO_T) void k_prologue(EngineDev D, int B) { prologue_body<false>(D, B); }
__global__ __launch_bounds__(PRO_T) void k_prologue_byval(ThetaArg, EngineDev D, int B) { prologue_body<true>(D, B); }

// ------------------------------------------------------------------------------------------------
// static G(k, mu) table
// ------------------------------------------------------------------------------------------------
__global__ void k_gk_table(double* out, const double* k, const double* mu, int nk, int nkp, int n_mu,
                           double bs_rp, double bs_rt, double mock_rp, double mock_rt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= nkp || j >= n_mu) return;
    double g = 0.0;
    if (i < nk) {
        const double kk = k[i], m = mu[j];
        const double kpar = kk * m;
        const double ktr = kk * sqrt(fmax(1.0 - m * m, 0.0));
        g = 1.0;
        // (a quadrature node sits at mu = 1, where k_trans = 0: sinc(0) = 1; the midpoint grid never gets there)
        if (bs_rp != 0.0) { const double x = kpar * bs_rp / 2.0; g = g * (x != 0.0 ? sin(x) / x : 1.0); }
        if (bs_rt != 0.0) { const double x = ktr * bs_rt / 2.0; g = g * (x != 0.0 ? sin(x) / x : 1.0); }
        // mock binning: a second factor of the same form (power_spectrum.py:143-160)
        double gm = 1.0;
        if (mock_rp != 0.0) { const double x = kpar * mock_rp / 2.0; gm = gm * (x != 0.0 ? sin(x) / x : 1.0); }
        if (mock_rt != 0.0) { const double x = ktr * mock_rt / 2.0; gm = gm * (x != 0.0 ? sin(x) / x : 1.0); }
        g *= gm;
    }
    out[(size_t)j * nkp + i] = g;
}

// even mu-moments of a G(k, mu) table (or of 1 when table == nullptr): out[n][i] = sum_j mu_j^(2n) G[j][i]
__global__ void k_gk_moments(double* out, const double* table, const double* mu, int nkp, int n_mu)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nkp) return;
    double m[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < n_mu; ++j) {
        const double mu2 = mu[j] * mu[j];
        double v = table ? table[(size_t)j * nkp + i] : 1.0;
        for (int n = 0; n < 6; ++n) { m[n] += v; v *= mu2; }
    }
    for (int n = 0; n < 6; ++n) out[(size_t)n * nkp + i] = m[n];
}

// ------------------------------------------------------------------------------------------------
// P(k, mu) evaluation fused with the Legendre projection
//   block = KT wavenumbers x MS mu-slices (KT * MS = 256); grid = (walkers, pipelines, k tiles):
//   walkers vary fastest so that co-resident blocks read the same slice of the G(k,mu) table from L2.
//   Each thread owns one k and walks mu_j = (j + 1/2)/n_mu for j = ms, ms + MS, ...; every factor of
//   the reference's product is rebuilt per walker (no allclose-keyed caches).
// ------------------------------------------------------------------------------------------------

// d = a * b + c as one v_fma_f64.  hipcc otherwise lowers a Horner step with a constant addend to
// v_mov_b64 + v_fmac_f64 (two VALU slots); plain VALU results are interlocked in hardware, so the
// statement needs no wait states of its own.
__device__ __forceinline__ double vmx_fma(double a, double b, double c)
{
    // (the constant addend in a scalar register pair: the exponential's coefficients then cost no vector registers)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// exp(x) for x <= 709, |relative error| ~ 2e-16: n = rint(x log2 e), r = x - n ln 2 in two pieces,
// degree-12 Taylor polynomial on |r| <= ln(2)/2, scaled by 2^n.  Very negative arguments give 0
// (v_cvt_i32_f64 saturates and v_ldexp_f64 underflows to zero).
__device__ __forceinline__ double vmx_exp(double x)
{
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;            // 1/12!
    p = vmx_fma(p, r, 2.50521083854417187751e-08);    // 1/11!
    p = vmx_fma(p, r, 2.75573192239858906526e-07);    // 1/10!
    p = vmx_fma(p, r, 2.75573192239858906526e-06);    // 1/9!
    p = vmx_fma(p, r, 2.48015873015873015873e-05);    // 1/8!
    p = vmx_fma(p, r, 1.98412698412698412698e-04);    // 1/7!
    p = vmx_fma(p, r, 1.38888888888888888889e-03);    // 1/6!
    p = vmx_fma(p, r, 8.33333333333333333333e-03);    // 1/5!
    p = vmx_fma(p, r, 4.16666666666666666667e-02);    // 1/4!
    p = vmx_fma(p, r, 1.66666666666666666667e-01);    // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// 1/sqrt(x) for x >= 1: hardware estimate plus one third-order correction (no special cases arise)
__device__ __forceinline__ double vmx_rsqrt(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// ln(x) for finite x > 0 (no special cases), |error| < 1 ulp: x = 2^k m with m in [sqrt(1/2), sqrt(2)), f = m - 1,
// s = f / (2 + f), ln m = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 minimax polynomial R of the classic
// formulation (coefficients Lg1..Lg7 of fdlibm's e_log.c); ~40 instructions against the library's ~100.
__device__ __forceinline__ double vmx_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const int up = m < 0.70710678118654752440 ? 1 : 0;
    m = ldexp(m, up); k -= up;                              // [sqrt(1/2), sqrt(2))
    const double f = m - 1.0, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    double sq = f * r;
    sq = fma(fma(-d, sq, f), r, sq);                        // s = f / d
    const double z = sq * sq, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define PK_REANCHOR 64      // steps between exact re-evaluations of the exponential recurrences
// Bounds on the exponent of a wavenumber's largest (k, mu) value: below VMX_PK_DEAD a whole tile is skipped (its multipoles
// are exact zeros; exp(-200) ~ 1e-87 leaves > 60 decades of margin for the amplitudes); below VMX_PK_NEGLIGIBLE
// (exp(-100) ~ 4e-44 of the unsuppressed spectrum - beyond the last bit of any xi) a wavenumber no longer decides
// whether its tile may take the mu node rule.
#define VMX_PK_DEAD (-200.0)
#define VMX_PK_NEGLIGIBLE (-100.0)

// One work group of the P(k,mu) stage: a pipeline and, when the item's peak component differs from its
// smooth component only by the peak non-linear broadening (power_spectrum.py:163-164), that peak
// pipeline as a partner evaluated in the same pass.  `variant` selects a compile-time specialisation
// of the mu loop (0 = generic).
struct PkGroup { int32_t pipe; int32_t peak_partner; int32_t variant; int32_t n_members; int32_t member_off;
                 int32_t xtab; };     // xtab: index of this group's D_NL * G table (-1: none)

enum { PKV_GENERIC = 0, PKV_AUTO_CORE = 1, PKV_CROSS_CORE = 2, PKV_PLAIN_SAME = 3, PKV_PLAIN_PAIR = 4, PKV_PLAIN_PAIR_VD = 5,
       PKV_POLY = 6, PKV_SHARED_W = 7 };
// Kaiser-term modes of the specialised loops
enum { KM_SAME_HCD = 0, KM_SAME_PLAIN = 1, KM_FIRST_HCD = 2, KM_BOTH_PLAIN = 3 };

struct PkThread {
    // per-thread (one wavenumber) constants of the mu loop
    double k, c0_1, c1_1, c0_2, c1_2, hb, hbb, L0, e0, e1, e2, vd1, vd2, ea, eb, mc_kvel;
    double p0, p1, pq, Fq;
    double mock_c;              // k / 2 x the walker's line-of-sight mock bin (mock_los), 0: none
    const double* gk;           // this thread's first table entry (row ms of its column)
    size_t gk_stride;           // MS rows
    size_t gk_row;              // one row
    bool hcd1, hcd2, div1, div2, same, arinyo, rogers, sinc, fvoigt, has_exp, mcdonald, paired, has_vd1, has_vd2, noexp;
    const double* fv_x; const double* fv_f; int fv_n;
};

// np.interp(x, xp, fp, left=1, right=0) on an increasing table (power_spectrum.py:378)
__device__ inline double fvoigt_interp(double x, const double* xp, const double* fp, int n)
{
    if (x < xp[0]) return 1.0;
    if (x > xp[n - 1]) return 0.0;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
    const double slope = (fp[hi] - fp[lo]) / (xp[hi] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

// mu loop of one thread: accumulates the even moments  M_n = sum_j mu_j^(2n) P(k, mu_j), n = 0..3, of this
// pipeline (s) and of its peak partner (q).
//   SPEC = true : KM / ARINYO / PAIRED / NVD are compile-time (production model set: Kaiser, UV, Rogers HCD,
//                 Arinyo, G(k), Gaussian smoothing / peak broadening, Lorentz velocity dispersion on tracer 2);
//   SPEC = false: every switch is read from T at run time; RARE adds sinc HCD, McDonald NL, exponential
//                 smoothing and the fast-metals division.
template <int MS, int WB, bool SPEC, int KM, bool ARINYO, bool PAIRED, int NVD, bool RARE>
__device__ __forceinline__ void pk_mu_loop(const PkThread& T, const double* s_mubv, int ms, int j_lo, int n_mu,
                                           double inv_nmu, double* s, double* q)
{
    // midpoints j_lo + ms, j_lo + ms + MS, ... < n_mu; the sums are ADDED to s / q
    const bool same = SPEC ? (KM == KM_SAME_HCD || KM == KM_SAME_PLAIN) : T.same;
    const bool hcd1 = SPEC ? (KM == KM_SAME_HCD || KM == KM_FIRST_HCD) : T.hcd1;
    const bool hcd2 = SPEC ? false : T.hcd2;
    const bool arinyo = SPEC ? ARINYO : T.arinyo;
    const bool paired = SPEC ? PAIRED : T.paired;
    const bool has_vd1 = SPEC ? false : T.has_vd1;
    const bool has_vd2 = SPEC ? (NVD == 1) : T.has_vd2;
    const bool rogers = SPEC ? hcd1 : T.rogers;

    const double dmu = (double)MS * inv_nmu;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const double* gk = T.gk != nullptr ? T.gk + (size_t)j_lo * T.gk_row : nullptr;
    double g_next = (gk != nullptr && j_lo + ms < n_mu) ? *gk : 1.0;
    double m_next = (arinyo && j_lo + ms < n_mu) ? s_mubv[j_lo + ms] : 0.0;

    for (int j0 = j_lo + ms; j0 < n_mu; j0 += MS * PK_REANCHOR) {
        if (WB > 1) __syncthreads();       // keep the walkers of a block on the same table rows (L1 reuse)
        // exact anchors of the progressions along this thread's mu sequence
        double mu = ((double)j0 + 0.5) * inv_nmu;
        double F = 0.0, pg = 1.0, pr = 1.0;
        if (rogers) F = vmx_exp(-T.L0 * T.k * mu);
        if (paired) {
            pg = vmx_exp(fma(T.p1, mu * mu, T.p0));
            pr = vmx_exp(T.p1 * fma(2.0 * mu, dmu, dmu * dmu));
        }
        int jend = j0 + MS * PK_REANCHOR;
        if (jend > n_mu) jend = n_mu;
        for (int j = j0; j < jend; j += MS) {
            const double mu2 = mu * mu;
            const double g = g_next;
            const double m = m_next;
            if (j + MS < n_mu) {            // prefetch the next step's table entries
                if (gk != nullptr) { gk += T.gk_stride; g_next = *gk; }
                if (arinyo) m_next = s_mubv[j + MS];
            }
            if (!SPEC && RARE && T.sinc) { const double x = T.k * mu * T.L0; F = sin(x) / x; }
            if (!SPEC && RARE && T.fvoigt) F = fvoigt_interp(T.L0 * (T.k * mu), T.fv_x, T.fv_f, T.fv_n);

            // tracer amplitudes b_eff (1 + beta_eff mu^2) = b + b beta mu^2 + F b_hcd (1 + beta_hcd mu^2)
            const double hmu = fma(T.hbb, mu2, T.hb);
            double A1 = fma(T.c1_1, mu2, T.c0_1);
            if (hcd1) A1 = fma(F, hmu, A1);
            if (!SPEC && RARE && T.div1) A1 /= fma(F, T.hb, T.c0_1);
            double AA;
            if (same) AA = A1 * A1;
            else {
                double A2 = fma(T.c1_2, mu2, T.c0_2);
                if (hcd2) A2 = fma(F, hmu, A2);
                if (!SPEC && RARE && T.div2) A2 /= fma(F, T.hb, T.c0_2);
                AA = A1 * A2;
            }

            double E = fma(T.e1, mu2, T.e0);
            if (arinyo) E = fma(T.e2, m, E);
            if (!SPEC && RARE) {
                if (T.has_exp) E -= T.k * fma(mu, T.ea, sqrt(1.0 - mu2) * T.eb);
                if (T.mcdonald) { const double x = T.k * mu / T.mc_kvel; E -= x * sqrt(x); }
                E = fmin(E, 709.0);
            }

            double val = AA * vmx_exp(E) * g;
            if (!SPEC && RARE && T.mock_c != 0.0) { const double x = T.mock_c * mu; val *= sin(x) / x; }        // (mu > 0 on the midpoints)
            if (has_vd1 || has_vd2) {
                const double kpar = T.k * mu;
                const double kp2 = kpar * kpar;
                if (has_vd1) val *= vmx_rsqrt(fma(kp2, T.vd1, 1.0));
                if (has_vd2) val *= vmx_rsqrt(fma(kp2, T.vd2, 1.0));
            }

            const double mu4 = mu2 * mu2, mu6 = mu4 * mu2;
            s0 += val;
            s1 = fma(mu2, val, s1);
            s2 = fma(mu4, val, s2);
            s3 = fma(mu6, val, s3);
            if (paired) {
                const double vp = val * pg;
                q0 += vp;
                q1 = fma(mu2, vp, q1);
                q2 = fma(mu4, vp, q2);
                q3 = fma(mu6, vp, q3);
                pg *= pr;
                pr *= T.pq;
            }
            F *= T.Fq;
            mu += dmu;
        }
    }
    s[0] += s0; s[1] += s1; s[2] += s2; s[3] += s3;
    q[0] += q0; q[1] += q1; q[2] += q2; q[3] += q3;
}

// mu loop of a shared-W group: six even moments of the amplitude-free factor
//   W(k, mu) = G(k, mu) exp(e0 + e1 mu^2) / sqrt((1 + (k mu s1)^2)(1 + (k mu s2)^2)),
// from which every member pipeline P = P_lin (c0_1 + c1_1 mu^2)(c0_2 + c1_2 mu^2) W forms its own moments.
template <int MS, int WB>
__device__ __forceinline__ void pk_w_loop(const PkThread& T, int ms, int j_lo, int n_mu, double inv_nmu, const v2d* s_mu24, double* wm)
{
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0, m5 = 0.0;
    const double* gk = T.gk != nullptr ? T.gk + (size_t)j_lo * T.gk_row : nullptr;
    const double* gk_first = gk;
    const double k2 = T.k * T.k;
    const double k2vd1 = k2 * T.vd1, k2vd2 = k2 * T.vd2;
#define VMX_PK_W_STEP(G, J)                                                                                           \
    {                                                                                                                 \
        /* (mu^2, mu^4) from the block's LDS table when the launch provides one */                                    \
        double mu2, mu4;                                                                                              \
        if (s_mu24 != nullptr) { const v2d mm = s_mu24[J]; mu2 = mm.x; mu4 = mm.y; }                                  \
        else { const double mu = ((double)(J) + 0.5) * inv_nmu; mu2 = mu * mu; mu4 = mu2 * mu2; }                     \
        double val = G;                                                                                               \
        if (!T.noexp) val *= vmx_exp(fma(T.e1, mu2, T.e0));                                                           \
        if (T.has_vd1) val *= vmx_rsqrt(fma(k2vd1, mu2, 1.0));                                                        \
        if (T.has_vd2) val *= vmx_rsqrt(fma(k2vd2, mu2, 1.0));                                                        \
        m0 += val;                                                                                                    \
        m1 = fma(mu2, val, m1);                                                                                       \
        m2 = fma(mu4, val, m2);                                                                                       \
        const double v6 = val * (mu4 * mu2);                                                                          \
        m3 += v6;                                                                                                     \
        m4 = fma(mu2, v6, m4);                                                                                        \
        m5 = fma(mu4, v6, m5);                                                                                        \
    }
    const int n_steps = n_mu > j_lo + ms ? (n_mu - j_lo - ms + MS - 1) / MS : 0;
    int j = j_lo + ms, done = 0;
    if (gk != nullptr && n_steps >= 8) {
        // the table runs four steps ahead in four registers used in turn (see pk_tab_loop); the groups of this loop
        // request rows that exist, the last four requested rows are consumed after it
        double g0 = gk[0], g1 = gk[T.gk_stride], g2 = gk[2 * T.gk_stride], g3 = gk[3 * T.gk_stride];
        gk += 4 * T.gk_stride;
        const int groups = (n_steps - 4) / 4;
        for (int it = 0; it < groups; ++it, j += 4 * MS) {
            { const double g = g0; g0 = *gk; gk += T.gk_stride; VMX_PK_W_STEP(g, j) }
            { const double g = g1; g1 = *gk; gk += T.gk_stride; VMX_PK_W_STEP(g, j + MS) }
            { const double g = g2; g2 = *gk; gk += T.gk_stride; VMX_PK_W_STEP(g, j + 2 * MS) }
            { const double g = g3; g3 = *gk; gk += T.gk_stride; VMX_PK_W_STEP(g, j + 3 * MS) }
        }
        VMX_PK_W_STEP(g0, j) VMX_PK_W_STEP(g1, j + MS) VMX_PK_W_STEP(g2, j + 2 * MS) VMX_PK_W_STEP(g3, j + 3 * MS)
        j += 4 * MS;
        done = 4 * groups + 4;
    }
    for (int st = done; st < n_steps; ++st, j += MS) {
        const double g = (gk_first != nullptr) ? gk_first[(size_t)st * T.gk_stride] : 1.0;
        VMX_PK_W_STEP(g, j)
    }
#undef VMX_PK_W_STEP
    wm[0] += m0; wm[1] += m1; wm[2] += m2; wm[3] += m3; wm[4] += m4; wm[5] += m5;
}

#define VMX_XTAB_KEY 12
// what the tables of group g depend on: {level, six Arinyo parameters, ga, gb of the pipeline, ga, gb of its peak partner}
// of the batch's first walker t0 (the expressions of k_prologue's scalars)
__device__ inline void xtab_key_now(const EngineDev& D, int g, double* key, const double* t0)
{
    const int pipe = D.xtab_pipe[g], partner = D.xtab_partner[g];
    const vmx_pipe_desc& d = D.pipes[pipe].d;
    key[0] = (double)D.xtab_level;
    for (int i = 0; i < 6; ++i) key[1 + i] = d.arinyo_slot[i] >= 0 ? t0[d.arinyo_slot[i]] : 0.0;
    for (int i = 7; i < VMX_XTAB_KEY; ++i) key[i] = 0.0;
    if (D.xtab_level >= 2) {
        gauss_exponents(t0, d, th(t0, d.growth_rate_slot, d.growth_rate_default), key[7], key[8]);
        if (partner >= 0) {
            const vmx_pipe_desc& dp = D.pipes[partner].d;
            gauss_exponents(t0, dp, th(t0, dp.growth_rate_slot, dp.growth_rate_default), key[9], key[10]);
        }
    }
}

// xtab_key is [2][tables][VMX_XTAB_KEY]: what the tables hold, and what the table blocks of the running evaluation found for
// walker 0 (written by one of them whether they rebuild or not).  The chi2 kernels move the second to the first - one thread,
// after every reader of the old key has finished.
__device__ inline void xtab_key_store(const EngineDev& D)
{
    if (D.xtab_level <= 0) return;
    for (int i = 0; i < D.n_xtab * VMX_XTAB_KEY; ++i) D.xtab_key[i] = D.xtab_key[D.n_xtab * VMX_XTAB_KEY + i];
}

// The tables of a batch that shares its non-linear (level 1) and Gaussian (level 2) parameters, from the first walker
// (power_spectrum.py:435-479 D_NL, :526-556 smoothing, :382-417 peak broadening).  blocks = (k blocks, row blocks, groups),
// as part of k_prologue's grid or as k_xtab; blocks that find their tables current read the key and leave.
__device__ void xtab_body(const EngineDev& D, int xtab, int bx, int by, int rows_per_block, const double* t0)
{
    const int pipe = D.xtab_pipe[xtab];
    const int i = bx * (int)blockDim.x + threadIdx.x;
    const size_t plane = (size_t)D.n_rows * D.nkp;
    double key[VMX_XTAB_KEY];
    xtab_key_now(D, xtab, key, t0);
    bool stale = false;
    for (int q = 0; q < VMX_XTAB_KEY; ++q) stale |= !(D.xtab_key[xtab * VMX_XTAB_KEY + q] == key[q]);      // keys start as NaN
    if (bx == 0 && by == 0 && threadIdx.x == 0)
        for (int q = 0; q < VMX_XTAB_KEY; ++q) D.xtab_key[(D.n_xtab + xtab) * VMX_XTAB_KEY + q] = key[q];
    if (!stale || i >= D.nkp) return;      // built from the same parameters by an earlier batch
    for (int j = by * rows_per_block; j < min((by + 1) * rows_per_block, D.n_rows); ++j) {
        double* cell = D.xtab + (size_t)xtab * 2 * plane + (size_t)j * D.nkp + i;
        double val = 0.0, val_q = 0.0;
        if (i < D.nk) {
            const vmx_pipe_desc& d = D.pipes[pipe].d;
            const double k = D.k[i], d2 = D.delta2[i];
            const double g = key[1] * d2 + key[2] * d2 * d2;
            const double gv = g * pow(k / key[3], key[4]);
            const double kp = k / key[6];
            const double gp = g - kp * kp;
            const double m = vmx_exp(key[5] * D.lnmu[j]);
            val = exp(fmin(d.arinyo_power * fma(-gv, m, gp), 709.0));
            if (d.gk_table >= 0) val *= D.gk[((size_t)d.gk_table * D.n_rows + j) * D.nkp + i];
            if (D.xtab_level >= 2) {
                const double mu = D.mu[j], mu2 = mu * mu, k2 = k * k;
                val *= vmx_exp(-k2 * fma(key[7] - key[8], mu2, key[8]));
                val_q = val * vmx_exp(-k2 * fma((key[9] - key[7]) - (key[10] - key[8]), mu2, key[10] - key[8]));
            }
            if (j == 0) {
                // per wavenumber: the Arinyo part of e0, e2 (the underflow bound of k_pk_multipoles) and VegaArinyoError - NaN or
                // Inf in exp(growth (1 - pec) - pressure) anywhere on the grid (power_spectrum.py:466-469); the exponent is
                // monotonic in mu^bv, so its extremes sit at the two ends of the mu grid
                double* kk = D.xtab_k + (size_t)xtab * 4 * D.nkp + i;
                kk[0] = d.arinyo_power * gp;
                kk[D.nkp] = -d.arinyo_power * gv;
                const double lo = fma(-gv, vmx_exp(key[5] * D.lnmu[0]), gp), hi = fma(-gv, vmx_exp(key[5] * D.lnmu[D.n_mu - 1]), gp);
                kk[2 * D.nkp] = (!(lo < 709.0) || !(hi < 709.0)) ? 1.0 : 0.0;
                // level 2: every exponent is shared by the batch - the underflow bound of k_pk_multipoles, per wavenumber
                const double k2 = k * k, ga = key[7], gb = key[8], dga = key[9] - key[7], dgb = key[10] - key[8];
                kk[3 * D.nkp] = fma(-k2, gb, kk[0]) + fmax(-k2 * (ga - gb), 0.0) + fmax(kk[D.nkp], 0.0) +
                                fmax(-k2 * dgb + fmax(-k2 * (dga - dgb), 0.0), 0.0);
            }
        }
        *cell = val;
        if (D.xtab_level >= 2) cell[plane] = val_q;
    }
}

__global__ __launch_bounds__(256) void k_xtab(EngineDev D) { xtab_body(D, blockIdx.z, blockIdx.x, blockIdx.y, XTAB_ROWS, D.theta); }

// mu loop against the tabulated D_NL * G: no exponential is left in the loop - the HCD factor, the Gaussian
// smoothing exp(e0 + e1 mu^2) and the peak broadening all advance as geometric progressions (re-anchored exactly
// every PK_REANCHOR steps).
template <int MS, int WB, int KM, bool PAIRED, int NVD>
__device__ __forceinline__ void pk_tab_loop(const PkThread& T, double e0g, int ms, int j_lo, int n_mu, double inv_nmu,
                                            const v2d* s_mu24, double* s, double* q)
{
    // mu^2 and mu^4 of every step come from an LDS table (a wave shares its mu: one broadcast read instead of three
    // multiplications and an addition in the VALU-bound loop)
    const double k2vd2 = T.k * T.k * T.vd2;
    const bool same = (KM == KM_SAME_HCD);
    const double dmu = (double)MS * inv_nmu;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    // The table stream runs four steps ahead of its use, in four registers used in turn (the loop is unrolled by four: a
    // rotating register window costs five moves per step and makes the compiler wait for ALL outstanding loads at every
    // use - vmcnt(0) - which serialises each step behind an L2 round trip; with fixed registers it waits vmcnt(3)).
    const double* tab = T.gk + (size_t)j_lo * T.gk_row;
    double g0 = *tab, g1 = tab[T.gk_stride], g2 = tab[2 * T.gk_stride], g3 = tab[3 * T.gk_stride];
    tab += 4 * T.gk_stride;
    const double gq = vmx_exp(2.0 * T.e1 * dmu * dmu);
    static_assert(PK_REANCHOR % 4 == 0, "the four-register window restarts with every re-anchoring chunk");
    for (int j0 = j_lo + ms; j0 < n_mu; j0 += MS * PK_REANCHOR) {
        if (WB > 1) __syncthreads();       // keep the walkers of a block on the same table rows (L1 reuse)
        double mu = ((double)j0 + 0.5) * inv_nmu;
        double F = vmx_exp(-T.L0 * T.k * mu);
        double gs = vmx_exp(fma(T.e1, mu * mu, e0g));
        double gr = vmx_exp(T.e1 * fma(2.0 * mu, dmu, dmu * dmu));
        double pg = 1.0, pr = 1.0;
        if (PAIRED) {
            pg = vmx_exp(fma(T.p1, mu * mu, T.p0));
            pr = vmx_exp(T.p1 * fma(2.0 * mu, dmu, dmu * dmu));
        }
        int jend = j0 + MS * PK_REANCHOR;
        if (jend > n_mu) jend = n_mu;
#define VMX_PK_TAB_STEP(GREG, J)                                                                                      \
        {                                                                                                             \
            const v2d mm = s_mu24[J];                                                                                 \
            const double mu2 = mm.x, mu4 = mm.y;                                                                      \
            const double g = GREG;                                                                                    \
            GREG = *tab; tab += T.gk_stride;    /* (rows past the end are padding of the table buffer) */              \
            const double hmu = fma(T.hbb, mu2, T.hb);                                                                 \
            const double A1 = fma(F, hmu, fma(T.c1_1, mu2, T.c0_1));                                                  \
            const double AA = same ? A1 * A1 : A1 * fma(T.c1_2, mu2, T.c0_2);                                         \
            double val = AA * (g * gs);                                                                               \
            if (NVD == 1) val *= vmx_rsqrt(fma(k2vd2, mu2, 1.0));                                                     \
            const double mu6 = mu4 * mu2;                                                                             \
            s0 += val;                                                                                                \
            s1 = fma(mu2, val, s1);                                                                                   \
            s2 = fma(mu4, val, s2);                                                                                   \
            s3 = fma(mu6, val, s3);                                                                                   \
            if (PAIRED) {                                                                                             \
                const double vp = val * pg;                                                                           \
                q0 += vp;                                                                                             \
                q1 = fma(mu2, vp, q1);                                                                                \
                q2 = fma(mu4, vp, q2);                                                                                \
                q3 = fma(mu6, vp, q3);                                                                                \
                pg *= pr;                                                                                             \
                pr *= T.pq;                                                                                           \
            }                                                                                                         \
            gs *= gr;                                                                                                 \
            gr *= gq;                                                                                                 \
            F *= T.Fq;                                                                                                \
        }
        // whole groups of four steps in a loop without exits (the compiler then counts the outstanding loads exactly),
        // the remainder - last chunk only - after it
        const int chunk_steps = (jend - j0 + MS - 1) / MS;
        int j = j0;
        for (int it = 0; it < chunk_steps / 4; ++it, j += 4 * MS) {
            VMX_PK_TAB_STEP(g0, j)
            VMX_PK_TAB_STEP(g1, j + MS)
            VMX_PK_TAB_STEP(g2, j + 2 * MS)
            VMX_PK_TAB_STEP(g3, j + 3 * MS)
        }
        if (chunk_steps % 4 > 0) VMX_PK_TAB_STEP(g0, j)
        if (chunk_steps % 4 > 1) VMX_PK_TAB_STEP(g1, j + MS)
        if (chunk_steps % 4 > 2) VMX_PK_TAB_STEP(g2, j + 2 * MS)
#undef VMX_PK_TAB_STEP
    }
    s[0] += s0; s[1] += s1; s[2] += s2; s[3] += s3;
    q[0] += q0; q[1] += q1; q[2] += q2; q[3] += q3;
}

// The extra quadrature nodes (rows n_mu .. n_rows of the tables, weights D.node_w) of a core / plain pipeline: direct
// evaluation - the nodes are not equally spaced, so nothing advances as a progression here.  tab: the D_NL * G table
// replaces exp(Arinyo) * G.  Adds W_j mu_j^(2n) P(k, mu_j) to s (and the peak partner's to q).
template <int MS, int KM, bool ARINYO, bool PAIRED, int NVD, int TAB>
__device__ __forceinline__ void pk_extra_nodes(const EngineDev& D, const PkThread& T, double e0g, const double* s_mubv,
                                               int ms, double* s, double* q)
{
    constexpr bool same = KM == KM_SAME_HCD || KM == KM_SAME_PLAIN;
    constexpr bool hcd1 = KM == KM_SAME_HCD || KM == KM_FIRST_HCD;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const double k2vd2 = T.k * T.k * T.vd2;
    const double fk = -T.L0 * T.k;
    for (int jj = ms; jj < D.n_extra; jj += MS) {
        const int row = D.n_mu + jj;
        const double mu = D.mu[row], w = D.node_w[jj];
        const double mu2 = mu * mu;
        const double g = T.gk != nullptr ? T.gk[((size_t)row - ms) * T.gk_row] : 1.0;
        double A1 = fma(T.c1_1, mu2, T.c0_1);
        if (hcd1) A1 = fma(vmx_exp(fk * mu), fma(T.hbb, mu2, T.hb), A1);
        double AA = same ? A1 * A1 : A1 * fma(T.c1_2, mu2, T.c0_2);
        if (NVD == 1) AA *= vmx_rsqrt(fma(k2vd2, mu2, 1.0));
        AA *= w;
        double val;
        if (TAB == 1) val = AA * (g * vmx_exp(fma(T.e1, mu2, e0g)));
        else {
            double E = fma(T.e1, mu2, T.e0);
            if (ARINYO) E = fma(T.e2, s_mubv[row], E);
            val = AA * vmx_exp(E) * g;
        }
        const double mu4 = mu2 * mu2, mu6 = mu4 * mu2;
        s0 += val; s1 = fma(mu2, val, s1); s2 = fma(mu4, val, s2); s3 = fma(mu6, val, s3);
        if (PAIRED) {
            const double vp = val * vmx_exp(fma(T.p1, mu2, T.p0));
            q0 += vp; q1 = fma(mu2, vp, q1); q2 = fma(mu4, vp, q2); q3 = fma(mu6, vp, q3);
        }
    }
    s[0] += s0; s[1] += s1; s[2] += s2; s[3] += s3;
    if (PAIRED) { q[0] += q0; q[1] += q1; q[2] += q2; q[3] += q3; }
}

// ... and of a shared-W group: six moments of W at the extra nodes
template <int MS>
__device__ __forceinline__ void pk_w_extra_nodes(const EngineDev& D, const PkThread& T, int ms, double* wm)
{
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0, m5 = 0.0;
    const double k2vd1 = T.k * T.k * T.vd1, k2vd2 = T.k * T.k * T.vd2;
    for (int jj = ms; jj < D.n_extra; jj += MS) {
        const int row = D.n_mu + jj;
        const double mu = D.mu[row];
        const double mu2 = mu * mu, mu4 = mu2 * mu2;
        double val = D.node_w[jj] * (T.gk != nullptr ? T.gk[((size_t)row - ms) * T.gk_row] : 1.0);
        if (!T.noexp) val *= vmx_exp(fma(T.e1, mu2, T.e0));
        if (T.has_vd1) val *= vmx_rsqrt(fma(k2vd1, mu2, 1.0));
        if (T.has_vd2) val *= vmx_rsqrt(fma(k2vd2, mu2, 1.0));
        m0 += val; m1 = fma(mu2, val, m1); m2 = fma(mu4, val, m2);
        const double v6 = val * (mu4 * mu2);
        m3 += v6; m4 = fma(mu2, v6, m4); m5 = fma(mu4, v6, m5);
    }
    wm[0] += m0; wm[1] += m1; wm[2] += m2; wm[3] += m3; wm[4] += m4; wm[5] += m5;
}

// Multipoles of the member pipelines of a shared-W group from the six even moments wm of W (power_spectrum.py:163-222 with
// P = P_lin (c0_1 + c1_1 mu^2)(c0_2 + c1_2 mu^2) W): walker b, wavenumber index i (value k).
__device__ __forceinline__ void w_members_store(const EngineDev& D, const PkGroup& G, const int32_t* members, int b, int B,
                                                int i, double k, const double* wm)
{
    const double inv_nmu = 1.0 / (double)D.n_mu;
    const size_t ncols = (size_t)B * D.n_active;
    for (int mi = 0; mi < G.n_members; ++mi) {
        const int pm = members[G.member_off + mi];
        const vmx_pipe_desc& dm = D.pipes[pm].d;
        const double* scm = D.scal + ((size_t)b * D.n_pipe + pm) * VMX_NS;
        double c01 = scm[S_BIAS1], c02 = scm[S_BIAS2];
        const double c11 = scm[S_BB1], c12 = scm[S_BB2];
        if (dm.uvb || dm.heii) {
            double add = 0.0;
            if (dm.uvb) { const double x = k * scm[S_UV_LAM]; const double W = atan(x) / x; add += scm[S_UV_BG] * W / (1.0 + scm[S_UV_BP] * W); }
            if (dm.heii) { const double x = k * scm[S_HE_LAM]; const double W = atan(x) / x; add += scm[S_HE_BG] * W / (1.0 + scm[S_UV_BP] * W); }
            if (dm.tracer[0].is_lya) c01 += add;
            if (dm.tracer[1].is_lya) c02 += add;
        }
        if (dm.same_tracer) c02 = c01;
        const double c12e = dm.same_tracer ? c11 : c12;
        const double a0 = c01 * c02, a1 = fma(c01, c12e, c11 * c02), a2 = c11 * c12e;
        double mm[4];
        for (int m = 0; m < 4; ++m) mm[m] = fma(a2, wm[m + 2], fma(a1, wm[m + 1], a0 * wm[m]));
        double damp = 1.0;
        if (dm.damping_scale > 0.0) damp = exp(-dm.damping_scale * dm.damping_scale * pow(k, (double)dm.damping_power) / 2.0);
        const double pk = damp * ((D.pk_direct && dm.pk_lin_kind == VMX_PKLIN_SMOOTH) ? D.pk_direct[(size_t)b * D.nkp + i]
                                                                                      : D.pklin[(size_t)dm.pk_lin_kind * D.nkp + i]) * inv_nmu;
        const size_t col = (size_t)D.pipes[pm].col * B + b;
        D.pl[((size_t)0 * ncols + col) * D.nkp + i] = pk * mm[0];
        D.pl[((size_t)1 * ncols + col) * D.nkp + i] = pk * (7.5 * mm[1] - 2.5 * mm[0]);
        D.pl[((size_t)2 * ncols + col) * D.nkp + i] = pk * (39.375 * mm[2] - 33.75 * mm[1] + 3.375 * mm[0]);
        D.pl[((size_t)3 * ncols + col) * D.nkp + i] = pk * (187.6875 * mm[3] - 255.9375 * mm[2] + 85.3125 * mm[1] - 4.0625 * mm[0]);
    }
}

// Block = KT wavenumbers x MS mu-slices x WB walkers (KT * MS * WB = 256).  WB > 1 lets the waves of a block share the
// rows of the static table through the CU's L1 (94 % L1 hit rate measured with WB = 4); it did not shorten the kernel
// on MI355X (L2 was not the limiter), so the engine launches WB = 1 shapes only.
#ifndef VMX_PK_WAVES
#define VMX_PK_WAVES 4
#endif
template <int KT, int MS, int WB, bool GENERIC>
__global__ __launch_bounds__(256, GENERIC ? 1 : VMX_PK_WAVES) void k_pk_multipoles(EngineDev D, const PkGroup* groups, const int32_t* members,
                                                       int tab_mode, int B, int mu_tab_off)
{
    extern __shared__ double smem[];
    const int wb = (WB > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x / (KT * MS)) : 0;
    // LDS: [8][256] reduction scratch first; the per-walker mu^bv tables follow only when the launch needs them
    // (the tabulated-D_NL mode does not: a smaller footprint lets more blocks share a CU)
    double* s_red = smem;
    double* s_mubv = smem + 2048 + (size_t)wb * D.n_rows;      // [WB][n_rows]  mu^bv (Arinyo), one table per walker
    // (tab_mode + 32: the shared-W groups run in k_pk_w and are skipped here; block-uniform)
    if ((tab_mode & 32) && groups[blockIdx.y].variant == PKV_SHARED_W) return;
    tab_mode &= 3;
    const bool use_tab = tab_mode && groups[blockIdx.y].xtab >= 0 &&
                         (groups[blockIdx.y].variant == PKV_AUTO_CORE || groups[blockIdx.y].variant == PKV_CROSS_CORE);
    if (tab_mode >= 2 && use_tab) return;             // (k_pk_tab2 serves this group; block-uniform)
    v2d* s_mu24 = (v2d*)(smem + (mu_tab_off >= 0 ? mu_tab_off : 0));        // [n_mu] (mu^2, mu^4) when the launch has room
    if (mu_tab_off >= 0 && (use_tab || groups[blockIdx.y].variant == PKV_SHARED_W))
        for (int j = threadIdx.x; j < D.n_mu; j += 256) { const double m = D.mu[j], m2 = m * m; s_mu24[j] = (v2d){m2, m2 * m2}; }

    int b = blockIdx.x * WB + wb;
    const bool walker_ok = b < B;
    if (!walker_ok) b = B - 1;                        // surplus waves shadow the last walker and store nothing
    const int p = groups[blockIdx.y].pipe, pp = groups[blockIdx.y].peak_partner;
    const int variant = groups[blockIdx.y].variant;
    const vmx_pipe_desc& d = D.pipes[p].d;
    const double* sc = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;
    const int kk = threadIdx.x % KT;
    const int lt = threadIdx.x % (KT * MS);           // thread index within this walker's sub-block
    // with KT = 64 a wave shares its mu slice: keep the slice index in a scalar register
    const int ms = (KT == 64) ? __builtin_amdgcn_readfirstlane((threadIdx.x / KT) % MS) : (int)((threadIdx.x / KT) % MS);
    const int i = blockIdx.z * KT + kk;
    const bool valid = i < D.nk;
    const int ic = valid ? i : D.nk - 1;
    const int n_mu = D.n_mu;
    const double inv_nmu = 1.0 / (double)n_mu;

    PkThread T;
    T.paired = pp >= 0;
    T.arinyo = d.nl_model == VMX_NL_ARINYO;
    if (T.arinyo && !use_tab) {
        const double bv = sc[S_ABV];
        for (int j = lt; j < D.n_rows; j += KT * MS) s_mubv[j] = vmx_exp(bv * D.lnmu[j]);     // mu^bv (midpoints, then nodes)
    }

    __syncthreads();
    const double k = D.k[ic], k2 = k * k;
    T.k = k;
    const bool hcd = d.hcd_model != VMX_HCD_NONE;
    const bool lya1 = d.tracer[0].is_lya, lya2 = d.tracer[1].is_lya;
    T.same = d.same_tracer;
    T.rogers = d.hcd_model == VMX_HCD_ROGERS;
    T.sinc = d.hcd_model == VMX_HCD_SINC;
    T.fvoigt = d.hcd_model == VMX_HCD_FVOIGT;
    // a mock's line-of-sight bin that follows a sampled parameter: sinc(k_par L (1 + p) / 2) (power_spectrum.py:143-160, :499)
    T.mock_c = d.mock_los_slot >= 0 ? 0.5 * k * (d.mock_los_size * (1.0 + D.theta[(size_t)b * D.n_params + d.mock_los_slot])) : 0.0;
    T.fv_x = D.fv_x; T.fv_f = D.fv_f; T.fv_n = D.fv_n;

    // k-dependent effective bias from UV / HeII (power_spectrum.py:224-261): only the bias changes,
    // bias * beta is invariant under that step.
    T.c0_1 = sc[S_BIAS1]; T.c0_2 = sc[S_BIAS2]; T.c1_1 = sc[S_BB1]; T.c1_2 = sc[S_BB2];
    if (d.uvb || d.heii) {
        double add = 0.0;
        if (d.uvb) { const double x = k * sc[S_UV_LAM]; const double W = atan(x) / x; add += sc[S_UV_BG] * W / (1.0 + sc[S_UV_BP] * W); }
        if (d.heii) { const double x = k * sc[S_HE_LAM]; const double W = atan(x) / x; add += sc[S_HE_BG] * W / (1.0 + sc[S_UV_BP] * W); }
        if (lya1) T.c0_1 += add;
        if (lya2) T.c0_2 += add;
    }
    T.hb = sc[S_HCD_B]; T.hbb = sc[S_HCD_BB]; T.L0 = sc[S_HCD_L0];
    T.hcd1 = lya1 && hcd; T.hcd2 = lya2 && hcd;
    T.div1 = d.fast_metals && lya1 && (d.uvb || d.heii || hcd);
    T.div2 = d.fast_metals && lya2 && (d.uvb || d.heii || hcd);

    // exponent  E = e0 + e1 mu^2 + e2 mu^bv  (+ rarely used extra terms)
    const double ga = sc[S_GA], gb = sc[S_GB];
    T.ea = sc[S_EA]; T.eb = sc[S_EB];
    T.has_exp = (T.ea != 0.0) || (T.eb != 0.0);
    T.vd1 = sc[S_VD1]; T.vd2 = sc[S_VD2];
    T.has_vd1 = T.vd1 != 0.0; T.has_vd2 = T.vd2 != 0.0;
    T.e0 = -k2 * gb; T.e1 = -k2 * (ga - gb); T.e2 = 0.0;
    T.noexp = (ga == 0.0) && (gb == 0.0);
    bool bad = false;
    if (T.arinyo && use_tab) {
        // the Arinyo terms of this wavenumber were formed with the table (k_xtab): the batch shares them
        const double* kk = D.xtab_k + (size_t)groups[blockIdx.y].xtab * 4 * D.nkp + ic;
        T.e0 += kk[0];
        T.e2 = kk[D.nkp];
        bad = kk[2 * D.nkp] != 0.0;
    } else if (T.arinyo) {
        const double apow = d.arinyo_power;
        const double d2 = D.delta2[ic];
        const double ar_g = sc[S_AQ1] * d2 + sc[S_AQ2] * d2 * d2;
        const double ar_gv = ar_g * pow(k / sc[S_AKV], sc[S_AAV]);
        const double kp = k / sc[S_AKP];
        const double ar_gp = ar_g - kp * kp;
        T.e0 = fma(apow, ar_gp, T.e0);
        T.e2 = -apow * ar_gv;
        // VegaArinyoError: NaN or Inf in exp(growth (1 - pec) - pressure) anywhere on the grid
        // (power_spectrum.py:466-469).  The exponent is monotonic in mu^bv, so its extremes sit at the two
        // ends of the mu grid.
        const double bv = sc[S_ABV];
        const double lo = fma(-ar_gv, vmx_exp(bv * D.lnmu[0]), ar_gp), hi = fma(-ar_gv, vmx_exp(bv * D.lnmu[n_mu - 1]), ar_gp);
        if (!(lo < 709.0) || !(hi < 709.0)) bad = true;
    }
    T.mc_kvel = 1.0;
    T.mcdonald = d.nl_model == VMX_NL_MCDONALD;
    if (T.mcdonald) {
        T.mc_kvel = 1.22 * pow(1.0 + k / 0.923, 0.451);
        T.e0 += pow(k / 6.4, 0.569) - pow(k / 15.3, 2.01);
    }
    T.gk_stride = (size_t)MS * D.nkp;
    T.gk_row = (size_t)D.nkp;
    T.gk = d.gk_table >= 0 ? D.gk + (size_t)d.gk_table * D.n_rows * D.nkp + (size_t)ms * D.nkp + ic : nullptr;
    if constexpr (KT == 8) {
        // single-walker shape (one wave per SIMD, ~31 mu steps per thread): nothing hides the latency of the table
        // reads inside the loop, so every thread requests its own entries back to back here and the loop reads LDS
        if (T.gk != nullptr) {
            double* s_g = smem + 2048 + D.n_rows + (size_t)ms * KT + kk;      // [n_rows][KT]
            const double* src = T.gk;
#pragma unroll 8
            for (int j = ms; j < D.n_rows; j += MS) { s_g[(size_t)(j - ms) * KT] = *src; src += T.gk_stride; }
            T.gk = s_g;
            T.gk_stride = (size_t)MS * KT;
            T.gk_row = (size_t)KT;
        }
    }

    // peak partner: extra factor exp(p0 + p1 mu^2) from the additional Gaussian broadening
    const double dmu = (double)MS * inv_nmu;
    T.p0 = 0.0; T.p1 = 0.0; T.pq = 1.0;
    if (T.paired) {
        const double* scp = D.scal + ((size_t)b * D.n_pipe + pp) * VMX_NS;
        const double dga = scp[S_GA] - ga, dgb = scp[S_GB] - gb;
        T.p0 = -k2 * dgb;
        T.p1 = -k2 * (dga - dgb);
        T.pq = vmx_exp(2.0 * T.p1 * dmu * dmu);
    }
    // F = exp(-L0 k mu_j) along this thread's mu sequence is a geometric progression
    T.Fq = T.rogers ? vmx_exp(-T.L0 * k * dmu) : 0.0;

    // Wavenumbers whose every (k, mu) value underflows the double range of the multipole sums contribute
    // exactly nothing: bound the exponent over mu in (0, 1] (mu^2 and mu^bv lie in (0, 1]) and skip the mu loop
    // when a whole wave is past that bound.  exp(-200) ~ 1e-87 leaves > 60 decades of margin for the amplitudes.
    double e_max = T.e0 + fmax(T.e1, 0.0) + fmax(T.e2, 0.0);
    if (T.paired) e_max += fmax(T.p0 + fmax(T.p1, 0.0), 0.0);
    const bool live_block = __syncthreads_or(!(e_max < VMX_PK_DEAD)) != 0;     // block-uniform: the mu loops contain barriers
    // the node rule (first mu_lo and last mu_hi midpoints plus the extra nodes) serves a tile whose wavenumbers are all
    // within its range or negligible (VMX_PK_NEGLIGIBLE); block-uniform
    const bool node_mode = GENERIC ? false : (D.n_extra > 0 && variant != PKV_GENERIC && sc[S_NO_RULE] == 0.0 &&
                                              __syncthreads_and(k <= D.k_node_max || e_max < VMX_PK_NEGLIGIBLE) != 0);
    if (live_block && node_mode && threadIdx.x == 0) atomicMax(D.k_live + 1, min((int)(blockIdx.z + 1) * KT, D.nk));
    // the FFTLog product skips the wavenumbers past the last live block (their P_ell is exactly zero)
    if (live_block && threadIdx.x == 0) atomicMax(D.k_live, min((int)(blockIdx.z + 1) * KT, D.nk));
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (variant == PKV_SHARED_W) {
        // one mu loop for all member pipelines (e.g. QSO x each metal line): they share W and differ only in
        // the Kaiser polynomials
        double wm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (live_block) {
            const v2d* mt = mu_tab_off >= 0 ? s_mu24 : nullptr;
            if (node_mode) {
                pk_w_loop<MS, WB>(T, ms, 0, D.mu_lo, inv_nmu, mt, wm);
                pk_w_loop<MS, WB>(T, ms, n_mu - D.mu_hi, n_mu, inv_nmu, mt, wm);
                pk_w_extra_nodes<MS>(D, T, ms, wm);
            } else pk_w_loop<MS, WB>(T, ms, 0, n_mu, inv_nmu, mt, wm);
        }
        for (int n = 0; n < 6; ++n) s_red[n * 256 + threadIdx.x] = wm[n];
        __syncthreads();
        if (lt < KT && valid && walker_ok) {
            for (int n = 0; n < 6; ++n) {
                double sum = 0.0;
                for (int qq = 0; qq < MS; ++qq) sum += s_red[n * 256 + (wb * MS + qq) * KT + kk];
                wm[n] = sum;
            }
            w_members_store(D, groups[blockIdx.y], members, b, B, i, k, wm);
        }
        return;
    }
    const int xt = groups[blockIdx.y].xtab;
    // uniform midpoint ranges of this block: everything, or the two ends of the node rule
    const int n_ranges = node_mode ? 2 : 1;
    if (use_tab) {
        if (live_block) {
            const size_t plane = (size_t)D.n_rows * D.nkp;
            T.gk = D.xtab + (size_t)xt * 2 * plane + (size_t)ms * D.nkp + ic;
            T.gk_row = (size_t)D.nkp; T.gk_stride = (size_t)MS * D.nkp;
            for (int rg = 0; rg < n_ranges; ++rg) {
                const int j_lo = rg == 0 ? 0 : n_mu - D.mu_hi, j_hi = (node_mode && rg == 0) ? D.mu_lo : n_mu;
                if (variant == PKV_AUTO_CORE) pk_tab_loop<MS, WB, KM_SAME_HCD, true, 0>(T, -k2 * gb, ms, j_lo, j_hi, inv_nmu, s_mu24, s, q);
                else pk_tab_loop<MS, WB, KM_FIRST_HCD, true, 1>(T, -k2 * gb, ms, j_lo, j_hi, inv_nmu, s_mu24, s, q);
            }
            if (node_mode) {
                if (variant == PKV_AUTO_CORE) pk_extra_nodes<MS, KM_SAME_HCD, true, true, 0, 1>(D, T, -k2 * gb, s_mubv, ms, s, q);
                else pk_extra_nodes<MS, KM_FIRST_HCD, true, true, 1, 1>(D, T, -k2 * gb, s_mubv, ms, s, q);
            }
        }
    } else if (live_block) {
    for (int rg = 0; rg < n_ranges; ++rg) {
    const int j_lo = rg == 0 ? 0 : n_mu - D.mu_hi, j_hi = (node_mode && rg == 0) ? D.mu_lo : n_mu;
    switch (variant) {
        case PKV_AUTO_CORE: pk_mu_loop<MS, WB, true, KM_SAME_HCD, true, true, 0, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q); break;
        case PKV_CROSS_CORE: pk_mu_loop<MS, WB, true, KM_FIRST_HCD, true, true, 1, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q); break;
        case PKV_PLAIN_SAME: pk_mu_loop<MS, WB, true, KM_SAME_PLAIN, false, false, 0, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q); break;
        case PKV_PLAIN_PAIR: pk_mu_loop<MS, WB, true, KM_BOTH_PLAIN, false, false, 0, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q); break;
        case PKV_PLAIN_PAIR_VD: pk_mu_loop<MS, WB, true, KM_BOTH_PLAIN, false, false, 1, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q); break;
        default:
            // the run-time-switched loops (rare model options) live in the GENERIC instantiation only: their register
            // footprint would otherwise cap the occupancy of the production loops
            if constexpr (GENERIC) {
                if (T.sinc || T.fvoigt || T.has_exp || T.mcdonald || T.div1 || T.div2 || T.mock_c != 0.0)
                    pk_mu_loop<MS, WB, false, 0, false, false, 0, true>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q);
                else
                    pk_mu_loop<MS, WB, false, 0, false, false, 0, false>(T, s_mubv, ms, j_lo, j_hi, inv_nmu, s, q);
            }
    }
    }
    if (node_mode)
        switch (variant) {
            case PKV_AUTO_CORE: pk_extra_nodes<MS, KM_SAME_HCD, true, true, 0, 0>(D, T, 0.0, s_mubv, ms, s, q); break;
            case PKV_CROSS_CORE: pk_extra_nodes<MS, KM_FIRST_HCD, true, true, 1, 0>(D, T, 0.0, s_mubv, ms, s, q); break;
            case PKV_PLAIN_SAME: pk_extra_nodes<MS, KM_SAME_PLAIN, false, false, 0, 0>(D, T, 0.0, s_mubv, ms, s, q); break;
            case PKV_PLAIN_PAIR: pk_extra_nodes<MS, KM_BOTH_PLAIN, false, false, 0, 0>(D, T, 0.0, s_mubv, ms, s, q); break;
            case PKV_PLAIN_PAIR_VD: pk_extra_nodes<MS, KM_BOTH_PLAIN, false, false, 1, 0>(D, T, 0.0, s_mubv, ms, s, q); break;
            default: break;
        }
    }

    // moments -> Legendre multipoles  P_ell = (2 ell + 1) / n_mu * sum_n c_{ell n} M_n  (pktoxi.py:37,55,138)
    for (int half = 0; half < 2; ++half) {
        const double* m = half ? q : s;
        s_red[(half * 4 + 0) * 256 + threadIdx.x] = m[0] * inv_nmu;
        s_red[(half * 4 + 1) * 256 + threadIdx.x] = (7.5 * m[1] - 2.5 * m[0]) * inv_nmu;
        s_red[(half * 4 + 2) * 256 + threadIdx.x] = (39.375 * m[2] - 33.75 * m[1] + 3.375 * m[0]) * inv_nmu;
        s_red[(half * 4 + 3) * 256 + threadIdx.x] = (187.6875 * m[3] - 255.9375 * m[2] + 85.3125 * m[1] - 4.0625 * m[0]) * inv_nmu;
    }
    __syncthreads();
    if (bad && valid && walker_ok) atomicOr(&D.status[b], VMX_STATUS_ARINYO);

    if (lt < KT && valid && walker_ok) {
        double damp = 1.0;
        if (d.damping_scale > 0.0) damp = exp(-d.damping_scale * d.damping_scale * pow(k, (double)d.damping_power) / 2.0);
        const size_t ncols = (size_t)B * D.n_active;
        for (int half = 0; half < (T.paired ? 2 : 1); ++half) {
            const int pipe = half ? pp : p;
            const int kind = D.pipes[pipe].d.pk_lin_kind;
            const double pk = damp * ((D.pk_direct && kind == VMX_PKLIN_SMOOTH) ? D.pk_direct[(size_t)b * D.nkp + i]
                                                                                 : D.pklin[(size_t)kind * D.nkp + i]);
            const size_t col = (size_t)D.pipes[pipe].col * B + b;
            for (int e = 0; e < D.n_ell; ++e) {
                double sum = 0.0;
                for (int qq = 0; qq < MS; ++qq) sum += s_red[(half * 4 + e) * 256 + (wb * MS + qq) * KT + kk];
                D.pl[((size_t)e * ncols + col) * D.nkp + i] = pk * sum;
            }
        }
    }
}

// The P(k,mu) stage of the core groups at table level 2 (EngineDev::xtab_level), as its own lean kernel: per walker and
// (k, mu) node only the tracer amplitudes are left -
//   P_s = P_lin A1 A2 T_s(k,mu) [/ sqrt(1 + (k mu sigma_v)^2)],  P_q = the same with the peak partner's table T_q,
//   A1 = b1 (1 + beta1 mu^2) + F_hcd(k mu) b_hcd (1 + beta_hcd mu^2)              (power_spectrum.py:163-222, :263-380)
// - so the kernel needs half the registers of k_pk_multipoles (its other loops carry the exponentials' constants) and
// twice the waves hide the latency of the two table streams.  Block = KT wavenumbers x MS mu-slices of NW walkers (each
// thread evaluates its nodes for NW walkers: one pair of table entries serves all of them).
// grid = (ceil(B / NW), level-2 groups - the costlier cross groups first -, k tiles: the launch ends on the dead tiles); LDS: (mu^2, mu^4) of the midpoints and {mu, mu^2, mu^4, w} of the extra
// nodes, reused as the [NW][8][KT MS] reduction scratch.
// what k_pk_tab2 needs to know about a group, passed in the kernel arguments (no descriptor loads ahead of the set-up)
struct Tab2Group { int32_t pipe, partner, xtab, cross, kind_s, kind_q, col_s, col_q, uvb, heii, lya1, lya2, damping_power, pad;
                   double damping_scale; };
#define VMX_TAB2_GROUPS 8
struct Tab2Args { Tab2Group g[VMX_TAB2_GROUPS]; };

// MODE: whose choice the mu rule is.  0: every walker of the block agrees (the common case: this instantiation is the
// kernel's hot path) - the rule when all lie in its box, the reference's loop when none does.  A block whose walkers DISAGREE
// (a sampler's stray point next to an ordinary one) runs the body twice: MODE 1 = the rule, stored for the walkers inside
// only; MODE 2 = the loop, stored for those outside - so that a walker's arithmetic never depends on its neighbours in the
// batch: bitwise the same alone, paired, or on another rank (vega_amd/parallel.py).
// table rows requested ahead of their use (two streams: the pipeline's table and its peak partner's)
#ifndef VMX_TAB2_PF
#define VMX_TAB2_PF 4
#endif
template <int KT, int MS, int NW, bool CROSS, int MODE = 0>
__device__ __forceinline__ void pk_tab2_body(const EngineDev& D, const Tab2Group& G, int B)
{
    extern __shared__ double smem[];
    constexpr int NT = KT * MS;
    // (the reduction scratch takes the place of the node tables once the loops are done: five or six blocks share a CU)
    double* s_red = smem;
    v2d* s_mu24 = (v2d*)smem;
    v4d* s_node = (v4d*)(smem + 2 * D.n_mu);
    const int p = G.pipe, pp = G.partner, xt = G.xtab;
    constexpr bool cross = CROSS;
    const int tile = blockIdx.z;
    const unsigned long long t_start = D.pk_trace ? wall_clock64() : 0ull;
    const int kk = threadIdx.x % KT;
    const int ms = (KT == 64) ? __builtin_amdgcn_readfirstlane(threadIdx.x / KT) : (int)(threadIdx.x / KT);
    const int i = tile * KT + kk;
    const bool valid = i < D.nk;
    const int ic = valid ? i : D.nk - 1;
    const int n_mu = D.n_mu;
    const double inv_nmu = 1.0 / (double)n_mu;
    // A block's set-up used to be a third of its life (block phase stamps, round 4: 7 - 9 us of ~20, 15 in the first round of
    // blocks): seven dependent round trips - the tile's bound, the walkers' scalars, four passes of the (mu^2, mu^4) table
    // through registers, the extra nodes - each waited for before the next was asked for.  Now everything the set-up reads
    // is requested at once: the node tables as direct global -> LDS copies of their static image (EngineDev::mu_img:
    // [n_mu] {mu^2, mu^4}, [n_extra] {mu, mu^2, mu^4, w} - the LDS layout below), the tile's and the walkers' values into
    // registers; then the block decides.
    {
        const unsigned total = (2u * (unsigned)n_mu + 4u * (unsigned)D.n_extra) * 8u;           // bytes, a multiple of 16
        const char* img = (const char*)D.mu_img;
        const unsigned wave_off = (threadIdx.x >> 6) * 1024u, lane_off = (threadIdx.x & 63) * 16u;
        for (unsigned c = 0; c < total; c += (unsigned)NT * 16u) {
            const unsigned off = c + wave_off;                  // (a wave's 64 x 16 bytes land contiguously at its LDS base)
            if (off + lane_off < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + off + lane_off),
                                                 (__attribute__((address_space(3))) void*)((char*)smem + off), 16, 0, 0);
        }
    }
    // the exponents are shared by the batch at this level: their bound over mu in (0, 1] came with the table (k_xtab).  A
    // tile whose every value underflows is skipped (as in k_pk_multipoles).
    const double* kx = D.xtab_k + (size_t)xt * 4 * D.nkp + ic;
    const double e_max = kx[3 * D.nkp];
    const double bad_flag = kx[2 * D.nkp];
    const double k = D.k[ic], k2 = k * k;
    bool ok[NW];
    double raw[NW][14];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        int b = blockIdx.x * NW + w;
        ok[w] = b < B;
        if (!ok[w]) b = B - 1;                        // a surplus walker slot shadows the last walker and stores nothing
        const double* sc = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;
        raw[w][0] = sc[S_BIAS1]; raw[w][1] = sc[S_BIAS2]; raw[w][2] = sc[S_BB1]; raw[w][3] = sc[S_BB2];
        raw[w][4] = sc[S_HCD_B]; raw[w][5] = sc[S_HCD_BB]; raw[w][6] = sc[S_HCD_L0]; raw[w][7] = sc[S_VD2]; raw[w][8] = sc[S_NO_RULE];
        raw[w][9] = G.uvb ? sc[S_UV_LAM] : 0.0; raw[w][10] = G.uvb ? sc[S_UV_BG] : 0.0; raw[w][11] = (G.uvb || G.heii) ? sc[S_UV_BP] : 0.0;
        raw[w][12] = G.heii ? sc[S_HE_LAM] : 0.0; raw[w][13] = G.heii ? sc[S_HE_BG] : 0.0;
    }
    // ONE barrier for the whole set-up (there were three: the tile's liveness vote, the rule's vote, the tables): every wave
    // leaves its two votes and its share of the UV / HeII terms in LDS, waits for its table copies, and meets the others once.
    // k-dependent effective bias from UV / HeII (power_spectrum.py:224-261): an arctangent and two divisions per (walker,
    // wavenumber, term) that do not depend on the mu slice - the block's MS slices used to compute each of them MS times
    // (a sixth of the block's instructions).  Role term NW + w - term `term` of walker w for the tile's wavenumbers - is
    // computed by slice role % MS; everyone reads it from LDS behind the node tables.
    double* s_x = smem + 2 * (size_t)n_mu + 4 * (size_t)D.n_extra;         // [2][NW][KT]
    int* s_vote = (int*)(s_x + 2 * NW * KT);                               // [NT / 64][2]
    if (G.uvb || G.heii) {
#pragma unroll
        for (int role = 0; role < 2 * NW; ++role) {
            if (role % MS != ms) continue;
            const int term = role / NW, wsel = role % NW;
            const double lam = term ? raw[wsel][12] : raw[wsel][9], bg = term ? raw[wsel][13] : raw[wsel][10], bp = raw[wsel][11];
            double v = 0.0;
            if (term ? G.heii : G.uvb) { const double x = k * lam; const double W = atan(x) / x; v = bg * W / (1.0 + bp * W); }
            s_x[(size_t)role * KT + kk] = v;
        }
    }
    {
        const unsigned long long live_v = __ballot(!(e_max < VMX_PK_DEAD));
        const unsigned long long rule_v = __ballot(k <= D.k_node_max || e_max < VMX_PK_NEGLIGIBLE), all_v = __ballot(true);
        if ((threadIdx.x & 63) == 0) {
            s_vote[2 * (threadIdx.x >> 6)] = live_v != 0ull;
            s_vote[2 * (threadIdx.x >> 6) + 1] = rule_v == all_v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the node tables have landed
    __syncthreads();
    bool live_block = false, rule_ok = true;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) { live_block = live_block || s_vote[2 * wv] != 0; rule_ok = rule_ok && s_vote[2 * wv + 1] != 0; }
    if (!live_block) {
        if (threadIdx.x >= KT || !valid) return;
        const size_t ncols = (size_t)B * D.n_active;
        for (int w = 0; w < NW; ++w) {
            const int b = blockIdx.x * NW + w;
            if (b >= B) continue;
            for (int half = 0; half < 2; ++half)
                for (int e = 0; e < D.n_ell; ++e)
                    D.pl[((size_t)e * ncols + (size_t)(half ? G.col_q : G.col_s) * B + b) * D.nkp + i] = 0.0;
        }
        return;
    }
    const double dmu = (double)MS * inv_nmu;
    const bool bad = bad_flag != 0.0;
    double c01[NW], c11[NW], c02[NW], c12[NW], hb[NW], hbb[NW], fk[NW], Fq[NW], k2vd2[NW];
    bool in_box = true;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        c01[w] = raw[w][0]; c02[w] = raw[w][1]; c11[w] = raw[w][2]; c12[w] = raw[w][3];
        hb[w] = raw[w][4]; hbb[w] = raw[w][5];
        fk[w] = -raw[w][6] * k;
        Fq[w] = vmx_exp(fk[w] * dmu);
        k2vd2[w] = k2 * raw[w][7];
        const bool inside = raw[w][8] == 0.0;
        in_box = in_box && inside;
        if (MODE == 1) ok[w] = ok[w] && inside;         // (a surplus slot shadows the last walker: stores nothing either way)
        if (MODE == 2) ok[w] = ok[w] && !inside;
    }
    if (MODE == 1) in_box = true;
    if (MODE == 2) in_box = false;
    if (threadIdx.x == 0) atomicMax(D.k_live, min((tile + 1) * KT, D.nk));
    // the node rule (first mu_lo and last mu_hi midpoints plus the extra nodes) serves a tile whose wavenumbers are all
    // within its range or negligible (VMX_PK_NEGLIGIBLE) - and whose walkers lie in the box the rule is validated on
    const bool node_mode = D.n_extra > 0 && in_box && rule_ok;
    if (node_mode && threadIdx.x == 0) atomicMax(D.k_live + 1, min((tile + 1) * KT, D.nk));
    const int lo_end = node_mode ? D.mu_lo : n_mu, hi_beg = node_mode ? n_mu - D.mu_hi : n_mu;
    if (G.uvb || G.heii) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            double add = 0.0;
            if (G.uvb) add += s_x[(size_t)w * KT + kk];
            if (G.heii) add += s_x[(size_t)(NW + w) * KT + kk];
            if (G.lya1) c01[w] += add;
            if (G.lya2) c02[w] += add;
        }
    }
#ifdef VMX_EXP_PK_PHASE     /* experiment build (scripts/gpu_pk_phase.py): the trace's last two words are phase stamps, not hardware ids */
    const unsigned long long t_setup = D.pk_trace ? wall_clock64() : 0ull;
#endif

    double s[NW][4], q[NW][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int n = 0; n < 4; ++n) { s[w][n] = 0.0; q[w][n] = 0.0; }

    {
        const size_t plane = (size_t)D.n_rows * D.nkp, row = (size_t)D.nkp, stride = (size_t)MS * D.nkp;
        const double* base = D.xtab + (size_t)xt * 2 * plane + (size_t)ms * row + ic;
        // one node for all NW walkers: AA = A1 A2 [/ sqrt(..)] [x weight]; the pipeline's table entry g, its partner's h
#define VMX_TAB2_NODE(MU2, MU4, WGT, G_, H_)                                                                          \
        {                                                                                                             \
            const double mu6 = (MU4) * (MU2);                                                                         \
            _Pragma("unroll")                                                                                         \
            for (int w = 0; w < NW; ++w) {                                                                            \
                const double A1 = fma(F[w], fma(hbb[w], (MU2), hb[w]), fma(c11[w], (MU2), c01[w]));                   \
                double AA = cross ? A1 * fma(c12[w], (MU2), c02[w]) : A1 * A1;                                        \
                if (cross) AA *= vmx_rsqrt(fma(k2vd2[w], (MU2), 1.0));                                                \
                if (WGT) AA *= wgt;                                                                                   \
                const double val = AA * (G_), vp = AA * (H_);                                                         \
                s[w][0] += val; s[w][1] = fma((MU2), val, s[w][1]); s[w][2] = fma((MU4), val, s[w][2]); s[w][3] = fma(mu6, val, s[w][3]); \
                q[w][0] += vp; q[w][1] = fma((MU2), vp, q[w][1]); q[w][2] = fma((MU4), vp, q[w][2]); q[w][3] = fma(mu6, vp, q[w][3]);     \
            }                                                                                                         \
        }
        double F[NW];
        const double wgt = 1.0;
        // midpoint ranges: both table streams run four steps ahead of their use, in fixed registers used in turn (a rotating
        // window would make every use wait for ALL outstanding loads); rows past a range's end exist (other rows / padding)
        for (int rg = 0; rg < (node_mode ? 2 : 1); ++rg) {
            const int j_lo = rg == 0 ? 0 : hi_beg, j_hi = rg == 0 ? lo_end : n_mu;
            const double* tab = base + (size_t)j_lo * row;
            double gw[VMX_TAB2_PF], hw[VMX_TAB2_PF];
#pragma unroll
            for (int u = 0; u < VMX_TAB2_PF; ++u) { gw[u] = tab[u * stride]; hw[u] = tab[plane + u * stride]; }
            tab += VMX_TAB2_PF * stride;
            for (int j0 = j_lo + ms; j0 < j_hi; j0 += MS * PK_REANCHOR) {
                // exact anchor of the HCD progression F = exp(-L0 k mu) along this thread's mu sequence
#pragma unroll
                for (int w = 0; w < NW; ++w) F[w] = vmx_exp(fk[w] * (((double)j0 + 0.5) * inv_nmu));
                const int jend = min(j0 + MS * PK_REANCHOR, j_hi);
                const int steps = (jend - j0 + MS - 1) / MS;
#define VMX_TAB2_STEP(GREG, HREG, J)                                                                                  \
                {                                                                                                     \
                    const v2d mm = s_mu24[J];                                                                         \
                    const double g = GREG, h = HREG;                                                                  \
                    GREG = *tab; HREG = tab[plane]; tab += stride;                                                    \
                    VMX_TAB2_NODE(mm.x, mm.y, false, g, h)                                                            \
                    _Pragma("unroll")                                                                                 \
                    for (int w = 0; w < NW; ++w) F[w] *= Fq[w];                                                       \
                }
                int j = j0;
                for (int it = 0; it < steps / VMX_TAB2_PF; ++it, j += VMX_TAB2_PF * MS) {
#pragma unroll
                    for (int u = 0; u < VMX_TAB2_PF; ++u) VMX_TAB2_STEP(gw[u], hw[u], j + u * MS)
                }
#pragma unroll
                for (int u = 0; u < VMX_TAB2_PF - 1; ++u)
                    if (u < steps % VMX_TAB2_PF) VMX_TAB2_STEP(gw[u], hw[u], j + u * MS)
#undef VMX_TAB2_STEP
            }
        }
        if (node_mode) {
            // the extra nodes (rows n_mu + jj): not equally spaced, so the HCD factor is exponentiated directly; the table
            // entries run four nodes ahead as above
            const double* tab = base + (size_t)n_mu * row;
            double gw[VMX_TAB2_PF], hw[VMX_TAB2_PF];
#pragma unroll
            for (int u = 0; u < VMX_TAB2_PF; ++u) { gw[u] = tab[u * stride]; hw[u] = tab[plane + u * stride]; }
            tab += VMX_TAB2_PF * stride;
            const int steps = D.n_extra > ms ? (D.n_extra - ms + MS - 1) / MS : 0;
#define VMX_TAB2_XSTEP(GREG, HREG, JJ)                                                                                \
            {                                                                                                         \
                const v4d nd = s_node[JJ];                                                                            \
                const double wgt = nd.w;                                                                              \
                const double g = GREG, h = HREG;                                                                      \
                GREG = *tab; HREG = tab[plane]; tab += stride;                                                        \
                _Pragma("unroll")                                                                                     \
                for (int w = 0; w < NW; ++w) F[w] = vmx_exp(fk[w] * nd.x);                                            \
                VMX_TAB2_NODE(nd.y, nd.z, true, g, h)                                                                 \
            }
            int jj = ms;
            for (int it = 0; it < steps / VMX_TAB2_PF; ++it, jj += VMX_TAB2_PF * MS) {
#pragma unroll
                for (int u = 0; u < VMX_TAB2_PF; ++u) VMX_TAB2_XSTEP(gw[u], hw[u], jj + u * MS)
            }
#pragma unroll
            for (int u = 0; u < VMX_TAB2_PF - 1; ++u)
                if (u < steps % VMX_TAB2_PF) VMX_TAB2_XSTEP(gw[u], hw[u], jj + u * MS)
#undef VMX_TAB2_XSTEP
        }
#undef VMX_TAB2_NODE
    }

    // moments -> Legendre multipoles  P_ell = (2 ell + 1) / n_mu * sum_n c_{ell n} M_n  (pktoxi.py:37,55,138)
#ifdef VMX_EXP_PK_PHASE
    const unsigned long long t_loop = D.pk_trace ? wall_clock64() : 0ull;
#endif
    __syncthreads();            // every wave is done with the node tables
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const double* m = half ? q[w] : s[w];
            double* r = s_red + (size_t)w * 8 * NT + (half * 4) * NT + threadIdx.x;
            r[0] = m[0] * inv_nmu;
            r[NT] = (7.5 * m[1] - 2.5 * m[0]) * inv_nmu;
            r[2 * NT] = (39.375 * m[2] - 33.75 * m[1] + 3.375 * m[0]) * inv_nmu;
            r[3 * NT] = (187.6875 * m[3] - 255.9375 * m[2] + 85.3125 * m[1] - 4.0625 * m[0]) * inv_nmu;
        }
    __syncthreads();
    if (D.pk_trace && threadIdx.x == 0) {
        unsigned long long* tr = D.pk_trace + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        tr[0] = t_start; tr[1] = wall_clock64();
#ifdef VMX_EXP_PK_PHASE
        tr[2] = t_setup; tr[3] = t_loop;
#else
        tr[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tr[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
#endif
    }
    if (threadIdx.x >= KT || !valid) return;
    double damp = 1.0;
    if (G.damping_scale > 0.0) damp = exp(-G.damping_scale * G.damping_scale * pow(k, (double)G.damping_power) / 2.0);
    const size_t ncols = (size_t)B * D.n_active;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (!ok[w]) continue;
        const int b = blockIdx.x * NW + w;
        if (bad) atomicOr(&D.status[b], VMX_STATUS_ARINYO);
        for (int half = 0; half < 2; ++half) {
            const int kind = half ? G.kind_q : G.kind_s;
            const double pk = damp * ((D.pk_direct && kind == VMX_PKLIN_SMOOTH) ? D.pk_direct[(size_t)b * D.nkp + i]
                                                                                 : D.pklin[(size_t)kind * D.nkp + i]);
            const size_t col = (size_t)(half ? G.col_q : G.col_s) * B + b;
            for (int e = 0; e < D.n_ell; ++e) {
                double sum = 0.0;
                for (int qq = 0; qq < MS; ++qq) sum += s_red[(size_t)w * 8 * NT + (half * 4 + e) * NT + qq * KT + kk];
                D.pl[((size_t)e * ncols + col) * D.nkp + i] = pk * sum;
            }
        }
    }
}

// the two passes of a block whose walkers disagree on the rule (inlined: as a function of its own it would bring a stack,
// i.e. scratch memory, to every launch of the kernel)
template <int KT, int MS, int NW>
__device__ __forceinline__ void pk_tab2_mixed(const EngineDev& D, const Tab2Group& G, int B)
{
    if (G.cross) pk_tab2_body<KT, MS, NW, true, 1>(D, G, B); else pk_tab2_body<KT, MS, NW, false, 1>(D, G, B);
    __syncthreads();        // (the reduction scratch of the first pass shares its LDS with the second pass's node tables)
    if (G.cross) pk_tab2_body<KT, MS, NW, true, 2>(D, G, B); else pk_tab2_body<KT, MS, NW, false, 2>(D, G, B);
}

// waves per SIMD the register allocation aims at (measured: three or four waves of the two-walker shape run the same)
#ifndef VMX_TAB2_BLOCKS
#define VMX_TAB2_BLOCKS 4
#define VMX_TAB2_W2 3
#endif
template <int KT, int MS, int NW>
__global__ __launch_bounds__(KT * MS, KT != 64 ? 4 : (NW == 1 ? VMX_TAB2_BLOCKS : VMX_TAB2_W2) * (4 / MS)) void k_pk_tab2(EngineDev D, Tab2Args A, int B)
{
    const Tab2Group& G = A.g[blockIdx.y];
    if constexpr (NW > 1) {
        // do the walkers of this block agree on the mu rule? (block-uniform: the flags are the prologue's, per walker)
        bool any_in = false, all_in = true;
        for (int w = 0; w < NW; ++w) {
            const int b = min((int)blockIdx.x * NW + w, B - 1);
            const bool inside = D.scal[((size_t)b * D.n_pipe + G.pipe) * VMX_NS + S_NO_RULE] == 0.0;
            any_in = any_in || inside; all_in = all_in && inside;
        }
        if (any_in && !all_in) { pk_tab2_mixed<KT, MS, NW>(D, G, B); return; }
    }
    if (G.cross) pk_tab2_body<KT, MS, NW, true>(D, G, B);
    else pk_tab2_body<KT, MS, NW, false>(D, G, B);
}

// `fht_extrap` (reference pktoxi.py:41,141; mcfit's `_pad(extrap=True)`): behind the nk samples F of every P_ell row the
// engine writes the FFTLog's power-law pads - F[0] (F[1] / F[0])^-t for t = 1 .. pad_l, then F[nk-1] (F[nk-1] / F[nk-2])^t for
// t = 1 .. pad_r - which the padded operator's columns multiply (vmx_set_fftlog_padding).  IEEE arithmetic as numpy's:
// 0 / 0 end segments give NaN.  One block per (multipole, column, walker) row; multipoles a pipeline does not use stay zero.
__global__ __launch_bounds__(256) void k_pk_extrap(EngineDev D, int B)
{
    const size_t ncols = (size_t)B * D.n_active;
    const int e = (int)(blockIdx.x / ncols);
    const int col = (int)((blockIdx.x % ncols) / B);
    double* row = D.pl + (size_t)blockIdx.x * D.nkp;
    const bool used = e < D.pipes[D.pipe_active[col]].d.n_ell;
    const double f0 = row[0], f1 = row[1], fm = row[D.nk - 2], fn = row[D.nk - 1];
    const double ratio_l = f1 / f0, ratio_r = fn / fm;
    for (int t = threadIdx.x; t < D.pad_l + D.pad_r; t += 256) {
        double v = 0.0;
        if (used) v = t < D.pad_l ? f0 * pow(ratio_l, -(double)(t + 1)) : fn * pow(ratio_r, (double)(t - D.pad_l + 1));
        row[D.nk + t] = v;
    }
}

// The shared-W groups (e.g. QSO x every metal line: pipelines that differ in their Kaiser polynomials only) in their own lean
// kernel: six even moments of W(k, mu) = G(k, mu) exp(e0 + e1 mu^2) / sqrt((1 + (k mu s1)^2)(1 + (k mu s2)^2)) per walker
// and wavenumber, NW walkers per thread (one entry of the static G table and one (mu^2, mu^4) pair serve all of them), then
// every member's multipoles (w_members_store).  Block = 64 wavenumbers x 4 mu slices; grid = (ceil(B / NW), groups of the
// list, k tiles); LDS as k_pk_tab2 (node tables, reused as the [NW][6][256] reduction scratch).
// (MODE: as in pk_tab2_body - 0 the walkers of the block agree on the mu rule, 1 / 2 the two passes of a block whose walkers do not)
template <int NW, int MODE>
__device__ __forceinline__ void pk_w_body(const EngineDev& D, const PkGroup* groups, const int32_t* members, const int32_t* wlist, int B)
{
    extern __shared__ double smem[];
    constexpr int KT = 64, MS = 4;
    double* s_red = smem;
    v2d* s_mu24 = (v2d*)smem;
    v4d* s_node = (v4d*)(smem + 2 * D.n_mu);             // {mu^2, mu^4, mu^6, w} of the extra nodes
    const PkGroup& G = groups[wlist[blockIdx.y]];
    const int p = G.pipe;
    const vmx_pipe_desc& d = D.pipes[p].d;
    const int tile = blockIdx.z;
    const int kk = threadIdx.x % KT;
    const int ms = __builtin_amdgcn_readfirstlane(threadIdx.x / KT);
    const int i = tile * KT + kk;
    const bool valid = i < D.nk;
    const int ic = valid ? i : D.nk - 1;
    const int n_mu = D.n_mu;
    const double inv_nmu = 1.0 / (double)n_mu;
    const double k = D.k[ic], k2 = k * k;
    double e0[NW], e1[NW], k2vd1[NW], k2vd2[NW];
    bool noexp[NW], keep[NW];
    double e_max = -1e300;
    bool in_box = true;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        int b = blockIdx.x * NW + w;
        keep[w] = b < B;
        if (b >= B) b = B - 1;                        // a surplus walker slot shadows the last walker and stores nothing
        const double* sc = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;
        const double ga = sc[S_GA], gb = sc[S_GB];
        e0[w] = -k2 * gb; e1[w] = -k2 * (ga - gb);
        noexp[w] = (ga == 0.0) && (gb == 0.0);
        k2vd1[w] = k2 * sc[S_VD1]; k2vd2[w] = k2 * sc[S_VD2];
        e_max = fmax(e_max, e0[w] + fmax(e1[w], 0.0));
        const bool inside = sc[S_NO_RULE] == 0.0;
        in_box = in_box && inside;
        if (MODE == 1) keep[w] = keep[w] && inside;
        if (MODE == 2) keep[w] = keep[w] && !inside;
    }
    if (MODE == 1) in_box = true;
    if (MODE == 2) in_box = false;
    // ONE barrier for the set-up, as in pk_tab2_body: the node tables as direct global -> LDS copies of their static image
    // (EngineDev::mu_img_w: [n_mu] {mu^2, mu^4}, [n_extra] {mu^2, mu^4, mu^6, w}), the waves' two votes in LDS behind them
    int* s_vote = (int*)(smem + 2 * (size_t)n_mu + 4 * (size_t)D.n_extra);
    {
        const unsigned total = (2u * (unsigned)n_mu + 4u * (unsigned)D.n_extra) * 8u;
        const char* img = (const char*)D.mu_img_w;
        const unsigned wave_off = (threadIdx.x >> 6) * 1024u, lane_off = (threadIdx.x & 63) * 16u;
        for (unsigned c = 0; c < total; c += 256u * 16u) {
            const unsigned off = c + wave_off;
            if (off + lane_off < total)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + off + lane_off),
                                                 (__attribute__((address_space(3))) void*)((char*)smem + off), 16, 0, 0);
        }
        const unsigned long long live_v = __ballot(!(e_max < VMX_PK_DEAD));
        const unsigned long long rule_v = __ballot(k <= D.k_node_max || e_max < VMX_PK_NEGLIGIBLE), all_v = __ballot(true);
        if ((threadIdx.x & 63) == 0) {
            s_vote[2 * (threadIdx.x >> 6)] = live_v != 0ull;
            s_vote[2 * (threadIdx.x >> 6) + 1] = rule_v == all_v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool live_block = false, rule_ok = true;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { live_block = live_block || s_vote[2 * wv] != 0; rule_ok = rule_ok && s_vote[2 * wv + 1] != 0; }
    if (live_block && threadIdx.x == 0) atomicMax(D.k_live, min((tile + 1) * KT, D.nk));
    const bool node_mode = D.n_extra > 0 && in_box && rule_ok;
    if (live_block && node_mode && threadIdx.x == 0) atomicMax(D.k_live + 1, min((tile + 1) * KT, D.nk));
    const int lo_end = node_mode ? D.mu_lo : n_mu, hi_beg = node_mode ? n_mu - D.mu_hi : n_mu;

    double wm[NW][6];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int n = 0; n < 6; ++n) wm[w][n] = 0.0;
    if (live_block) {
        const bool has_g = d.gk_table >= 0;
        const size_t row = (size_t)D.nkp, stride = (size_t)MS * D.nkp;
        // (without a G table the stream reads the mu table's own storage: any finite values, multiplied out below)
        const double* base = has_g ? D.gk + (size_t)d.gk_table * D.n_rows * D.nkp + (size_t)ms * row + ic : D.k + ic;
        const size_t st = has_g ? stride : 0, rw = has_g ? row : 0;
#define VMX_W_NODE(MU2, MU4, MU6, WGT, G_)                                                                            \
        {                                                                                                             \
            _Pragma("unroll")                                                                                         \
            for (int w = 0; w < NW; ++w) {                                                                            \
                double val = has_g ? (G_) : 1.0;                                                                      \
                if (WGT) val *= wgt;                                                                                  \
                if (!noexp[w]) val *= vmx_exp(fma(e1[w], (MU2), e0[w]));                                              \
                if (k2vd1[w] != 0.0) val *= vmx_rsqrt(fma(k2vd1[w], (MU2), 1.0));                                     \
                if (k2vd2[w] != 0.0) val *= vmx_rsqrt(fma(k2vd2[w], (MU2), 1.0));                                     \
                wm[w][0] += val; wm[w][1] = fma((MU2), val, wm[w][1]); wm[w][2] = fma((MU4), val, wm[w][2]);          \
                const double v6 = val * (MU6);                                                                        \
                wm[w][3] += v6; wm[w][4] = fma((MU2), v6, wm[w][4]); wm[w][5] = fma((MU4), v6, wm[w][5]);             \
            }                                                                                                         \
        }
        const double wgt = 1.0;
        for (int rg = 0; rg < (node_mode ? 2 : 1); ++rg) {
            const int j_lo = rg == 0 ? 0 : hi_beg, j_hi = rg == 0 ? lo_end : n_mu;
            const double* tab = base + (size_t)j_lo * rw;
            double g0 = *tab, g1 = tab[st], g2 = tab[2 * st], g3 = tab[3 * st];
            tab += 4 * st;
            const int steps = j_hi > j_lo + ms ? (j_hi - j_lo - ms + MS - 1) / MS : 0;
#define VMX_W_STEP(GREG, J)                                                                                           \
            {                                                                                                         \
                const v2d mm = s_mu24[J];                                                                             \
                const double g = GREG;                                                                                \
                GREG = *tab; tab += st;     /* (rows past a range's end exist: other rows / padding) */               \
                VMX_W_NODE(mm.x, mm.y, mm.x * mm.y, false, g)                                                         \
            }
            int j = j_lo + ms;
            for (int it = 0; it < steps / 4; ++it, j += 4 * MS) {
                VMX_W_STEP(g0, j)
                VMX_W_STEP(g1, j + MS)
                VMX_W_STEP(g2, j + 2 * MS)
                VMX_W_STEP(g3, j + 3 * MS)
            }
            if (steps % 4 > 0) VMX_W_STEP(g0, j)
            if (steps % 4 > 1) VMX_W_STEP(g1, j + MS)
            if (steps % 4 > 2) VMX_W_STEP(g2, j + 2 * MS)
#undef VMX_W_STEP
        }
        if (node_mode) {
            const double* tab = base + (size_t)n_mu * rw;
            double g0 = *tab, g1 = tab[st], g2 = tab[2 * st], g3 = tab[3 * st];
            tab += 4 * st;
            const int steps = D.n_extra > ms ? (D.n_extra - ms + MS - 1) / MS : 0;
#define VMX_W_XSTEP(GREG, JJ)                                                                                         \
            {                                                                                                         \
                const v4d nd = s_node[JJ];                                                                            \
                const double wgt = nd.w;                                                                              \
                const double g = GREG;                                                                                \
                GREG = *tab; tab += st;                                                                               \
                VMX_W_NODE(nd.x, nd.y, nd.z, true, g)                                                                 \
            }
            int jj = ms;
            for (int it = 0; it < steps / 4; ++it, jj += 4 * MS) {
                VMX_W_XSTEP(g0, jj)
                VMX_W_XSTEP(g1, jj + MS)
                VMX_W_XSTEP(g2, jj + 2 * MS)
                VMX_W_XSTEP(g3, jj + 3 * MS)
            }
            if (steps % 4 > 0) VMX_W_XSTEP(g0, jj)
            if (steps % 4 > 1) VMX_W_XSTEP(g1, jj + MS)
            if (steps % 4 > 2) VMX_W_XSTEP(g2, jj + 2 * MS)
#undef VMX_W_XSTEP
        }
#undef VMX_W_NODE
    }
    __syncthreads();            // every wave is done with the node tables
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int n = 0; n < 6; ++n) s_red[(size_t)(w * 6 + n) * 256 + threadIdx.x] = wm[w][n];
    __syncthreads();
    if (threadIdx.x >= KT || !valid) return;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = blockIdx.x * NW + w;
        if (!keep[w]) continue;
        double tot[6];
        for (int n = 0; n < 6; ++n) {
            double sum = 0.0;
            for (int qq = 0; qq < MS; ++qq) sum += s_red[(size_t)(w * 6 + n) * 256 + qq * KT + kk];
            tot[n] = sum;
        }
        w_members_store(D, G, members, b, B, i, k, tot);
    }
}

template <int NW>
__device__ __forceinline__ void pk_w_mixed(const EngineDev& D, const PkGroup* groups, const int32_t* members, const int32_t* wlist, int B)
{
    pk_w_body<NW, 1>(D, groups, members, wlist, B);
    __syncthreads();
    pk_w_body<NW, 2>(D, groups, members, wlist, B);
}

template <int NW>
#ifndef VMX_PKW_OCC
#define VMX_PKW_OCC 4
#endif
__global__ __launch_bounds__(256, VMX_PKW_OCC) void k_pk_w(EngineDev D, const PkGroup* groups, const int32_t* members, const int32_t* wlist, int B)
{
    if constexpr (NW > 1) {
        const int p = groups[wlist[blockIdx.y]].pipe;
        bool any_in = false, all_in = true;
        for (int w = 0; w < NW; ++w) {
            const int b = min((int)blockIdx.x * NW + w, B - 1);
            const bool inside = D.scal[((size_t)b * D.n_pipe + p) * VMX_NS + S_NO_RULE] == 0.0;
            any_in = any_in || inside; all_in = all_in && inside;
        }
        if (any_in && !all_in) { pk_w_mixed<NW>(D, groups, members, wlist, B); return; }
    }
    pk_w_body<NW, 0>(D, groups, members, wlist, B);
}

// Pipelines whose only mu dependence is the Kaiser polynomial times the static G table (metal pairs without
// HCD / NL / smoothing / velocity dispersion): P(k,mu) = P_lin(k) (c0_1 + c1_1 mu^2)(c0_2 + c1_2 mu^2) G(k,mu), so
// the mu-moments are combinations of the table's own moments - no mu loop.  grid = (walkers, poly pipelines).
__global__ __launch_bounds__(256) void k_pk_poly(EngineDev D, const int32_t* poly_pipes, int B)
{
    const int b = blockIdx.x, p = poly_pipes[blockIdx.y];
    const vmx_pipe_desc& d = D.pipes[p].d;
    const double* sc = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;
    const double inv_nmu = 1.0 / (double)D.n_mu;
    const size_t ncols = (size_t)B * D.n_active, col = (size_t)D.pipes[p].col * B + b;
    const double* mg0 = D.gk_mom + (size_t)(d.gk_table >= 0 ? d.gk_table : D.n_gk) * 6 * D.nkp;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(D.k_live, D.nk);     // no damping factor: every wavenumber is live
    for (int i = threadIdx.x; i < D.nk; i += 256) {
        const double k = D.k[i];
        double c01 = sc[S_BIAS1], c02 = sc[S_BIAS2];
        const double c11 = sc[S_BB1], c12 = d.same_tracer ? sc[S_BB1] : sc[S_BB2];
        if (d.uvb || d.heii) {
            double add = 0.0;
            if (d.uvb) { const double x = k * sc[S_UV_LAM]; const double W = atan(x) / x; add += sc[S_UV_BG] * W / (1.0 + sc[S_UV_BP] * W); }
            if (d.heii) { const double x = k * sc[S_HE_LAM]; const double W = atan(x) / x; add += sc[S_HE_BG] * W / (1.0 + sc[S_UV_BP] * W); }
            if (d.tracer[0].is_lya) c01 += add;
            if (d.tracer[1].is_lya) c02 += add;
        }
        if (d.same_tracer) c02 = c01;
        const double a0 = c01 * c02, a1 = fma(c01, c12, c11 * c02), a2 = c11 * c12;
        const double* mg = mg0 + i;
        double mm[4];
        for (int m = 0; m < 4; ++m)
            mm[m] = fma(a2, mg[(size_t)(m + 2) * D.nkp], fma(a1, mg[(size_t)(m + 1) * D.nkp], a0 * mg[(size_t)m * D.nkp]));
        double damp = 1.0;
        if (d.damping_scale > 0.0) damp = exp(-d.damping_scale * d.damping_scale * pow(k, (double)d.damping_power) / 2.0);
        const double pk = damp * ((D.pk_direct && d.pk_lin_kind == VMX_PKLIN_SMOOTH) ? D.pk_direct[(size_t)b * D.nkp + i]
                                                                                     : D.pklin[(size_t)d.pk_lin_kind * D.nkp + i]) * inv_nmu;
        D.pl[((size_t)0 * ncols + col) * D.nkp + i] = pk * mm[0];
        D.pl[((size_t)1 * ncols + col) * D.nkp + i] = pk * (7.5 * mm[1] - 2.5 * mm[0]);
        D.pl[((size_t)2 * ncols + col) * D.nkp + i] = pk * (39.375 * mm[2] - 33.75 * mm[1] + 3.375 * mm[0]);
        D.pl[((size_t)3 * ncols + col) * D.nkp + i] = pk * (187.6875 * mm[3] - 255.9375 * mm[2] + 85.3125 * mm[1] - 4.0625 * mm[0]);
    }
}

// Kaiser-basis spectra of a pipeline with a static coefficient basis: with P(k,mu) = P_lin (a0 + a1 mu^2 + a2 mu^4) G(k,mu)
// the multipoles are a0 V0 + a1 V1 + a2 V2, V_i[ell](k) = P_lin(k) / n_mu * sum_n c_{ell n} M_{n+i}(k), M = moments of G.
// out: [n_ell][n_static][3][nkp] (the FFTLog o spline operator then turns every row into a coefficient vector).
__global__ __launch_bounds__(256) void k_poly_basis(EngineDev D, const int32_t* static_pipes, double* out)
{
    const int sb = blockIdx.y, p = static_pipes[sb];
    const vmx_pipe_desc& d = D.pipes[p].d;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D.nkp) return;
    const double inv_nmu = 1.0 / (double)D.n_mu;
    const double* mg = D.gk_mom + (size_t)(d.gk_table >= 0 ? d.gk_table : D.n_gk) * 6 * D.nkp + i;
    const double pk = i < D.nk ? D.pklin[(size_t)d.pk_lin_kind * D.nkp + i] * inv_nmu : 0.0;
    for (int q = 0; q < 3; ++q) {
        const double m0 = mg[(size_t)q * D.nkp], m1 = mg[(size_t)(q + 1) * D.nkp], m2 = mg[(size_t)(q + 2) * D.nkp],
                     m3 = mg[(size_t)(q + 3) * D.nkp];
        const size_t row = (size_t)sb * 3 + q, rows = (size_t)D.n_static * 3;
        out[((size_t)0 * rows + row) * D.nkp + i] = pk * m0;
        out[((size_t)1 * rows + row) * D.nkp + i] = pk * (7.5 * m1 - 2.5 * m0);
        out[((size_t)2 * rows + row) * D.nkp + i] = pk * (39.375 * m2 - 33.75 * m1 + 3.375 * m0);
        out[((size_t)3 * rows + row) * D.nkp + i] = pk * (187.6875 * m3 - 255.9375 * m2 + 85.3125 * m1 - 4.0625 * m0);
    }
}

// ------------------------------------------------------------------------------------------------
// broadband helpers
// ------------------------------------------------------------------------------------------------
__device__ inline double bb_total(const EngineDev& D, const ItemDev& it, int pos, const double* t, int bin, int n)
{
    const bool mul = (pos == VMX_BB_PRE_MUL || pos == VMX_BB_POST_MUL);
    double total = mul ? 1.0 : 0.0;
    for (int q = 0; q < it.n_bb[pos]; ++q) {
        const BBTermDev& term = it.bb[pos][q];
        const double* basis = D.bb_basis + term.basis_off;
        double corr = 0.0;
        if (term.func == VMX_BB_SKY) {
            const double scale = t[term.slot[0]], sigma = t[term.slot[1]];
            const double rt = basis[bin], w = basis[(size_t)n + bin];
            if (w != 0.0) {
                const double q2 = rt / sigma;
                corr = scale / (sigma * sqrt(2.0 * M_PI)) * exp(-0.5 * q2 * q2);
            }
        } else {
            for (int cidx = 0; cidx < term.n_coef; ++cidx) corr = fma(t[term.slot[cidx]], basis[(size_t)cidx * n + bin], corr);
        }
        if (mul) total *= (1.0 + corr); else total += corr;
    }
    return total;
}

// combine components, add metals, apply pre-distortion broadband (model.py:119-140,186)
// Metal matrix in Kronecker form (new_metals; metals.py:338-367, :501-655): out = A Xi B^T for one (walker, metal pair),
// Xi = the pair's correlation as [n_rp][n_rt].  One block per (walker, pair): Xi and the factors sit in LDS, the first
// product T = Xi B^T is written back to LDS, the second one out = A T goes to the metal-product buffer.  One launch
// covers every Kronecker-form metal of an item.
__global__ __launch_bounds__(256) void k_metal_kron(EngineDev D, int item, int B)
{
    extern __shared__ double sk[];
    const ItemDev& it = D.items[item];
    const MetalDev& md = D.metals[it.metal_begin + blockIdx.y];     // grid.y = every metal of the item
    if (!md.kron_a) return;                                         // (block-uniform)
    const PipeDev& P = D.pipes[md.d.pipeline];
    const int b = blockIdx.x;
    const int nrp = md.kron_nrp, nrt = md.kron_nrt, n = nrp * nrt;
    double* sx = sk;                    // Xi, then reused for nothing else
    double* st = sk + n;                // T = Xi B^T
    double* sm = sk + 2 * n;            // B, then A
    const double* x = D.xi + P.xi_off + (size_t)b * P.n_pad;
    for (int i = threadIdx.x; i < n; i += 256) sx[i] = x[i];
    if (md.kron_b) {
        for (int i = threadIdx.x; i < nrt * nrt; i += 256) sm[i] = md.kron_b[i];
        __syncthreads();
        // T[p][t] = sum_u Xi[p][u] B[t][u]; a thread owns 5 consecutive t of one p (one Xi read per 5 FMAs)
        const int nt5 = (nrt + 4) / 5;
        for (int w = threadIdx.x; w < nrp * nt5; w += 256) {
            const int p = w / nt5, tb = (w % nt5) * 5;
            const double* xr = sx + p * nrt;
            double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            for (int u = 0; u < nrt; ++u) {
                const double xv = xr[u];
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = fma(xv, sm[min(tb + i, nrt - 1) * nrt + u], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) if (tb + i < nrt) st[p * nrt + tb + i] = acc[i];
        }
        __syncthreads();
    } else {
        __syncthreads();
        for (int o = threadIdx.x; o < n; o += 256) st[o] = sx[o];
        __syncthreads();
    }
    for (int i = threadIdx.x; i < nrp * nrp; i += 256) sm[i] = md.kron_a[i];
    __syncthreads();
    // out[p][t] = sum_q A[p][q] T[q][t]; a thread owns 5 consecutive p of one t (one T read per 5 FMAs, t fastest over the
    // lanes: coalesced stores)
    double* out = D.xim + md.xim_off + (size_t)b * it.n_model_pad;
    const int np5 = (nrp + 4) / 5;
    for (int w = threadIdx.x; w < np5 * nrt; w += 256) {
        const int t = w % nrt, pb = (w / nrt) * 5;
        double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < nrp; ++q) {
            const double tv = st[q * nrt + t];
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = fma(sm[min(pb + i, nrp - 1) * nrp + q], tv, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) if (pb + i < nrp) out[(pb + i) * nrt + t] = acc[i];
    }
}

__device__ inline double assemble_bin(const EngineDev& D, const ItemDev& it, int b, int bin)
{
    const double* t = D.theta + (size_t)b * D.n_params;
    const double bao = t[it.d.bao_amp_slot];
    const PipeDev& Pp = D.pipes[it.d.pipe_peak];
    const PipeDev& Ps = D.pipes[it.d.pipe_smooth];
    const bool direct = D.pk_direct != nullptr;
    double v = 0.0;
    uint64_t summed = 0;
    bool peak_in = false, smooth_in = false;
    if (D.sums_on) {
        const ItemSums& sm = D.sums[&it - D.items];
        for (int a = 0; a < sm.n_arrays; ++a) v += D.xi[sm.off[a] + (size_t)b * sm.n_pad + bin];
        summed = sm.metal_mask; peak_in = sm.core_peak != 0; smooth_in = sm.core_smooth != 0;
    }
    if (!smooth_in) v += D.xi[Ps.xi_off + (size_t)b * Ps.n_pad + bin];
    if (!direct && !peak_in) v = fma(bao, D.xi[Pp.xi_off + (size_t)b * Pp.n_pad + bin], v);
    for (int m = 0; m < it.n_metals; ++m) {
        if ((summed >> m) & 1) continue;
        const MetalDev& md = D.metals[it.metal_begin + m];
        if (direct && !md.d.in_direct) continue;        // (direct_pk: the metal terms only with `no-metal-decomp = False`, smooth entries)
        const double* mb = D.metal_bias + (size_t)b * 3 * D.n_metals_total + it.metal_begin + m;
        const double f = mb[0];
        double x;
        if (md.basis) {
            const double* y = md.basis + bin;
            x = fma(mb[2 * D.n_metals_total], y[2 * (size_t)it.n_model_pad], fma(mb[D.n_metals_total], y[it.n_model_pad], y[0]));
        } else if (md.svec) x = md.svec[bin];
        else if (md.mat_off >= 0) x = D.xim[md.xim_off + (size_t)b * it.n_model_pad + bin];
        else { const PipeDev& Pm = D.pipes[md.d.pipeline]; x = D.xi[Pm.xi_off + (size_t)b * Pm.n_pad + bin]; }
        v = fma(f, x, v);
    }
    if (it.add_vec) v = fma(it.add_slot >= 0 ? t[it.add_slot] : it.add_default, it.add_vec[bin], v);
    if (it.n_bb[VMX_BB_PRE_MUL]) v *= bb_total(D, it, VMX_BB_PRE_MUL, t, bin, it.d.n_model);
    if (it.n_bb[VMX_BB_PRE_ADD]) v += (direct ? 1.0 : 1.0 + bao) * bb_total(D, it, VMX_BB_PRE_ADD, t, bin, it.d.n_model);
    return v;
}

__global__ __launch_bounds__(256) void k_assemble(EngineDev D, int item)
{
    const ItemDev& it = D.items[item];
    const int b = blockIdx.y;
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= it.d.n_model) return;
    it.vec[(size_t)b * it.n_model_pad + bin] = assemble_bin(D, it, b, bin);
}

// every item in one launch (blockIdx.z = item)
__global__ __launch_bounds__(256) void k_assemble_all(EngineDev D)
{
    const ItemDev& it = D.items[blockIdx.z];
    const int b = blockIdx.y;
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= it.d.n_model) return;
    it.vec[(size_t)b * it.n_model_pad + bin] = assemble_bin(D, it, b, bin);
}

// post-distortion broadband, model output and masked residual (model.py:147-149; vega_interface.py:310-315)
__device__ inline void post_bin(const EngineDev& D, const ItemDev& it, int b, int bin, double v)
{
    const double* t = D.theta + (size_t)b * D.n_params;
    if (it.n_bb[VMX_BB_POST_MUL]) v *= bb_total(D, it, VMX_BB_POST_MUL, t, bin, it.d.n_dist);
    if (it.n_bb[VMX_BB_POST_ADD])
        v += (D.pk_direct ? 1.0 : 1.0 + t[it.d.bao_amp_slot]) * bb_total(D, it, VMX_BB_POST_ADD, t, bin, it.d.n_dist);
    D.model[(size_t)b * D.model_size + it.model_off + bin] = v;
    const int mi = it.inv_mask[bin];
    if (mi >= 0) {
        const int mock = D.mock_index[b];
        const double dat = (mock >= 0 && it.mock_pool) ? it.mock_pool[(size_t)mock * it.n_masked + mi] : it.data[mi];
        const double diff = dat - v;
        it.res[(size_t)b * it.n_masked_pad + mi] = diff;
        if (D.gres) D.gres[(size_t)b * D.g_ld + it.masked_off + mi] = diff;
    }
}

__global__ __launch_bounds__(256) void k_post(EngineDev D, int item, int B, int dist_slabs)
{
    const ItemDev& it = D.items[item];
    const int b = blockIdx.y;
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= it.d.n_dist) return;
    double v;
    if (it.dm || it.dm_ptr) {
        v = 0.0;
        for (int s = 0; s < dist_slabs; ++s) v += it.dist[((size_t)s * B + b) * it.n_dist_pad + bin];
    } else v = it.vec[(size_t)b * it.n_model_pad + bin];
    post_bin(D, it, b, bin, v);
}

__global__ __launch_bounds__(256) void k_post_all(EngineDev D, int B, SlabInfo dist_slabs)
{
    const ItemDev& it = D.items[blockIdx.z];
    const int b = blockIdx.y;
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= it.d.n_dist) return;
    double v;
    if (it.dm || it.dm_ptr) {
        v = 0.0;
        for (int s = 0; s < dist_slabs.z[blockIdx.z]; ++s) v += it.dist[((size_t)s * B + b) * it.n_dist_pad + bin];
    } else v = it.vec[(size_t)b * it.n_model_pad + bin];
    post_bin(D, it, b, bin, v);
}

// ------------------------------------------------------------------------------------------------
// Static quadratic form of chi2 (chi2-only evaluations without a multiplicative post-distortion broadband).
//
// With x' = [x ; a] (x the pre-distortion vector of k_assemble, a_j = (1 + bao) c_j the coefficients of the additive
// polynomial post-distortion broadband) the model on the masked bins is S DM' x', DM' = [DM | B] with the broadband
// basis B as extra columns, so the residual r = d - S DM' x' and
//     chi2 = r0^T C^-1 r0 - 2 dx^T (DM'^T S^T C^-1 r0) + dx^T (DM'^T S^T C^-1 S DM') dx,     dx = x' - x0',
// expanded around a reference point x0' (r0 = d - S DM' x0': its residual), which keeps the three terms of the
// size of the walkers' spread instead of the size of the signal (no cancellation of large numbers: finite-difference
// gradients of a minimiser stay clean).  The matrix Q' = DM'^T S^T C^-1 S DM' is static and symmetric: ONE half-triangle
// product per item replaces the distortion product (model.py:143-144) and the C^-1 product (vega_interface.py:316).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble_quad(EngineDev D)
{
    const ItemDev& it = D.items[blockIdx.z];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= it.nq_pad) return;
    double v = 0.0;
    if (i < it.d.n_model) v = assemble_bin(D, it, b, i) - it.q_x0[i];
    else if (i < it.nq) {
        const double* t = D.theta + (size_t)b * D.n_params;
        const double f = D.pk_direct ? 1.0 : 1.0 + t[it.d.bao_amp_slot];
        v = f * t[it.q_slot[i - it.d.n_model]] - it.q_x0[i];
    }
    it.q_x[(size_t)b * it.nq_pad + i] = v;
}

// FACTORED: the same chi2 as a sum of squares, || u0 - F dx ||^2 per item (the slabs hold F dx): what the quadratic form costs
// when nq >> n_masked (a model grid COEFMOD times finer than the data grid: 2 n_masked nq instead of nq^2 flops per walker)
template <bool FACTORED>
__global__ __launch_bounds__(CHI2_THREADS) void k_chi2_quad(EngineDev D, int B, SlabInfo slabs)
{
    __shared__ double red[CHI2_THREADS / 64];
    const int b = blockIdx.x;
    const int mock = D.mock_index[b];
    double acc = 0.0;
    for (int q = 0; q < D.n_items; ++q) {
        const ItemDev& it = D.items[q];
        const int row = (mock >= 0 && it.mock_pool) ? 1 + mock : 0;
        if constexpr (FACTORED) {
            const double* u0 = it.q_u0 + (size_t)row * it.n_masked_pad;
            const int ns = slabs.z[q];
            for (int i = threadIdx.x; i < it.n_masked; i += CHI2_THREADS) {
                double y = 0.0;
                for (int s = 0; s < ns; ++s) y += it.q_y[((size_t)s * B + b) * it.n_masked_pad + i];       // fixed order
                const double d = u0[i] - y;
                acc = fma(d, d, acc);
            }
            continue;
        }
        const double* g = it.q_lin + (size_t)row * it.nq_pad;
        const int ns = slabs.z[q];
        for (int i = threadIdx.x; i < it.nq; i += CHI2_THREADS) {
            const double x = it.q_x[(size_t)b * it.nq_pad + i];
            double zs[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) zs[s] = it.q_z[((size_t)(s < ns ? s : 0) * B + b) * it.nq_pad + i];
            double z = zs[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) z += s < ns ? zs[s] : 0.0;
            acc = fma(x, 2.0 * (z - g[i]), acc);        // q_z holds L' dx of the half form: dx^T Q' dx = 2 dx^T (L' dx)
        }
        if (threadIdx.x == 0) acc += it.q_c0[row];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0;
        for (int w = 0; w < CHI2_THREADS / 64; ++w) c += red[w];
        const double* t = D.theta + (size_t)b * D.n_params;
        for (int q = 0; q < D.n_priors; ++q) {
            const double dlt = t[D.prior_slot[q]] - D.prior_mean[q];
            c += dlt * dlt / (D.prior_sigma[q] * D.prior_sigma[q]);
        }
        int st = D.status[b];
        if (!(c == c) || c > 1e300 || c < -1e300) { st |= VMX_STATUS_NONFINITE; D.status[b] = st; }
        D.chi2[b] = st ? 1e100 : c;
        if (D.chi2_host) D.chi2_host[b] = st ? 1e100 : c;
        if (D.status_host) D.status_host[b] = st;
        if (D.done_host) { __threadfence_system(); *D.done_host = D.done_seq; }      // (set for single-walker calls only)
        if (b == 0) {
            D.k_live[2] = D.coef_win[0]; D.k_live[3] = D.coef_win[1]; D.coef_win[0] = 0x7fffffff; D.coef_win[1] = -1;
            xtab_key_store(D);
        }
    }
}

// chi2 from the contraction partials of the quadratic-form launch (GemmArgs::part): the slots of a persistent launch are
// numbered per walker tile in tape order, [nt_off[nt], nt_off[nt + 1]); walker b of tile nt adds the two wave-row sums of
// every slot of its tile in that order - an order that does not depend on which block computed what.  One wave per walker.
__global__ __launch_bounds__(256) void k_chi2_parts(EngineDev D, int B, const double* part, const int32_t* nt_off, int with_c0)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int nt = b >> 6, nl = b & 63;
    // Everything the walker's last additions need is requested FIRST, by every lane (uniform addresses): behind the reduction
    // each of these was a dependent round trip of lane 0 - eight in a row, most of this kernel's time.
    constexpr int CHI2_PRE = 4;     // items whose constant is requested ahead
    const int s0 = nt_off[nt], s1 = nt_off[nt + 1];
    // (with_c0 = 0: the contraction was r^T C^-1 r of the full chain itself - no constants of a quadratic form to add)
    const int n_c0 = with_c0 ? D.n_items : 0;
    const int mock = with_c0 ? D.mock_index[b] : -1;
    int st = D.status[b];
    double c0[CHI2_PRE];
#pragma unroll
    for (int q = 0; q < CHI2_PRE; ++q) {
        c0[q] = 0.0;
        if (q < n_c0) { const ItemDev& it = D.items[q]; c0[q] = it.q_c0[(mock >= 0 && it.mock_pool) ? 1 + mock : 0]; }
    }
    double acc = 0.0;
    for (int j = s0 + lane; j < s1; j += 64) {
        const double* pp = part + ((size_t)j * 64 + nl) * 2;
        acc += pp[0] + pp[1];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane != 0) return;
    double c = acc;
#pragma unroll
    for (int q = 0; q < CHI2_PRE; ++q) if (q < n_c0) c += c0[q];
    for (int q = CHI2_PRE; q < n_c0; ++q) {
        const ItemDev& it = D.items[q];
        c += it.q_c0[(mock >= 0 && it.mock_pool) ? 1 + mock : 0];
    }
    const double* t = D.theta + (size_t)b * D.n_params;
    for (int q = 0; q < D.n_priors; ++q) {
        const double dlt = t[D.prior_slot[q]] - D.prior_mean[q];
        c += dlt * dlt / (D.prior_sigma[q] * D.prior_sigma[q]);
    }
    if (!(c == c) || c > 1e300 || c < -1e300) { st |= VMX_STATUS_NONFINITE; D.status[b] = st; }
    D.chi2[b] = st ? 1e100 : c;
    if (D.chi2_host) D.chi2_host[b] = st ? 1e100 : c;
    if (D.status_host) D.status_host[b] = st;
#ifdef VMX_EXP_SKIP_SMALL
    if (b == 0) { D.k_live[2] = D.coef_win[0]; D.k_live[3] = D.coef_win[1]; xtab_key_store(D); return; }      // (experiment: the window is never reset)
#endif
    if (b == 0) {
        D.k_live[2] = D.coef_win[0]; D.k_live[3] = D.coef_win[1]; D.coef_win[0] = 0x7fffffff; D.coef_win[1] = -1;
        xtab_key_store(D);
    }
}

// set-up kernels of the quadratic form ----------------------------------------------------------
// X[j][i] = DM'[mask_idx[i]][j]: the masked rows of [DM | post-add broadband basis], transposed (row j = column j of DM')
__global__ void k_quad_gather(double* X, int ldx, const double* dm, int dm_ld, const int32_t* mask_idx, int n_masked,
                              int n_model, int nq, const double* bb_basis, const int64_t* basis_off, int n_dist, int csr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= n_masked || j >= nq) return;
    const int r = mask_idx[i];
    double v;
    if (j < n_model) {
        if (csr) return;            // (k_quad_gather_csr scatters the non-zeros of a CSR matrix into the zeroed X)
        v = dm ? dm[(size_t)r * dm_ld + j] : (r == j ? 1.0 : 0.0);
    } else v = bb_basis[basis_off[j - n_model] + r];
    X[(size_t)j * ldx + i] = v;
}

// full symmetric matrix from the half form (L below the diagonal, half the diagonal on it)
__global__ void k_sym_from_half(double* full, const double* half, int n, int ld)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (i >= n || j >= n) return;
    full[(size_t)i * ld + j] = i > j ? half[(size_t)i * ld + j] : i < j ? half[(size_t)j * ld + i] : 2.0 * half[(size_t)i * ld + i];
}

__global__ void k_half_from_full(double* half, const double* full, int n, int ld)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (i >= n || j >= ld) return;
    double v = 0.0;
    if (j < i) v = 0.5 * (full[(size_t)i * ld + j] + full[(size_t)j * ld + i]);
    else if (j == i) v = 0.5 * full[(size_t)i * ld + i];
    half[(size_t)i * ld + j] = v;
}

// residuals of the reference point: R0[0] = data - S m0, R0[1 + k] = mock k - S m0   (m0: its model on the distorted grid)
__global__ void k_quad_rows(double* R0, int ld, const double* model0, const int32_t* mask_idx, const double* data,
                            const double* pool, int n_masked, int n_rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i >= n_masked || k >= n_rows) return;
    const double d = k == 0 ? data[i] : pool[(size_t)(k - 1) * n_masked + i];
    R0[(size_t)k * ld + i] = d - model0[mask_idx[i]];
}

// masked bins of the reference model (kept for the linear terms of mocks that arrive later)
__global__ void k_mask_gather(double* out, const double* model0, const int32_t* mask_idx, int n_masked)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_masked) out[i] = model0[mask_idx[i]];
}

// Monte-Carlo mocks made on the device: pool row = fiducial + noise (noise = cholesky(C) . normal draws, a product of the chain's
// kernels; reference vega/data.py:751-753), and the residual of the quadratic form's reference point for the same rows
__global__ void k_mock_rows(double* pool, double* R0, int ld, const double* noise, const double* fid, const double* m0, int n_masked, int n_rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (i >= n_masked || k >= n_rows) return;
    const double v = fid[i] + noise[(size_t)k * ld + i];
    pool[(size_t)k * n_masked + i] = v;
    if (R0) R0[(size_t)k * ld + i] = v - m0[i];
}

// out[k] = sum_i a[k][i] b[k][i]   (one block per row, fixed-order reduction)
__global__ __launch_bounds__(256) void k_rowdot(double* out, const double* a, const double* b, int ld, int n)
{
    __shared__ double red[4];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc = fma(a[(size_t)k * ld + i], b[(size_t)k * ld + i], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// D^T[n][m] = sum_k A[m][k] * X[n][k]      (A static matrix, X / D one vector per walker)
//   fp64 MFMA 16x16x4; block tile 64 x 64 x 16, 4 waves each owning a 32 x 32 quadrant.
//   split-K partial sums go to separate slabs of D
//   (summed in fixed order by the consumer: results are bitwise reproducible).
// ------------------------------------------------------------------------------------------------

#define GEMM_BM 64
#define GEMM_BN 64
#define GEMM_BK 32

struct GemmArgs {
    const double* A; int lda; int64_t a_batch;
    const double* X; int ldx; int64_t x_batch;
    double* D; int ldd; int64_t d_batch; int64_t d_slab;
    int M, N, K;            // K already padded to a multiple of 32 (zero padded operands)
    int nsplit, klen;       // klen multiple of the K step; nsplit in {1, 2, 4, 8} for the MFMA kernel
    int tm, tn;             // block tiles along matrix rows / walkers (MFMA kernel)
    const int32_t* k_limit; // optional device scalar: operand columns >= *k_limit are zero and skipped (MFMA kernel)
    const int32_t* m_window; // optional device int[2]: only the rows [lo, hi] of the result are needed (MFMA and streaming kernels)
    int tri;                // A is lower triangular (zeros above the diagonal): row tile mt needs k < (mt + 1) BM only
    // Quadratic-form launches (k_gemm_nt44<VMX_TAG_QUAD>, list mode) with `part` set do not store the product: the block
    // contracts its tile with the walker vectors it multiplied - sum_m X[n][m] 2 (D[n][m] - lin[row(n)][m]), the linear
    // term entering with the tile's first K segment - and leaves one partial sum per walker and wave row in
    // part[(block * 64 + walker in tile) * 2 + wave row]; k_chi2_parts adds them in block order.
    double* part; const double* lin; const int32_t* lin_row; int lin_pool;
    // ... and count their row tiles from the bottom of the triangle (vmx_plan.h: tape_row0): tile 0 = rows [0, row0), tile
    // t >= 1 = rows [row0 + 64 (t - 1), row0 + 64 t); 0: the plain tiling
    int row0;
};
#define VMX_TAG_QUAD 12
#define VMX_TAG_FFTLOG 2

// The persistent form of a grouped launch (GemmGroup::work / queue; the quadratic-form tape - GemmWork and its planner live
// in vmx_plan.h): block p walks the entries queue[p] .. queue[p + 1] - 1 one after the other (the first stage of the next
// entry is requested behind the epilogue of the current one), and the contraction partials of entry w go to slot
// work[w].slot - numbered per walker tile in a canonical order, so that the consumer (k_chi2_parts) adds them in an order
// that does not depend on which block computed what.

// LDS reads as explicit ds_read_b64 (2 LDS cycles per wave, banks (a/4) mod 64): left to the compiler, pairs of them
// are merged into ds_read2_b64, which costs twice the cycles and banks modulo 32.  The compiler does not count these
// reads in its own s_waitcnt bookkeeping, so lds_wait() drains the counter before the values are used (its waits for its
// own LDS operations stay correct: extra operations in flight only make them conservative).
__device__ __forceinline__ unsigned lds_offset(const double* p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double*)p;
}
template <int OFFSET>
__device__ __forceinline__ double lds_read_b64(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET));
    return v;
}
template <int STRIDE>
__device__ __forceinline__ void lds_read_fragments(double (&v)[4], unsigned addr)
{
    v[0] = lds_read_b64<0>(addr); v[1] = lds_read_b64<STRIDE>(addr); v[2] = lds_read_b64<2 * STRIDE>(addr);
    v[3] = lds_read_b64<3 * STRIDE>(addr);
}
template <int STRIDE>
__device__ __forceinline__ void lds_read_fragments(double (&v)[8], unsigned addr)
{
    v[0] = lds_read_b64<0>(addr); v[1] = lds_read_b64<STRIDE>(addr); v[2] = lds_read_b64<2 * STRIDE>(addr);
    v[3] = lds_read_b64<3 * STRIDE>(addr); v[4] = lds_read_b64<4 * STRIDE>(addr); v[5] = lds_read_b64<5 * STRIDE>(addr);
    v[6] = lds_read_b64<6 * STRIDE>(addr); v[7] = lds_read_b64<7 * STRIDE>(addr);
}
__device__ __forceinline__ void lds_wait(double (&a)[4], double (&b)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}
__device__ __forceinline__ void lds_wait(double (&a)[8], double (&b)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void lds_wait(double (&a)[8], double (&b)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}

// Several independent products in one launch (the distortion products of all correlation items, then their C^-1
// products): every tile of every problem is in flight at once, so the small problems fill the tail of the large one.
struct GemmGroup {
    GemmArgs p[VMX_MAX_GROUP]; int32_t n; int32_t seq_end[VMX_MAX_GROUP];
    const GemmWork* work;       // non-null (with queue): the persistent tape of the quadratic form - block p walks the entries
    const int32_t* queue;       //   work[queue[p]] .. work[queue[p + 1] - 1] one after the other
    unsigned long long* trace;  // debugging aid (VMX_GEMM_TRACE): per block {start, first stage landed, K loop done, end} in 100 MHz ticks
    int32_t batch_in_x;         // > 0 (the FFTLog product's windowed launch): a one-dimensional grid, block x = ((seq * batches + batch) << 3) | xcd
                                // - the live blocks of EVERY batch member come first in launch order (with the batch on grid.y the live
                                // blocks of the last multipole sat behind three multipoles' dead blocks and started 8 us late)
};

// Block tile BM (matrix rows) x BN (walkers), K step BK; 4 waves in a 2 x 2 arrangement, each owning a
// (BM/2) x (BN/2) sub-tile as (BN/32) x (BM/32) MFMA tiles.  LDS rows are padded to BK + 2 doubles, which
// makes the 16-row x 4-k operand read pattern of v_mfma_f64_16x16x4 conflict-free for ds_read_b64.
// TAG = kernel class of the caller: one instantiation (and one name in a profiler's kernel table) per product of the chain.
template <int BM, int BN, int BK, int TAG>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmGroup G)
{
    constexpr int LD = BK + 2;
    constexpr int TI = BN / 32;        // MFMA tiles per wave along walkers
    constexpr int TJ = BM / 32;        // MFMA tiles per wave along matrix rows
    constexpr int RPP = 512 / BK;      // rows staged per pass (256 threads x 2 doubles)
    constexpr int PA = BM / RPP, PX = BN / RPP;
    __shared__ double sA[2][BM * LD];
    __shared__ double sX[2][BN * LD];

    // Workgroups are handed to the 8 XCDs round-robin on the linear block index, and every XCD has its own L2.
    // XCD x works on K split (x % nsplit) only, so the walker operand it touches (N x klen doubles) stays in its
    // L2, and it visits the walker tiles of one matrix-row tile back to back, so the matrix tile is fetched from
    // HBM once and found in L2 by the other walker tiles.  gridDim.x = 8 * blocks per XCD.
    const int xcd = blockIdx.x & 7;
    int seq = blockIdx.x >> 3;
    int pi = 0;
    while (pi < G.n - 1 && seq >= G.seq_end[pi]) ++pi;          // which problem this block belongs to (block-uniform)
    if (pi > 0) seq -= G.seq_end[pi - 1];
    const GemmArgs& g = G.p[pi];
    const int split = xcd % g.nsplit, group = xcd / g.nsplit, ngroups = 8 / g.nsplit;
    // triangular A: a block takes row tile p and its mirror tm - 1 - p, whose K ranges add up to the same length for
    // every p, so all blocks carry equal work
    const int tm_eff = g.tri ? (g.tm + 1) / 2 : g.tm;
    const int mt0 = (seq / g.tn) * ngroups + group, nt = seq % g.tn;
    if (mt0 >= tm_eff) return;
    const int npass = (g.tri && g.tm - 1 - mt0 != mt0) ? 2 : 1;
    const int batch = blockIdx.y;
    const double* A = g.A + batch * g.a_batch;
    const double* X = g.X + batch * g.x_batch;
    double* Dp = g.D + batch * g.d_batch + split * g.d_slab;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * (BM / 2), wn = (wave >> 1) * (BN / 2);
    const int lrow = tid / (BK / 2);
    const int lk = (tid % (BK / 2)) * 2;
    const int frow = lane & 15, fk = lane >> 4;
    const int n0 = nt * BN;

  for (int pass = 0; pass < npass; ++pass) {
    const int mt = pass == 0 ? mt0 : g.tm - 1 - mt0;
    const int m0 = mt * BM;
    int kbeg, kend;
    if (g.tri) {
        int kmax = ((mt + 1) * BM + BK - 1) / BK * BK; if (kmax > g.K) kmax = g.K;
        const int klen = ((kmax + g.nsplit - 1) / g.nsplit + BK - 1) / BK * BK;
        kbeg = split * klen; kend = kbeg + klen; if (kend > kmax) kend = kmax;
    } else {
        kbeg = split * g.klen; kend = kbeg + g.klen; if (kend > g.K) kend = g.K;
    }
    if (g.k_limit) { const int kl = (*g.k_limit + BK - 1) / BK * BK; if (kend > kl) kend = kl; }
    if (g.m_window && (m0 + BM <= g.m_window[0] || m0 > g.m_window[1])) continue;      // block-uniform: no barrier is skipped
    if (pass > 0) __syncthreads();          // the LDS buffers of the first tile are free again

    const double* pa[PA];
    const double* px[PX];
#pragma unroll
    for (int p = 0; p < PA; ++p) { int r = m0 + lrow + p * RPP; if (r >= g.M) r = g.M - 1; pa[p] = A + (size_t)r * g.lda + lk; }
#pragma unroll
    for (int p = 0; p < PX; ++p) { int r = n0 + lrow + p * RPP; if (r >= g.N) r = g.N - 1; px[p] = X + (size_t)r * g.ldx + lk; }

    v4d acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    v2d ra[PA], rx[PX];
    if (kbeg < kend) {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = *(const v2d*)(pa[p] + kbeg);
#pragma unroll
        for (int p = 0; p < PX; ++p) rx[p] = *(const v2d*)(px[p] + kbeg);
    }
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int p = 0; p < PA; ++p) *(v2d*)&sA[buf][(lrow + p * RPP) * LD + lk] = ra[p];
#pragma unroll
        for (int p = 0; p < PX; ++p) *(v2d*)&sX[buf][(lrow + p * RPP) * LD + lk] = rx[p];
        __syncthreads();
        const int kn = k0 + BK;
        if (kn < kend) {
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p] = *(const v2d*)(pa[p] + kn);
#pragma unroll
            for (int p = 0; p < PX; ++p) rx[p] = *(const v2d*)(px[p] + kn);
        }
        // explicit ds_read_b64 (see lds_read_b64): compiler-merged ds_read2_b64 pairs bank modulo 32 and conflict on this
        // layout (40 % of the LDS cycles); the fragments of the next K step are requested before this step's MFMAs
        const unsigned a = lds_offset(&sA[buf][(wm + frow) * LD + fk]);
        const unsigned x = lds_offset(&sX[buf][(wn + frow) * LD + fk]);
        static_assert(TI == 2 && TJ == 2, "fragment registers below are written for 2 x 2 MFMA tiles per wave");
        double av[2][TJ], xv[2][TI];
        av[0][0] = lds_read_b64<0>(a); av[0][1] = lds_read_b64<16 * LD * 8>(a);
        xv[0][0] = lds_read_b64<0>(x); xv[0][1] = lds_read_b64<16 * LD * 8>(x);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            const int cur = (ks / 4) & 1, nxt = cur ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[cur][0]), "+v"(av[cur][1]), "+v"(xv[cur][0]), "+v"(xv[cur][1]));
            if (ks + 4 < BK) {
                av[nxt][0] = lds_read_b64<0>(a + (ks + 4) * 8); av[nxt][1] = lds_read_b64<16 * LD * 8>(a + (ks + 4) * 8);
                xv[nxt][0] = lds_read_b64<0>(x + (ks + 4) * 8); xv[nxt][1] = lds_read_b64<16 * LD * 8>(x + (ks + 4) * 8);
            }
            // MFMA computes C[i][j] += sum_k Aop[i][k] Bop[k][j]; rows <- walkers (X), cols <- matrix rows (A):
            // the result tile is D^T[n][m], whose register layout (row = (lane>>4) + 4 r, col = lane & 15)
            // stores 16 consecutive m per 16 lanes -> coalesced along m.
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[cur][i], av[cur][j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }

#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn + 16 * i + (lane >> 4) + 4 * r;
                const int m = m0 + wm + 16 * j + (lane & 15);
                if (n < g.N && m < g.M) Dp[(size_t)n * g.ldd + m] = acc[i][j][r];
            }
  }
}

// v of lane (l ^ MASK), MASK < 32: ds_swizzle in bit mode - an LDS-crossbar instruction (no LDS memory, no VALU slot)
template <int MASK>
__device__ __forceinline__ double lane_xor(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    int lo = (int)bits, hi = (int)(bits >> 32);
    lo = __builtin_amdgcn_ds_swizzle(lo, (MASK << 10) | 0x1f);
    hi = __builtin_amdgcn_ds_swizzle(hi, (MASK << 10) | 0x1f);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// A DPP move of a double: two v_mov_b32 with a lane-select modifier (fp64 VALU instructions take none themselves).  The
// reductions below take these, not ds_swizzle: a swizzle answers after an LDS round trip and the compiler waits for each
// one before the add that needs it - a chain of 72 per walker row made the contraction epilogue 5.6 us long; the DPP moves
// are plain VALU instructions with a two-cycle hazard.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const long long bits = __builtin_bit_cast(long long, v);
    int lo = (int)bits, hi = (int)(bits >> 32);
    // (mov_dpp, not update_dpp(lo, lo, ...): every lane is written - full row and bank masks, rotations and permutations have
    // no invalid source - so there is no "old" value to keep, and without one the compiler needs no copy in front of each move:
    // 300 of the contraction epilogue's 1500 instructions were such copies)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// sum over the 16 lanes of a row, the total in every lane: row rotations by 8 and 4 (after the first step lanes l and
// l ^ 8 agree, so rotating by 4 adds the same value as l ^ 4 would), then the quad permutations [2,3,0,1] and [1,0,3,2]
// - bit for bit the butterfly v += v[l^8]; v += v[l^4]; v += v[l^2]; v += v[l^1]
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_move<0x128>(v);        // row_ror:8
    v += dpp_move<0x124>(v);        // row_ror:4
    v += dpp_move<0x4e>(v);         // quad_perm:[2,3,0,1]
    return v + dpp_move<0xb1>(v);   // quad_perm:[1,0,3,2]
}
// the four K-quarter partial sums of an accumulator sit in the lane groups b = (lane >> 2) & 3 of a 16-lane row: two
// steps leave their sum in every lane - (v_b + v_{b^2}) + (v_{b^1} + v_{b^3}), the same value in all four
__device__ __forceinline__ double sum_k_quarters(double v)
{
    v += dpp_move<0x128>(v);
    return v + dpp_move<0x124>(v);
}

#ifndef GEMM44_THREADS
#define GEMM44_THREADS 256
#endif
// The same product on the four-block fp64 MFMA (v_mfma_f64_4x4x4_4b_f64: 73-77 TFLOP/s issue rate against the 46-49 of
// the 16x16x4 form).  The instruction multiplies four independent 4 x 4 x 4 blocks; here the four blocks are four
// interleaved quarters of a 16-deep K step of ONE 4 x 4 output tile (operand lane 16 kk + 4 b + row holds k = 4 kk + b
// of block b), so a fragment is 4 rows x 16 k for both operands - one ds_read_b64 each, no rotated copies.  Each
// accumulator keeps the four partial sums of its tile in the four lane groups of a row; they are added once, after the K
// loop (two butterfly steps across the lane groups, sum_k_quarters).
//
// Staging is direct global -> LDS (global_load_lds_dwordx4: no staging registers, no ds_write): one wave instruction
// fills 4 rows x 32 doubles of a tile, lane-linear.  The 16-byte chunks of a row are stored XOR-swizzled with
// 4 (row & 3) - applied to the SOURCE address here and to the fragment read address there - so the four rows of a
// fragment read fall on disjoint quarters of the 64 banks without padding.
//
// Pipeline: one barrier per 32-deep K stage, placed in the MIDDLE of the stage's MFMA work.  The SIMD issues from its
// oldest wave first, so the second resident block does not fill this block's bubbles: every latency has to be covered
// by the wave's own MFMAs.  Stage s (LDS buffer s & 1; fragments f0 / f1 = first / second 16-deep K step):
//     read f1(s)                                   | MFMAs on f0(s)
//     drain LDS reads and the DMA of stage s + 1, barrier      (stage s + 1 visible, buffer s & 1 free)
//     DMA stage s + 2 -> buffer s & 1, read f0(s + 1)          | MFMAs on f1(s)
//
// NBUF = 2 (above) relies on a second resident block to cover the latency of its loads.  A launch with fewer tiles than
// twice the CUs (the FFTLog product: ~256 tiles of 18 stages) gets NBUF = 4: a ring of four stage buffers in 128 KB of
// dynamic LDS - one block per CU, loads three and a half stages ahead, counted waits (vmcnt) for the oldest stage only.
// BNT = 32 (four waves only): block tile 64 matrix rows x 32 walkers, wave tiles 32 x 16 - half the MFMA work per stage and
// 48 KB of LDS per block.  For launches with about one 64 x 64 tile per CU (the FFTLog product at B = 256: 256 live tiles of
// 18 stages) it doubles the blocks, so that every CU holds two and each fills the other's barrier and latency bubbles.
template <int TAG, int NBUF = 2, int BNT = 64>
__global__ __launch_bounds__(GEMM44_THREADS, NBUF == 2 ? GEMM44_THREADS / 128 : 1) void k_gemm_nt44(GemmGroup G)
{
    constexpr int BM = 64, BN = BNT, BK = 32;
    constexpr int NT = GEMM44_THREADS, NW = NT / 64;
    static_assert(BN == 64 || (BN == 32 && NT == 256 && NBUF == 2), "the 32-walker tile exists for the four-wave, two-buffer kernel");
    constexpr int FJ = 4 * 256 / NT * 2;    // A fragments per wave: 8 (4 waves, 32 x 32 wave tiles) or 4 (8 waves, 32 x 16)
    constexpr int NI = BN == 64 ? 8 : 4;    // X fragments per wave (4 walkers each)
    constexpr int AR = BN == 64 ? 1 : 2;    // A fragments read per X fragment in the interleaved read / MFMA steps
    constexpr int NP = 16 / NW;             // DMA instructions per wave and stage for the matrix operand (each fills 4 rows)
    constexpr int NPX = BN / 4 / NW;        // ... for the walker operand
    __shared__ double sA_static[NBUF == 2 ? 2 : 1][NBUF == 2 ? BM * BK : 1];
    __shared__ double sX_static[NBUF == 2 ? 2 : 1][NBUF == 2 ? BN * BK : 1];
    extern __shared__ double s_ring[];          // NBUF > 2: [NBUF][BM * BK] A stages, then [NBUF][BN * BK] X stages
    double (*sA)[BM * BK] = NBUF == 2 ? (double (*)[BM * BK])&sA_static[0][0] : (double (*)[BM * BK])s_ring;
    double (*sX)[BN * BK] = NBUF == 2 ? (double (*)[BN * BK])&sX_static[0][0] : (double (*)[BN * BK])(s_ring + NBUF * BM * BK);
    // waits until at most `ahead` younger stages' DMA instructions are outstanding (loads complete in order)
    auto wait_stages = [](int ahead) {
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * 2 * NP) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NP) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * 2 * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // contraction epilogue of the quadratic form (GemmArgs::part): the linear term's row over the tile's own rows; the
    // walker vectors over those rows follow by DMA into the stage buffer that falls free first (see the K loop)
    constexpr bool QUAD = TAG == VMX_TAG_QUAD && NT == 256 && NBUF == 2 && BN == 64;
    __shared__ double sL[QUAD ? 2 * BM : 1];

    // list mode = the persistent tape of the quadratic form (GemmGroup::work / queue): block p walks its queue's entries
    const bool persist = QUAD && G.work != nullptr && G.queue != nullptr;
    const bool list = persist;
    const unsigned long long t_start = G.trace ? wall_clock64() : 0ull;
    unsigned long long t_first = 0ull, t_loop = 0ull;
#ifdef VMX_EPI_TRACE
    unsigned long long t_e1 = 0ull, t_e2 = 0ull;
#endif
    GemmWork wk{};
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (TAG == VMX_TAG_FFTLOG) {
        if (G.batch_in_x > 0) { const int s2 = bx >> 3; by = s2 % G.batch_in_x; bx = ((s2 / G.batch_in_x) << 3) | (bx & 7); }
    }
    int w_first = bx, w_count = 1;
    if (persist) { w_first = G.queue[blockIdx.x]; w_count = G.queue[blockIdx.x + 1] - w_first; if (w_count <= 0) return; }
    if (list) { wk = G.work[w_first]; if (wk.prob < 0) return; }        // (padding entries of the list)
    const int xcd = bx & 7;
    int seq = bx >> 3;
    int pi = 0;
    if (list) pi = wk.prob;
    else {
        while (pi < G.n - 1 && seq >= G.seq_end[pi]) ++pi;
        if (pi > 0) seq -= G.seq_end[pi - 1];
    }
    const GemmArgs& g = G.p[pi];
    const int split = list ? wk.slab : xcd % g.nsplit, group = xcd / g.nsplit, ngroups = 8 / g.nsplit;
    const int tm_eff = g.tri ? (g.tm + 1) / 2 : g.tm;
    // A problem with a row window spreads its WALKER tiles over the XCDs (XCD x takes nt = x, x + 8, ...): the live row tiles
    // are a few consecutive ones, which the row-major order below would hand to as many XCDs and leave the others idle.
    const bool n_major = !list && g.m_window != nullptr && g.nsplit == 1 && !g.tri;
    const int tnx = (g.tn + 7) / 8;
    // (... and consecutive blocks take consecutive ROW tiles: the few live row tiles are then spread evenly over the launch
    // order, and with them over the CUs - with the walker tiles innermost the live blocks were one contiguous run of the grid
    // that the dispatcher packed onto a fraction of the CUs)
    // ... and the blocks are numbered over the LIVE row tiles only (the window is device data, the grid is sized for every row
    // tile): the first blocks of the launch are all live, the surplus ones come last in launch order and leave at once - with
    // the dead row tiles' blocks in between, each held a slot for the ~3 us of its window load and a fifth of the live blocks
    // started a round late.
    int tm_live = tm_eff;
    if (n_major) {
        const int w_lo = max(g.m_window[0], 0), w_hi = min(g.m_window[1], g.M - 1);
        tm_live = w_hi >= w_lo ? (w_hi - w_lo) / BM + 1 : 0;
        if (tm_live == 0) return;
    }
    const int mt0 = list ? wk.mt : n_major ? seq % tm_live : (seq / g.tn) * ngroups + group;
    const int nt = list ? wk.nt : n_major ? (seq / tm_live) * 8 + xcd : seq % g.tn;
    if (!list && (mt0 >= tm_eff || nt >= g.tn)) return;
    const int npass = persist ? w_count : (!list && g.tri && g.tm - 1 - mt0 != mt0) ? 2 : 1;
    const int batch = by;
    const char* A = (const char*)(g.A + batch * g.a_batch);
    const char* X = (const char*)(g.X + batch * g.x_batch);
    double* Dp = g.D + batch * g.d_batch + split * g.d_slab;
    // what the K loop and the contraction epilogue read of the current entry's problem: scalars, so that a persistent block
    // (whose entries may belong to different problems) reloads a dozen values per entry - a pointer into the kernel
    // arguments that moves would send the whole GemmGroup to scratch memory
    int p_M = g.M, p_N = g.N, p_lda = g.lda, p_ldx = g.ldx, p_lin_pool = g.lin_pool, p_row0 = g.row0;
    int p_mend = g.M;           // one past the last row of the current tile that counts (a ragged first tile: row0)
    double* p_part = g.part;
    const double* p_lin = g.lin;
    const int32_t* p_lin_row = g.lin_row;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NT == 256 ? (wave & 1) * 32 : (wave & 3) * 16, wn = NT == 256 ? (wave >> 1) * (BN / 2) : (wave >> 2) * 32;
    const int fq = lane & 3, fb = (lane >> 2) & 3, fkk = lane >> 4;
    // fragment read: row (4 f + fq), k = ks + 4 fkk + fb -> chunk (k / 2) ^ (4 fq), half k & 1
    const unsigned frag0 = (unsigned)(((2 * fkk + (fb >> 1)) ^ (4 * fq)) * 16 + (fb & 1) * 8);      // ks = 0; ks = 16 is ^ 128
    const unsigned fa = lds_offset(&sA[0][(wm + fq) * BK]), fx = lds_offset(&sX[0][(wn + fq) * BK]);
    // DMA: lane l of a wave instruction brings row (l >> 4) of its 4-row group, LDS chunk l & 15 <- source chunk (l & 15) ^ (4 (l >> 4))
    const int drow = lane >> 4;
    const unsigned dchunk = (unsigned)(((lane & 15) ^ (4 * drow)) * 16);
    int n0 = nt * BN;

    // per-pass state: tile origin, K range, DMA source offsets (32-bit byte offsets from the scalar operand bases)
    int m0 = 0, kbeg = 0, kend = 0;
    int slot = blockIdx.x;          // where the contraction partials of the current entry go (GemmArgs::part)
    bool skip = false;
    unsigned oa[NP], ox[NPX];
    // (a windowed problem in walker-major order tiles its rows from the window's first row: 240 wanted rows are four tiles
    // wherever they sit, not five)
    const int m_base = n_major ? max(g.m_window[0], 0) : 0;
    auto setup = [&](int pass) {
        if (persist) {
            // the block's next entry: its problem, walker tile, row tile, K segment and output slot
            wk = G.work[w_first + pass];
            const GemmArgs& q = G.p[__builtin_amdgcn_readfirstlane(wk.prob)];
            A = (const char*)q.A; X = (const char*)q.X;
            p_M = q.M; p_N = q.N; p_lda = q.lda; p_ldx = q.ldx; p_lin_pool = q.lin_pool; p_row0 = q.row0;
            p_part = q.part; p_lin = q.lin; p_lin_row = q.lin_row;
            n0 = wk.nt * BN;
            slot = wk.slot;
        }
        const int mt = persist ? wk.mt : pass == 0 ? mt0 : g.tm - 1 - mt0;
        m0 = m_base + mt * BM;
        p_mend = p_M;
        if (persist && p_row0 > 0) {            // tiles counted from the bottom of the triangle: the ragged tile is the first
            m0 = mt == 0 ? 0 : p_row0 + (mt - 1) * BM;
            if (mt == 0) p_mend = p_row0;
        }
        if (list) { kbeg = wk.kbeg; kend = wk.kend; }
        else if (g.tri) {
            int kmax = ((mt + 1) * BM + BK - 1) / BK * BK; if (kmax > g.K) kmax = g.K;
            const int klen = ((kmax + g.nsplit - 1) / g.nsplit + BK - 1) / BK * BK;
            kbeg = split * klen; kend = kbeg + klen; if (kend > kmax) kend = kmax;
        } else {
            kbeg = split * g.klen; kend = kbeg + g.klen; if (kend > g.K) kend = g.K;
        }
        if (g.k_limit) { const int kl = (*g.k_limit + BK - 1) / BK * BK; if (kend > kl) kend = kl; }
        skip = g.m_window && (m0 + BM <= g.m_window[0] || m0 > g.m_window[1]);      // block-uniform
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int r = m0 + (p * NW + wave) * 4 + drow; if (r >= p_M) r = p_M - 1;
            oa[p] = (unsigned)(r * p_lda) * 8u + dchunk;
        }
#pragma unroll
        for (int p = 0; p < NPX; ++p) {
            int r = n0 + (p * NW + wave) * 4 + drow; if (r >= p_N) r = p_N - 1;
            ox[p] = (unsigned)(r * p_ldx) * 8u + dchunk;
        }
    };
    auto dma_stage = [&](int k, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (oa[p] + (unsigned)k * 8u)),
                                             (__attribute__((address_space(3))) void*)&sA[buf][(p * NW + wave) * 4 * BK], 16, 0, 0);
            if (p < NPX)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (ox[p] + (unsigned)k * 8u)),
                                                 (__attribute__((address_space(3))) void*)&sX[buf][(p * NW + wave) * 4 * BK], 16, 0, 0);
        }
    };
    // the first stage of a pass is requested before the previous pass stores its results (a triangular problem has two
    // passes per block: the second one's pipeline fills behind the first one's epilogue)
    setup(0);
    int first_buf = 0;              // stage buffer of the pass's first K stage
    if (!skip && kbeg < kend) dma_stage(kbeg, first_buf);

  for (int pass = 0; pass < npass; ++pass) {
    // (setup(pass + 1) moves the entry's values on before this pass's epilogue: the epilogue reads these copies)
    const int c_m0 = m0, kbeg_c = kbeg, kend_c = kend, c_n0 = n0, c_slot = slot, c_first = first_buf;
    const int c_M = p_mend, c_N = p_N, c_ldx = p_ldx, c_lin_pool = p_lin_pool;
    double* const c_part = p_part;
    const double* const c_lin = p_lin;
    const int32_t* const c_lin_row = p_lin_row;
    const bool c_skip = skip;

    double acc[NI][FJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = 0.0;

    double a0[FJ], x0[NI], a1[FJ], x1[NI];
    if (!c_skip && kbeg_c < kend_c) {
        if constexpr (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();            // (every wave has left the previous pass's epilogue: sL and its E buffer are free)
            if (kbeg_c + BK < kend_c) dma_stage(kbeg_c + BK, c_first ^ 1);
            if constexpr (QUAD) {
                // the linear term's row over this tile's rows (a tile's first K segment subtracts it in the epilogue)
                if (list && c_part && wave == 0 && kbeg_c == 0 && !c_lin_pool) {
                    const unsigned loff = (unsigned)(c_m0 + 2 * lane < c_ldx ? (c_m0 + 2 * lane) * 8 : 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)c_lin + loff),
                                                     (__attribute__((address_space(3))) void*)&sL[0], 16, 0, 0);
                }
            }
        } else {
            // the ring starts full: stages 1 .. NBUF - 1 follow stage 0 at once; then wait for stage 0 alone
            int ahead = 0;
#pragma unroll
            for (int q = 1; q < NBUF; ++q)
                if (kbeg_c + q * BK < kend_c) { dma_stage(kbeg_c + q * BK, q); ++ahead; }
            wait_stages(ahead);
            __syncthreads();
        }
        if (G.trace) t_first = wall_clock64();
        lds_read_fragments<4 * BK * 8>(a0, fa + (unsigned)c_first * (BM * BK * 8) + frag0);
        lds_read_fragments<4 * BK * 8>(x0, fx + (unsigned)c_first * (BN * BK * 8) + frag0);
        lds_wait(a0, x0);
    }
    int buf = NBUF == 2 ? c_first : 0;
    int epi_buf = -1;
    for (int k0 = kbeg_c; k0 < (c_skip ? kbeg_c : kend_c); k0 += BK) {
        const int nbuf = NBUF == 2 ? buf ^ 1 : (buf + 1 == NBUF ? 0 : buf + 1);        // the next stage's buffer
        // first half: MFMAs on f0(s), the reads of f1(s) spread between them (one read ahead of every group of MFMAs, so
        // the wave's MFMA stream is never held up by a burst of LDS instructions)
        const unsigned a1p = fa + (unsigned)buf * (BM * BK * 8) + (frag0 ^ 128u);
        const unsigned x1p = fx + (unsigned)buf * (BN * BK * 8) + (frag0 ^ 128u);
#define VMX_G44_FIRST(i)                                                                                              \
        if constexpr (i < NI) {                                                                                           \
            if constexpr (AR * i < FJ) a1[AR * i] = lds_read_b64<AR * i * 4 * BK * 8>(a1p);                               \
            if constexpr (AR == 2) a1[AR * i + 1] = lds_read_b64<(AR * i + 1) * 4 * BK * 8>(a1p);                        \
            x1[i] = lds_read_b64<i * 4 * BK * 8>(x1p);                                                                   \
            _Pragma("unroll") for (int j = 0; j < FJ; ++j)                                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x0[i], a0[j], acc[i][j], 0, 0, 0);                        \
            __builtin_amdgcn_sched_barrier(0);      /* keeps this order: the waits are placed by hand */                  \
        }
        VMX_G44_FIRST(0) VMX_G44_FIRST(1) VMX_G44_FIRST(2) VMX_G44_FIRST(3)
        VMX_G44_FIRST(4) VMX_G44_FIRST(5) VMX_G44_FIRST(6) VMX_G44_FIRST(7)
#undef VMX_G44_FIRST
        lds_wait(a1, x1);
        if constexpr (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA of stage s + 1 has landed
        else {
            // ... of stage s + 1, while up to NBUF - 2 younger ones stay in flight
            const int left = (kend_c - k0) / BK - 2;           // stages beyond s + 1 (K ranges are multiples of BK)
            wait_stages(left < 0 ? 0 : left > NBUF - 2 ? NBUF - 2 : left);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // second half: DMA of stage s + NBUF into the buffer just released, MFMAs on f1(s) with the reads of f0(s + 1) between them
        if (k0 + NBUF * BK < kend_c) dma_stage(k0 + NBUF * BK, buf);
        else if constexpr (QUAD) {
            if (epi_buf < 0 && list && c_part) {
                // no further stage: this buffer stays free.  It takes E[n][m] = X[n0 + n][m0 + m] for the contraction epilogue
                // - walkers 0..31 in the A half, 32..63 in the X half; a wave instruction brings two rows (lane l: row l >> 5,
                // chunk l & 31), chunks past the row's padded length are redirected to its start (never used)
                epi_buf = buf;
                const unsigned coff = (unsigned)(c_m0 + 2 * (lane & 31) < c_ldx ? (c_m0 + 2 * (lane & 31)) * 8 : 0);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int pr = p * NW + wave;
                    int r = c_n0 + 2 * pr + (lane >> 5); if (r >= c_N) r = c_N - 1;
                    double* dst = pr < 16 ? &sA[buf][2 * pr * BM] : &sX[buf][(2 * pr - 32) * BM];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + ((unsigned)(r * c_ldx) * 8u + coff)),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        }
        // (after the last stage these reads fetch stale data that nothing uses: one code path, no branch)
        const unsigned a0p = fa + (unsigned)nbuf * (BM * BK * 8) + frag0;
        const unsigned x0p = fx + (unsigned)nbuf * (BN * BK * 8) + frag0;
#define VMX_G44_SECOND(i)                                                                                             \
        if constexpr (i < NI) {                                                                                           \
            _Pragma("unroll") for (int j = 0; j < FJ; ++j)                                                               \
                acc[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1[i], a1[j], acc[i][j], 0, 0, 0);                        \
            if constexpr (AR * i < FJ) a0[AR * i] = lds_read_b64<AR * i * 4 * BK * 8>(a0p);                               \
            if constexpr (AR == 2) a0[AR * i + 1] = lds_read_b64<(AR * i + 1) * 4 * BK * 8>(a0p);                        \
            x0[i] = lds_read_b64<i * 4 * BK * 8>(x0p);                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }
        VMX_G44_SECOND(0) VMX_G44_SECOND(1) VMX_G44_SECOND(2) VMX_G44_SECOND(3)
        VMX_G44_SECOND(4) VMX_G44_SECOND(5) VMX_G44_SECOND(6) VMX_G44_SECOND(7)
#undef VMX_G44_SECOND
        lds_wait(a0, x0);
        buf = nbuf;
    }

    if (G.trace) t_loop = wall_clock64();
    if (pass + 1 < npass) {
        // every wave is past the last barrier of the K loop: both buffers are free - but for the one that took the E tile of
        // the contraction epilogue (the other one held the last K stage)
        first_buf = (NBUF == 2 && epi_buf >= 0) ? epi_buf ^ 1 : 0;
        setup(pass + 1);
        if (!skip && kbeg < kend) dma_stage(kbeg, first_buf);
    }
    if (c_skip) continue;
    if constexpr (TAG == VMX_TAG_QUAD && NT == 256) {
        if (list && c_part) {
            // contraction epilogue (see GemmArgs::part): lane (r, c) of group jg holds D[n][m] for n = n0 + wn + 4 i + r,
            // m = c_m0 + wm + 16 jg + c after the rotations below
            const int c = lane & 15, r = lane >> 4;
            const bool first_seg = kbeg_c == 0;
#ifdef VMX_EPI_TRACE      /* experiment build (scripts/gpu_epi_trace.py): the block trace then holds the LAST entry's epilogue phases */
            if (G.trace) t_first = wall_clock64();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                // E has landed for every wave
#ifdef VMX_EPI_TRACE
            if (G.trace) t_e1 = wall_clock64();
#endif
            if (epi_buf < 0) {              // an empty K range (never a tile's first segment): nothing was multiplied
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c == 0) c_part[((size_t)c_slot * 64 + wn + 4 * i + r) * 2 + (wave & 1)] = 0.0;
                continue;
            }
            const double* sE0 = &sA[epi_buf][0];
            const double* sE1 = &sX[epi_buf][0];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = c_n0 + wn + 4 * i + r;
                const int nc = n < c_N ? n : c_N - 1;
                const int e_n = wn + 4 * i + r;                             // walker within the tile (block-half uniform per wave)
                const double* e_row = (wn == 0 ? sE0 + e_n * BM : sE1 + (e_n - 32) * BM) + wm + c;     // + 16 jg: this lane's entries
                const double* lin = c_lin;
                if (first_seg && c_lin_pool) {          // per-walker data (mocks): the walker's own row of the linear term
                    const int mock = c_lin_row[nc];
                    lin += (size_t)(mock >= 0 ? 1 + mock : 0) * c_ldx;
                }
                double sum = 0.0;
#pragma unroll
                for (int jg = 0; jg < FJ / 4; ++jg) {
                    double tot[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double v = acc[i][4 * jg + u];
                        tot[u] = sum_k_quarters(v);
                    }
                    double out = fb == 0 ? tot[0] : fb == 1 ? tot[1] : fb == 2 ? tot[2] : tot[3];
                    const int m = c_m0 + wm + 16 * jg + c;
                    if (m < c_M) {
                        if (first_seg) out -= c_lin_pool ? lin[m] : sL[wm + 16 * jg + c];
                        sum = fma(e_row[16 * jg], 2.0 * out, sum);
                    }
                }
                // the 16 lanes of a row (same walker): total in every lane
                sum = row_sum16(sum);
                if (c == 0) c_part[((size_t)c_slot * 64 + wn + 4 * i + r) * 2 + (wave & 1)] = n < c_N ? sum : 0.0;
            }
#ifdef VMX_EPI_TRACE
            if (G.trace) t_e2 = wall_clock64();
#endif
            continue;
        }
    }
    // result lane 16 r + 4 b + c of acc[i][j]: partial sum b of D^T[n0 + wn + 4 i + r][m0 + wm + 4 j + c].  After the
    // rotations every lane group b holds the total; group b then stores column block j = 4 jg + b, so that a row of 16
    // lanes writes 16 consecutive m.
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jg = 0; jg < FJ / 4; ++jg) {
            double tot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double v = acc[i][4 * jg + u];
                tot[u] = sum_k_quarters(v);
            }
            const double out = fb == 0 ? tot[0] : fb == 1 ? tot[1] : fb == 2 ? tot[2] : tot[3];
            const int n = n0 + wn + 4 * i + (lane >> 4);
            const int m = c_m0 + wm + 16 * jg + (lane & 15);
            if (n < g.N && m < g.M) Dp[(size_t)n * g.ldd + m] = out;
        }
  }
    if (G.trace && threadIdx.x == 0 && !skip) {
        unsigned long long* tr = G.trace + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
#ifdef VMX_EPI_TRACE
        tr[0] = t_loop; tr[1] = t_first; tr[2] = t_e1; tr[3] = t_e2;      // K loop done, epilogue entered, E landed + barrier, reductions done
#else
        tr[0] = t_start; tr[1] = t_first; tr[2] = t_loop; tr[3] = wall_clock64();
#endif
    }
}

// Distortion product with a CSR matrix (the reference keeps it as scipy csr_array: data.py:342-346, model.py:143-144):
//   D[b][row] = sum_k val[k] X[b][idx[k]],  k in [ptr[row], ptr[row + 1])
// one wave per matrix row, NB walkers per pass (blockIdx.y = walker tile): the row's (index, value) pairs are read once
// per tile - 12 bytes per non-zero - and the gathers of X hit L2 (a walker's vector is ~20-40 KB).  HBM-bound for
// B <= 8: 12 nnz + 8 (rows + 1) + 8 B (n_in + n_out) bytes.
template <int NB>
__global__ __launch_bounds__(256) void k_csr_spmm(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                  const double* __restrict__ val, int rows, const double* __restrict__ X,
                                                  int ldx, int N, double* __restrict__ Dp, int ldd)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n0 = blockIdx.y * NB;
    const int64_t k0 = ptr[row], k1 = ptr[row + 1];
    double acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0;
    // two (index, value) pairs in flight per lane
    int64_t k = k0 + lane;
    for (; k + 64 < k1; k += 128) {
        const int c0 = idx[k], c1 = idx[k + 64];
        const double v0 = val[k], v1 = val[k + 64];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double* x = X + (size_t)(n0 + b < N ? n0 + b : N - 1) * ldx;
            acc[b] = fma(v1, x[c1], fma(v0, x[c0], acc[b]));
        }
    }
    if (k < k1) {
        const int c0 = idx[k];
        const double v0 = val[k];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = fma(v0, X[(size_t)(n0 + b < N ? n0 + b : N - 1) * ldx + c0], acc[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double v = acc[b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && n0 + b < N) Dp[(size_t)(n0 + b) * ldd + row] = v;
    }
}

// X[j][i] += DM[mask_idx[i]][j] for a CSR distortion matrix (the columns j < n_model of k_quad_gather's X; X starts zeroed)
__global__ void k_quad_gather_csr(double* X, int ldx, const int64_t* ptr, const int32_t* idx, const double* val,
                                  const int32_t* mask_idx, int n_masked)
{
    const int i = blockIdx.x;
    if (i >= n_masked) return;
    const int r = mask_idx[i];
    for (int64_t k = ptr[r] + threadIdx.x; k < ptr[r + 1]; k += blockDim.x) X[(size_t)idx[k] * ldx + i] = val[k];
}

// y[n][m] = sum over slabs (fixed order) - used by the stand-alone product entry point
__global__ void k_sum_slabs(const double* __restrict__ part, double* __restrict__ y, int64_t count, int nslab)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double v = part[i];
    for (int s = 1; s < nslab; ++s) v += part[(int64_t)s * count + i];
    y[i] = v;
}

// Small-batch product (NB <= 8 walkers): HBM-bound streaming of A, one wave per matrix row.
template <int NB>
__global__ __launch_bounds__(256) void k_gemv(GemmArgs g)
{
    const int batch = blockIdx.z;
    const double* A = g.A + batch * g.a_batch;
    const double* X = g.X + batch * g.x_batch;
    double* Dp = g.D + batch * g.d_batch;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= g.M) return;
    if (g.m_window && (row < g.m_window[0] || row > g.m_window[1])) return;
    const double* a = A + (size_t)row * g.lda;
    double acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0;
    // K is a multiple of 16; each lane consumes double2 chunks, 128 doubles per wave step, 4 steps in flight
    // (a lower-triangular A has nothing but zeros past its diagonal: the row ends there)
    const int K = g.tri ? min(g.K, (row + 2) & ~1) : g.K;
    int k = lane * 2;
    for (; k + 384 < K; k += 512) {
        const v2d a0 = *(const v2d*)(a + k), a1 = *(const v2d*)(a + k + 128);
        const v2d a2 = *(const v2d*)(a + k + 256), a3 = *(const v2d*)(a + k + 384);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double* x = X + (size_t)(b < g.N ? b : g.N - 1) * g.ldx + k;
            const v2d x0 = *(const v2d*)(x), x1 = *(const v2d*)(x + 128), x2 = *(const v2d*)(x + 256), x3 = *(const v2d*)(x + 384);
            acc[b] += a0.x * x0.x + a0.y * x0.y + a1.x * x1.x + a1.y * x1.y
                    + a2.x * x2.x + a2.y * x2.y + a3.x * x3.x + a3.y * x3.y;
        }
    }
    for (; k < K; k += 128) {
        const v2d a0 = *(const v2d*)(a + k);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const v2d x0 = *(const v2d*)(X + (size_t)(b < g.N ? b : g.N - 1) * g.ldx + k);
            acc[b] += a0.x * x0.x + a0.y * x0.y;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double v = acc[b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && b < g.N) Dp[(size_t)b * g.ldd + row] = v;
    }
}

// Single-walker product y = A x, the HBM-bound case (B = 1: one chi2 evaluation inside a minimiser).
//   Persistent blocks (2 per CU): x is staged once per block in LDS; a block owns rows blockIdx.x + t gridDim.x
//   and its four waves split every row into interleaved 1 KiB chunks, so each wave keeps CPW x 16 B per lane in
//   flight per row and the next row is requested before the current one is reduced.  Partial sums of the four
//   waves meet in LDS once at the end (fixed order: bitwise reproducible).
#define GEMV1_MAX_ROWS 24
//   FUSED (distortion step of a single walker): x is assembled on the fly while it is staged (k_assemble's work, redone
//   by every block - 10 bins per thread) and each finished row goes straight through k_post's epilogue, which
//   removes two launches from a ~100 us latency-bound chain.
// MODE 0: D = A x.  MODE 1 (FUSED): x assembled from the item's components while staging, the result goes through post_bin.
// MODE 2: the single-walker quadratic form - the block contracts its rows with x, sum_r x[r] 2 (z[r] - lin[r]), and leaves
// that ONE number in g.part[block] (mapped host memory: the host adds the blocks' numbers, constants and priors itself - no
// reduction kernel at the end of a latency-bound chain); block 0 also hands over the status word and does the chain's
// end-of-evaluation duties (item = 1: this is the last product of the chain).
template <int CPW, int MODE>
__global__ __launch_bounds__(256) void k_gemv1(GemmArgs g, EngineDev D, int item)
{
    constexpr bool FUSED = MODE == 1;
    extern __shared__ double sx[];                 // [K]
    __shared__ double part[GEMV1_MAX_ROWS][4];
    const int batch = blockIdx.z;
    const double* A = g.A + batch * g.a_batch;
    const double* X = g.X + batch * g.x_batch;
    double* Dp = g.D + batch * g.d_batch;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    int koff[CPW];
    bool kval[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) { koff[c] = (w + 4 * c) * 128 + lane * 2; kval[c] = koff[c] < g.K; if (!kval[c]) koff[c] = 0; }

    // Two row buffers used in turn (the row loop is unrolled by two): with fixed registers the compiler waits for the
    // buffer it is about to reduce only - vmcnt(CPW) - while the other row's loads stay in flight; a rotating
    // "current / next" pair made it wait for every outstanding load (vmcnt(0)), one row in flight per wave.
    // Loads are unconditional: a lane past the end of the row (or, for a lower-triangular A, past the diagonal, whose
    // bytes need not be read) re-reads the row's first 16 bytes and its product is masked when the row is reduced.
    v2d bufa[CPW], bufb[CPW];
    const int row0 = blockIdx.x, G = gridDim.x;
    const int n_rows = row0 < g.M ? (g.M - row0 + G - 1) / G : 0;        // rows of this block
    auto load_row = [&](v2d (&dst)[CPW], int r) {
        if (r > g.M - 1) r = g.M - 1;               // (a request past the last row repeats it; it is never reduced)
        const double* a = A + (size_t)r * g.lda;
#pragma unroll
        for (int c = 0; c < CPW; ++c)
            dst[c] = __builtin_nontemporal_load((const v2d*)(a + ((!g.tri || koff[c] <= r) ? koff[c] : 0)));
    };
    auto reduce_row = [&](const v2d (&src)[CPW], int r, int t) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const v2d x = *(const v2d*)&sx[koff[c]];
            const double term = src[c].x * x.x + src[c].y * x.y;
            acc += (kval[c] && (!g.tri || koff[c] <= r)) ? term : 0.0;
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) part[t][w] = acc;
    };
    // request the first row of the matrix stream before staging x, so both latencies overlap
    if (n_rows > 0) load_row(bufa, row0);
    if constexpr (FUSED) {
        const ItemDev& it = D.items[item];
        for (int k = tid; k < g.K; k += 256) sx[k] = k < it.d.n_model ? assemble_bin(D, it, 0, k) : 0.0;
    } else {
        for (int k = tid * 2; k < g.K; k += 512) *(v2d*)&sx[k] = *(const v2d*)(X + k);
    }
    __syncthreads();
    const int t = n_rows;
    for (int it = 0; it < n_rows / 2; ++it) {
        const int ra = row0 + 2 * it * G;
        load_row(bufb, ra + G);
        reduce_row(bufa, ra, 2 * it);
        load_row(bufa, ra + 2 * G);
        reduce_row(bufb, ra + G, 2 * it + 1);
    }
    if (n_rows & 1) reduce_row(bufa, row0 + (n_rows - 1) * G, n_rows - 1);
    __syncthreads();
    if constexpr (MODE == 2) {
        static_assert(GEMV1_MAX_ROWS <= 64, "the rows of a block are contracted by its first wave");
        if (tid < 64) {
            double pr = 0.0;
            if (tid < t) {
                const int r = blockIdx.x + tid * gridDim.x;
                const double v = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]);
                pr = sx[r] * (2.0 * (v - g.lin[r]));
            }
            for (int off = 32; off > 0; off >>= 1) pr += __shfl_down(pr, off, 64);
            if (tid == 0) {
                *(volatile double*)&g.part[blockIdx.x] = pr;
                if (blockIdx.x == 0 && item) {
                    *(volatile int32_t*)D.status_host = D.status[0];
                    D.k_live[2] = D.coef_win[0]; D.k_live[3] = D.coef_win[1]; D.coef_win[0] = 0x7fffffff; D.coef_win[1] = -1;
                    xtab_key_store(D);
                }
            }
        }
        return;
    }
    if (tid < t) {
        const int r = blockIdx.x + tid * gridDim.x;
        const double v = (part[tid][0] + part[tid][1]) + (part[tid][2] + part[tid][3]);
        if constexpr (FUSED) post_bin(D, D.items[item], 0, r, v);
        else Dp[r] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// xi on the (rescaled) bins of every pipeline
// ------------------------------------------------------------------------------------------------
__device__ inline double legendre_even(int e, double x)
{
    const double x2 = x * x;
    switch (e) {
        case 0: return 1.0;
        case 1: return 0.5 * (3.0 * x2 - 1.0);
        case 2: return 0.125 * ((35.0 * x2 - 30.0) * x2 + 3.0);
        default: return 0.0625 * (((231.0 * x2 - 315.0) * x2 + 105.0) * x2 - 5.0);
    }
}

// xi of bin `bin` of pipeline p for walker b of a batch of nB (everything k_xi_bins computes); oob: the rescaled
// separation left the spline's range
// Legendre sum of the cubic-spline multipoles at ln r' = x, mu' = rmu for pipeline P (pktoxi.py:144-162).
//   STATIC_BASIS = false: the walker's spline coefficients (column P.col of D.coef);
//   STATIC_BASIS = true : a0 C0 + a1 C1 + a2 C2 of the pipeline's static coefficient basis (D.poly_coef).
template <bool STATIC_BASIS>
__device__ __forceinline__ double spline_legendre(const EngineDev& D, const PipeDev& P, int b, int nB, double x, double rmu,
                                                  double a0, double a1, double a2, bool& oob)
{
    const vmx_pipe_desc& d = P.d;
    const int n_ell = d.n_ell, single_ell = d.single_ell;
    const size_t ncols = (size_t)nB * D.n_active;
    const size_t col = (size_t)(P.col >= 0 ? P.col : 0) * nB + b;       // pipeline-major: a pipeline's walkers are consecutive columns
    // knot index and taps of every multipole first (a multipole the pipeline does not have reads the taps of ell = 0
    // and is dropped), then the 16 coefficient loads in one go, then the arithmetic
    const double* cf[4];
    double tt[4];
    bool on[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ee = e < n_ell ? e : 0;
        const bool inside = D.extrapolate || !(x < D.x0[ee] || x > D.xlast[ee]);     // VegaBoundsError (pktoxi.py:149-152)
        on[e] = e < n_ell && inside;
        if (e < n_ell && !inside) oob = true;
        const double u = (x - D.x0[ee]) * D.inv_h[ee];
        int j = (int)floor(u);
        if (j < 0) j = 0;
        if (j > D.n_coef - 4) j = D.n_coef - 4;
        if (!(u == u)) j = 0;                   // (a NaN separation must not become an address)
        tt[e] = u - (double)j;
        cf[e] = STATIC_BASIS ? D.poly_coef + ((size_t)ee * D.n_static + P.poly_basis) * 3 * D.ncp + j
                             : D.coef + ((size_t)ee * ncols + col) * D.ncp + j;
    }
    double tap[4][4];
    if (STATIC_BASIS) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                tap[e][q] = fma(a2, cf[e][2 * D.ncp + q], fma(a1, cf[e][D.ncp + q], a0 * cf[e][q]));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) tap[e][q] = cf[e][q];
    }
    double xi = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double t = tt[e], t2 = t * t, t3 = t2 * t;
        const double omt = 1.0 - t;
        const double w0 = omt * omt * omt;
        const double w1 = 3.0 * t3 - 6.0 * t2 + 4.0;
        const double w2 = -3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0;
        const double s = (tap[e][0] * w0 + tap[e][1] * w1 + tap[e][2] * w2 + tap[e][3] * t3) * (1.0 / 6.0);
        if (on[e]) {
            if (single_ell >= 0) { if (e == single_ell) xi = s; }      // one multipole, no Legendre factor
            else xi += s * legendre_even(e, rmu);
        }
    }
    return xi;
}

// Kaiser coefficients a0 + a1 mu^2 + a2 mu^4 = (c0_1 + c1_1 mu^2)(c0_2 + c1_2 mu^2) of a static-basis pipeline, as
// k_pk_poly forms them from the walker's scalars
__device__ __forceinline__ void kaiser_coefficients(const vmx_pipe_desc& d, const double* sc, double& a0, double& a1, double& a2)
{
    double c01 = sc[S_BIAS1], c02 = sc[S_BIAS2];
    const double c11 = sc[S_BB1], c12 = d.same_tracer ? sc[S_BB1] : sc[S_BB2];
    if (d.same_tracer) c02 = c01;
    a0 = c01 * c02; a1 = fma(c01, c12, c11 * c02); a2 = c11 * c12;
}

// xi of bin `bin` of pipeline p for walker b of a batch of nB (everything k_xi_bins computes); oob: the rescaled
// separation left the spline's range.  MODE 0: per-walker spline coefficients; 1: static coefficient basis; 2: static
// coefficient basis on static coordinates - the spline and Legendre sums of the three basis vectors were evaluated per
// bin at set-up (k_poly_bins), three loads and three FMAs are left.
template <int MODE>
__device__ __forceinline__ double xi_bin_value(const EngineDev& D, int p, int b, int bin, int nB, bool& oob_out)
{
    const PipeDev& P = D.pipes[p];
    const vmx_pipe_desc& d = P.d;
    const double* sc = D.scal + ((size_t)b * D.n_pipe + p) * VMX_NS;
    const size_t c = P.coord_off + bin;
    // The kernel is a chain of dependent lookups (coordinates -> r' -> knot index -> coefficients -> evolution ...); left
    // in program order every one of them is a separate round trip to L2 and a wave spends its life waiting (~12 trips,
    // 0.079 ms per step).  Everything that does not depend on a computed address is therefore requested up front -
    // coordinates, per-bin factors, the walker's scalars, the descriptor fields - and the coefficient taps of all
    // multipoles are requested together (spline_legendre): three trips instead.
    const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c];
    const double lnrelz = D.clnrelz[c], growth = D.cgrowth[c];
    const double drp = sc[S_DRP], s_ap = sc[S_AP], s_at = sc[S_AT];
    const double ev1a = sc[S_EV1A], ev2a = sc[S_EV2A];
    const bool std_evol = d.tracer[0].evol_kind == VMX_EVOL_STD && d.tracer[1].evol_kind == VMX_EVOL_STD;
    const bool radiation = d.radiation && !d.is_peak, uv_shotnoise = d.uv_shotnoise != 0;
    const bool split_evol = P.split_evol, odd_terms = P.odd_rel || P.odd_asy;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (MODE != 0) kaiser_coefficients(d, sc, a0, a1, a2);

    // reference correlation_func.py:200-236: r' = sqrt(r_par'^2 + r_perp'^2), mu' = r_par' / r'.  The kernel is VALU-bound
    // (~600 instructions per bin, most of them in fp64 division, square root and logarithm sequences): mu' comes from one
    // reciprocal square root, ln r' = ln(r'^2) / 2, the knot coordinate from a multiplication by 1 / h.
    double rr2 = 0.0, rmu = 0.0;
    double xi = 0.0;
    bool oob = false;
    if (MODE == 2) {
        const double* y = D.poly_bins + P.poly_bins_off + bin;
        xi = fma(a2, y[2 * (size_t)P.n_pad], fma(a1, y[P.n_pad], a0 * y[0]));
    } else {
        if (r != 0.0) {
            const double rrp = s_ap * (rp0 + drp), rrt = s_at * rt0;
            rr2 = fma(rrp, rrp, rrt * rrt);
            if (rr2 != 0.0) rmu = rrp * vmx_rsqrt(rr2);
        }
        if (rr2 != 0.0) xi = spline_legendre<MODE == 1>(D, P, b, nB, 0.5 * vmx_log(rr2), rmu, a0, a1, a2, oob);
    }

    // bias evolution (correlation_func.py:276-370) and growth (:143)
    double ev;
    if (std_evol) {
        // relz^a1 * relz^a2 = exp((a1 + a2) ln relz), ln relz tabulated at upload
        ev = split_evol ? vmx_exp(fma(ev1a, lnrelz, ev2a * D.clnrelz2[c])) : vmx_exp((ev1a + ev2a) * lnrelz);
    } else {
        ev = 1.0;
        for (int q = 0; q < 2; ++q) {
            const double a = sc[q == 0 ? S_EV1A : S_EV2A], cc = sc[q == 0 ? S_EV1B : S_EV2B];
            if (d.tracer[q].evol_kind == VMX_EVOL_CROOM) {
                const double z1 = 1.0 + D.cz[c], ze = 1.0 + d.z_eff;
                ev *= (a + cc * z1 * z1) / (a + cc * ze * ze);
            } else ev *= pow(D.crelz[c], a);
        }
    }
    xi *= ev;
    xi *= growth;

    if (radiation) {
        // reference correlation_func.py:446-489: unrescaled coordinates shifted by delta_rp, or - radiation == 2,
        // `rescale-coords-systematics` - the rescaled ones, shifted by delta_rp once more (:470-472 as written)
        const bool resc = d.radiation == 2;
        const double rp = resc ? fma(s_ap, rp0 + drp, drp) : rp0 + drp;
        const double rtr = resc ? s_at * rt0 : rt0;
        const double rs2 = fma(rp, rp, rtr * rtr);
        const double irs = vmx_rsqrt(rs2);              // (r = 0 bins carry rs2 = drp^2 > 0 or are masked downstream)
        const double rs = rs2 * irs, ms = rp * irs;
        double xr = sc[S_RAD_S] * (irs * irs) * (1.0 - sc[S_RAD_A] * (1.0 - ms * ms));
        xr *= vmx_exp(-rs * fma(1.0 + ms, sc[S_RAD_IL], sc[S_RAD_ID]));
        xi += xr;
    }
    if (uv_shotnoise) {
        // reference correlation_func.py:649-686 on the unrescaled separation
        const double* t = D.theta + (size_t)b * D.n_params;
        const double amp = t[d.uvsn_slot[0]], lam = t[d.uvsn_slot[1]], bg = t[d.uvsn_slot[2]];
        // (uv_shotnoise == 2, `rescale-coords-systematics`: r = sqrt(r'^2 + mu'^2) as the reference writes it, :681-682)
        const double rsn = d.uv_shotnoise == 2 ? sqrt(rr2 + rmu * rmu) : r;
        const double tau = rsn / lam;
        double a;
        const double pos = (tau - D.sn_tau0) / D.sn_dtau;
        if (pos <= 0.0) a = D.sn_a[0];
        else if (pos >= (double)(D.sn_n - 1)) a = (pos == (double)(D.sn_n - 1)) ? D.sn_a[D.sn_n - 1] : 0.0;
        else { const int j = (int)pos; const double f = pos - (double)j; a = D.sn_a[j] + f * (D.sn_a[j + 1] - D.sn_a[j]); }
        xi += bg * bg * amp * lam / rsn * a;
    }
    if (odd_terms) {
        // reference pktoxi.py:321-382 on the rescaled coordinates (correlation_func.py:491-551)
        const double* t = D.theta + (size_t)b * D.n_params;
        const double rr = sqrt(rr2);
        const double x = log(rr);
        const double u = (x - P.odd_x0) / P.odd_h;
        int j = (int)floor(u);
        if (j < 0) j = 0;
        if (j > P.odd_ncoef - 4) j = P.odd_ncoef - 4;
        const double tt = u - (double)j, t2 = tt * tt, t3 = t2 * tt, omt = 1.0 - tt;
        const double w0 = omt * omt * omt, w1 = 3.0 * t3 - 6.0 * t2 + 4.0, w2 = -3.0 * t3 + 3.0 * t2 + 3.0 * tt + 1.0;
        double sp[4];
        for (int q = 0; q < 4; ++q) {
            const double* cf = (D.pk_direct && P.odd_dyn_off >= 0)
                                   ? D.odd_dyn + P.odd_dyn_off + (size_t)b * P.odd_dyn_ld + (size_t)q * P.odd_ncoef + j
                                   : D.odd_coef + P.odd_off + (size_t)q * P.odd_ncoef + j;
            sp[q] = (cf[0] * w0 + cf[1] * w1 + cf[2] * w2 + cf[3] * t3) * (1.0 / 6.0);
        }
        const double l1 = rmu, l3 = 0.5 * (5.0 * rmu * rmu - 3.0) * rmu;
        if (P.odd_rel) xi += t[P.odd_slot[0]] * sp[0] * l1 + t[P.odd_slot[1]] * sp[1] * l3;
        if (P.odd_asy) xi += (t[P.odd_slot[2]] * sp[2] - t[P.odd_slot[3]] * sp[3]) * rr * l1 + t[P.odd_slot[4]] * sp[3] * rr * l3;
    }
    oob_out = oob;
    return xi;
}

// pipes: the pipelines of this launch (STATIC_BASIS = true: those with a static coefficient basis, PipeDev::poly_basis;
// false: those with a column of per-walker spline coefficients) - separate instantiations (and kernels, below), because the
// 48 tap loads of the static form would otherwise cost every pipeline its occupancy
template <bool STATIC_BASIS>
__global__ __launch_bounds__(256) void k_xi_bins(EngineDev D, const int32_t* pipes)
{
    const int p = pipes[blockIdx.y], b = blockIdx.z;
    const PipeDev& P = D.pipes[p];
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= P.n) return;
    bool oob;
    double xi;
    xi = xi_bin_value<STATIC_BASIS ? 1 : 0>(D, p, b, bin, (int)gridDim.z, oob);
    if (oob) atomicOr(&D.status[b], VMX_STATUS_BOUNDS);
    D.xi[P.xi_off + (size_t)b * P.n_pad + bin] = xi;
}

// ... the static-basis pipelines on static coordinates (PipeDev::poly_bins_off >= 0: three loads and three FMAs per bin) ...
__global__ __launch_bounds__(256) void k_xi_bins_static(EngineDev D, const int32_t* pipes)
{
    const int p = pipes[blockIdx.y], b = blockIdx.z;
    const PipeDev& P = D.pipes[p];
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= P.n) return;
    bool oob;
    D.xi[P.xi_off + (size_t)b * P.n_pad + bin] = xi_bin_value<2>(D, p, b, bin, (int)gridDim.z, oob);
}

// Set-up: static-basis pipelines on static coordinates (no rescaling, no delta_rp): Y_i[bin] = Legendre sum of the splines of
// basis vector i at the bin's own (r, mu).  grid = (bins, static pipelines, 3); `ok` is cleared for a pipeline with a bin
// outside the spline range (it then stays on the tap form, which flags walkers as the reference raises).
__global__ __launch_bounds__(256) void k_poly_bins(EngineDev D, const int32_t* pipes, double* out, int32_t* ok)
{
    const int sb = blockIdx.y, p = pipes[sb], i = blockIdx.z;
    const PipeDev& P = D.pipes[p];
    const int bin = blockIdx.x * 256 + threadIdx.x;
    if (bin >= P.n || P.poly_bins_off < 0) return;
    const size_t c = P.coord_off + bin;
    const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c];
    double xi = 0.0;
    bool oob = false;
    if (r != 0.0) {
        const double rr2 = fma(rp0, rp0, rt0 * rt0);
        if (rr2 != 0.0)
            xi = spline_legendre<true>(D, P, 0, 1, 0.5 * log(rr2), rp0 * vmx_rsqrt(rr2), i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0,
                                       i == 2 ? 1.0 : 0.0, oob);
    }
    if (oob) ok[sb] = 0;
    out[P.poly_bins_off + (size_t)i * P.n_pad + bin] = xi;
}

// Items without metal terms, chi2-only small batches: the bins of the peak and the smooth component and the entry
// x' - x0' of the quadratic form in one kernel (one launch less in a latency-bound chain).  grid = (bins, walkers, items).
// store_xi = 0: the per-pipeline bins (the stage taps behind vmx_debug_read) are not written - two of the three stores
// per bin, 31 of the launch's 46 MB of writes at B = 256, which nothing reads in a chi2-only evaluation.
__global__ __launch_bounds__(256) void k_xi_assemble_quad(EngineDev D, int item0, int store_xi)
{
    const ItemDev& it = D.items[item0 + blockIdx.z];
    const int b = blockIdx.y, nB = gridDim.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= it.nq_pad) return;
    const double* t = D.theta + (size_t)b * D.n_params;
    double v = 0.0;
    if (i < it.d.n_model) {
        const PipeDev& Pp = D.pipes[it.d.pipe_peak];
        const PipeDev& Ps = D.pipes[it.d.pipe_smooth];
        bool oob_p, oob_s;
        double xs, xp;
        if (it.plain_pair) {
            // Both components are plain spline sums with the standard bias evolution (the item's flag, set by the host: no
            // radiation / shot-noise / odd-multipole term, no single multipole): everything that is not the spline itself -
            // the evolution exponential, the growth factor, the coordinates - is formed once for the two of them.
            const size_t c = Pp.coord_off + i;
            const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c];
            const double* scp = D.scal + ((size_t)b * D.n_pipe + it.d.pipe_peak) * VMX_NS;
            const double* scs = D.scal + ((size_t)b * D.n_pipe + it.d.pipe_smooth) * VMX_NS;
            const double ev = (Pp.split_evol ? vmx_exp(fma(scp[S_EV1A], D.clnrelz[c], scp[S_EV2A] * D.clnrelz2[c]))
                                             : vmx_exp((scp[S_EV1A] + scp[S_EV2A]) * D.clnrelz[c])) * D.cgrowth[c];
            xs = 0.0; xp = 0.0; oob_p = false; oob_s = false;
            if (r != 0.0) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const double* sc = half ? scp : scs;
                    const double rrp = sc[S_AP] * (rp0 + sc[S_DRP]), rrt = sc[S_AT] * rt0;
                    const double rr2 = fma(rrp, rrp, rrt * rrt);
                    if (rr2 != 0.0) {
                        const double v = spline_legendre<false>(D, half ? Pp : Ps, b, nB, 0.5 * vmx_log(rr2), rrp * vmx_rsqrt(rr2),
                                                                0.0, 0.0, 0.0, half ? oob_p : oob_s);
                        if (half) xp = v * ev; else xs = v * ev;
                    }
                }
            }
        } else {
            xs = xi_bin_value<0>(D, it.d.pipe_smooth, b, i, nB, oob_s);
            xp = xi_bin_value<0>(D, it.d.pipe_peak, b, i, nB, oob_p);
        }
        if (oob_p || oob_s) atomicOr(&D.status[b], VMX_STATUS_BOUNDS);
        if (store_xi) {                                         // (the stage taps stay valid)
            D.xi[Ps.xi_off + (size_t)b * Ps.n_pad + i] = xs;
            D.xi[Pp.xi_off + (size_t)b * Pp.n_pad + i] = xp;
        }
        // assemble_bin without metals (model.py:119-140,186)
        const double bao = t[it.d.bao_amp_slot];
        v = fma(bao, xp, xs);
        if (it.add_vec) v = fma(it.add_slot >= 0 ? t[it.add_slot] : it.add_default, it.add_vec[i], v);
        if (it.n_bb[VMX_BB_PRE_MUL]) v *= bb_total(D, it, VMX_BB_PRE_MUL, t, i, it.d.n_model);
        if (it.n_bb[VMX_BB_PRE_ADD]) v += (1.0 + bao) * bb_total(D, it, VMX_BB_PRE_ADD, t, i, it.d.n_model);
        v -= it.q_x0[i];
    } else if (i < it.nq) {
        v = (1.0 + t[it.d.bao_amp_slot]) * t[it.q_slot[i - it.d.n_model]] - it.q_x0[i];
    }
    it.q_x[(size_t)b * it.nq_pad + i] = v;
}

// The same for the common case, lean: every item a plain peak / smooth pair (ItemDev::plain_pair) without additive template or
// pre-distortion broadband, large chi2-only batch.  What the kernel needs of an item travels in its ARGUMENTS (no descriptor
// load ahead of the first coordinate load), and a thread evaluates its bin for NW walkers: the bin's coordinates, evolution
// logarithm, growth factor and reference entry are loaded once, and the walkers' independent chains (logarithm -> knot index
// -> 16 coefficient loads -> spline and Legendre sums) hide each other's latencies.  grid = (bins, ceil(B / NW), items).
struct XiPlainItem { int64_t coord_off; const double* q_x0; double* q_x; int32_t n_model, nq, nq_pad, pipe_s, pipe_p, col_s, col_p,
                     n_ell, split_evol, bao_slot, item, radiation; };     // radiation: of the smooth component (0, 1, 2 = rescaled coordinates)
struct XiPlainArgs { XiPlainItem it[VMX_MAX_GROUP]; };

// cubic B-spline multipoles of one (walker, pipeline) at ln r' = x, Legendre-summed at mu' = rmu (pktoxi.py:144-162)
// SAME: every multipole on one ln r grid (EngineDev::same_grid - always, unless fht_lowring moves the grids apart): knot index,
// offset and with them the four B-spline weights are formed once instead of once per multipole (the same values bit for bit).
template <bool SAME>
__device__ __forceinline__ double xi_plain_spline(const EngineDev& D, const double* coef_col, size_t ell_stride, int n_ell, double x,
                                                  double rmu, bool& oob)
{
    const double* cf[4];
    double tt[4];
    bool on[4];
    if (SAME) {
        const bool inside = !(x < D.x0[0] || x > D.xlast[0]);       // VegaBoundsError (pktoxi.py:149-152)
        if (!inside) oob = true;
        const double u = (x - D.x0[0]) * D.inv_h[0];
        int j = (int)floor(u);
        if (j < 0) j = 0;
        if (j > D.n_coef - 4) j = D.n_coef - 4;
        if (!(u == u)) j = 0;
        const double t = u - (double)j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            on[e] = e < n_ell && inside;
            tt[e] = t;
            cf[e] = coef_col + (size_t)(e < n_ell ? e : 0) * ell_stride + j;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ee = e < n_ell ? e : 0;
            const bool inside = !(x < D.x0[ee] || x > D.xlast[ee]);
            on[e] = e < n_ell && inside;
            if (e < n_ell && !inside) oob = true;
            const double u = (x - D.x0[ee]) * D.inv_h[ee];
            int j = (int)floor(u);
            if (j < 0) j = 0;
            if (j > D.n_coef - 4) j = D.n_coef - 4;
            if (!(u == u)) j = 0;
            tt[e] = u - (double)j;
            cf[e] = coef_col + (size_t)ee * ell_stride + j;
        }
    }
    double tap[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) tap[e][q] = cf[e][q];
    double xi = 0.0;
    const double x2 = rmu * rmu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double t = tt[e], t2 = t * t, t3 = t2 * t;
        const double omt = 1.0 - t;
        const double w0 = omt * omt * omt;
        const double w1 = 3.0 * t3 - 6.0 * t2 + 4.0;
        const double w2 = -3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0;
        const double sp = (tap[e][0] * w0 + tap[e][1] * w1 + tap[e][2] * w2 + tap[e][3] * t3) * (1.0 / 6.0);
        const double leg = e == 0 ? 1.0 : e == 1 ? 0.5 * (3.0 * x2 - 1.0) : e == 2 ? 0.125 * ((35.0 * x2 - 30.0) * x2 + 3.0)
                                        : 0.0625 * (((231.0 * x2 - 315.0) * x2 + 105.0) * x2 - 5.0);
        if (on[e]) xi += sp * leg;
    }
    return xi;
}

template <int NW, bool SAME>
__global__ __launch_bounds__(256) void k_xi_quad_plain(EngineDev D, XiPlainArgs A, int item0, int B)
{
    const XiPlainItem& I = A.it[item0 + blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= I.nq_pad) return;
    const int b0 = blockIdx.y * NW;
    if (i >= I.n_model) {
        // the additive post-distortion broadband coefficients (and the zero padding behind them)
        const ItemDev& it = D.items[I.item];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int b = b0 + w;
            if (b >= B) break;
            const double* t = D.theta + (size_t)b * D.n_params;
            double v = 0.0;
            if (i < I.nq) v = (1.0 + t[I.bao_slot]) * t[it.q_slot[i - I.n_model]] - I.q_x0[i];
            I.q_x[(size_t)b * I.nq_pad + i] = v;
        }
        return;
    }
    const size_t c = (size_t)I.coord_off + i;
    const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c], lnz = D.clnrelz[c], growth = D.cgrowth[c], x0v = I.q_x0[i];
    const double lnz2 = I.split_evol ? D.clnrelz2[c] : 0.0;
    const size_t ncols = (size_t)B * D.n_active, ell_stride = ncols * D.ncp;
    double out[NW];
    bool oob[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = min(b0 + w, B - 1);           // (a surplus walker slot shadows the last walker and stores nothing)
        const double* scs = D.scal + ((size_t)b * D.n_pipe + I.pipe_s) * VMX_NS;
        const double* scp = D.scal + ((size_t)b * D.n_pipe + I.pipe_p) * VMX_NS;
        const double ev = (I.split_evol ? vmx_exp(fma(scp[S_EV1A], lnz, scp[S_EV2A] * lnz2)) : vmx_exp((scp[S_EV1A] + scp[S_EV2A]) * lnz)) * growth;
        double xs = 0.0, xp = 0.0;
        oob[w] = false;
        if (r != 0.0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const double* sc = half ? scp : scs;
                const double rrp = sc[S_AP] * (rp0 + sc[S_DRP]), rrt = sc[S_AT] * rt0;
                const double rr2 = fma(rrp, rrp, rrt * rrt);
                if (rr2 != 0.0) {
                    const double* col = D.coef + ((size_t)(half ? I.col_p : I.col_s) * B + b) * D.ncp;
                    const double v = xi_plain_spline<SAME>(D, col, ell_stride, I.n_ell, 0.5 * vmx_log(rr2), rrp * vmx_rsqrt(rr2), oob[w]);
                    if (half) xp = v * ev; else xs = v * ev;
                }
            }
        }
        if (I.radiation) {
            // QSO radiation on the smooth component (correlation_func.py:446-489), as xi_bin_value forms it
            const bool resc = I.radiation == 2;
            const double drp = scs[S_DRP];
            const double rp = resc ? fma(scs[S_AP], rp0 + drp, drp) : rp0 + drp;
            const double rtr = resc ? scs[S_AT] * rt0 : rt0;
            const double rs2 = fma(rp, rp, rtr * rtr);
            const double irs = vmx_rsqrt(rs2);
            const double rs = rs2 * irs, ms = rp * irs;
            double xr = scs[S_RAD_S] * (irs * irs) * (1.0 - scs[S_RAD_A] * (1.0 - ms * ms));
            xr *= vmx_exp(-rs * fma(1.0 + ms, scs[S_RAD_IL], scs[S_RAD_ID]));
            xs += xr;
        }
        const double bao = D.theta[(size_t)b * D.n_params + I.bao_slot];
        out[w] = fma(bao, xp, xs) - x0v;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = b0 + w;
        if (b >= B) break;
        if (oob[w]) atomicOr(&D.status[b], VMX_STATUS_BOUNDS);
        I.q_x[(size_t)b * I.nq_pad + i] = out[w];
    }
}

// k_xi_bins for the pipelines that need nothing but the spline sum, the standard bias evolution, growth and (smooth component)
// the QSO radiation term - every pipeline of the joint + metals fit that forms its multipoles per walker: the pipeline's fields
// in the kernel ARGUMENTS, NW walkers per thread (coordinates, evolution logarithm and growth loaded once; NW independent
// spline chains in flight).  grid = (bins, pipelines of the list, ceil(B / NW)).
struct XiLeanPipe { int64_t coord_off, xi_off, poly_off; int32_t n, n_pad, pipe, col, n_ell, split_evol, radiation, same_tracer; };
#define VMX_XI_LEAN_MAX 24
struct XiLeanArgs { XiLeanPipe p[VMX_XI_LEAN_MAX]; };

template <int NW, bool SAME>
__global__ __launch_bounds__(256) void k_xi_bins_lean(EngineDev D, XiLeanArgs A, int B)
{
    const XiLeanPipe& P = A.p[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    const int b0 = blockIdx.z * NW;
    const size_t c = (size_t)P.coord_off + i;
    const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c], lnz = D.clnrelz[c], growth = D.cgrowth[c];
    const double lnz2 = P.split_evol ? D.clnrelz2[c] : 0.0;
    const size_t ell_stride = (size_t)B * D.n_active * D.ncp;
    double out[NW];
    bool oob[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = min(b0 + w, B - 1);
        const double* sc = D.scal + ((size_t)b * D.n_pipe + P.pipe) * VMX_NS;
        const double ev = (P.split_evol ? vmx_exp(fma(sc[S_EV1A], lnz, sc[S_EV2A] * lnz2)) : vmx_exp((sc[S_EV1A] + sc[S_EV2A]) * lnz)) * growth;
        double xi = 0.0;
        oob[w] = false;
        if (r != 0.0) {
            const double rrp = sc[S_AP] * (rp0 + sc[S_DRP]), rrt = sc[S_AT] * rt0;
            const double rr2 = fma(rrp, rrp, rrt * rrt);
            if (rr2 != 0.0) {
                const double* col = D.coef + ((size_t)P.col * B + b) * D.ncp;
                xi = xi_plain_spline<SAME>(D, col, ell_stride, P.n_ell, 0.5 * vmx_log(rr2), rrp * vmx_rsqrt(rr2), oob[w]) * ev;
            }
        }
        if (P.radiation) {
            const bool resc = P.radiation == 2;
            const double drp = sc[S_DRP];
            const double rp = resc ? fma(sc[S_AP], rp0 + drp, drp) : rp0 + drp;
            const double rtr = resc ? sc[S_AT] * rt0 : rt0;
            const double rs2 = fma(rp, rp, rtr * rtr);
            const double irs = vmx_rsqrt(rs2);
            const double rs = rs2 * irs, ms = rp * irs;
            double xr = sc[S_RAD_S] * (irs * irs) * (1.0 - sc[S_RAD_A] * (1.0 - ms * ms));
            xr *= vmx_exp(-rs * fma(1.0 + ms, sc[S_RAD_IL], sc[S_RAD_ID]));
            xi += xr;
        }
        out[w] = xi;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = b0 + w;
        if (b >= B) break;
        if (oob[w]) atomicOr(&D.status[b], VMX_STATUS_BOUNDS);
        D.xi[P.xi_off + (size_t)b * P.n_pad + i] = out[w];
    }
}

// The lean bins kernel over GROUPS of pipelines of one item (descriptors in device memory, one array per group): a thread adds
// its bin's value of every member, times the member's walker factor, and stores the sum - the per-pipeline arrays of the
// members are never written nor read back by k_assemble_quad (joint + metals: one array per group instead of one of the 42 arrays of 15 MB per pipeline and step).
// presum = 0: a single member whose plain bins are wanted (its array has another reader).  grid = (bins, groups, ceil(B / NW)).
#define VMX_XI_GROUP_MEMBERS 4
#define VMX_XI_SGROUP_MEMBERS 16
struct XiMember { int64_t coord_off, poly_off; int32_t pipe, col, n_ell, split_evol, radiation, same_tracer, fkind, findex; };
struct XiLeanGroup { int64_t out_off; int32_t n, n_pad, n_members, presum; XiMember m[VMX_XI_GROUP_MEMBERS]; };
struct XiStaticGroup { int64_t out_off; int32_t n, n_pad, n_members, presum; XiMember m[VMX_XI_SGROUP_MEMBERS]; };

// the factor a member's bins enter the item's vector with: 1 (smooth component), bao_amp (peak component), the metal pair's
// bias product times its multiplicity (k_prologue's metal_bias)
__device__ __forceinline__ double xi_member_factor(const EngineDev& D, const XiMember& P, int b)
{
    if (P.fkind == 1) return D.theta[(size_t)b * D.n_params + P.findex];
    if (P.fkind == 2) return D.metal_bias[(size_t)b * 3 * D.n_metals_total + P.findex];
    return 1.0;
}

template <int NW, bool SAME>
__global__ __launch_bounds__(256) void k_xi_bins_group(EngineDev D, const XiLeanGroup* groups, int B)
{
    const XiLeanGroup& G = groups[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G.n) return;
    const int b0 = blockIdx.z * NW;
    const size_t ell_stride = (size_t)B * D.n_active * D.ncp;
    double out[NW];
    bool oob[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { out[w] = 0.0; oob[w] = false; }
    // (unrolled: the members' chains of dependent lookups are independent of each other and run interleaved)
#pragma unroll
    for (int m = 0; m < VMX_XI_GROUP_MEMBERS; ++m) {
        if (m >= G.n_members) break;
        const XiMember& P = G.m[m];
        const size_t c = (size_t)P.coord_off + i;
        const double r = D.cr[c], rp0 = D.crp[c], rt0 = D.crt[c], lnz = D.clnrelz[c], growth = D.cgrowth[c];
        const double lnz2 = P.split_evol ? D.clnrelz2[c] : 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int b = min(b0 + w, B - 1);
            const double* sc = D.scal + ((size_t)b * D.n_pipe + P.pipe) * VMX_NS;
            const double ev = (P.split_evol ? vmx_exp(fma(sc[S_EV1A], lnz, sc[S_EV2A] * lnz2)) : vmx_exp((sc[S_EV1A] + sc[S_EV2A]) * lnz)) * growth;
            double xi = 0.0;
            if (r != 0.0) {
                const double rrp = sc[S_AP] * (rp0 + sc[S_DRP]), rrt = sc[S_AT] * rt0;
                const double rr2 = fma(rrp, rrp, rrt * rrt);
                if (rr2 != 0.0) {
                    const double* col = D.coef + ((size_t)P.col * B + b) * D.ncp;
                    bool o = false;
                    xi = xi_plain_spline<SAME>(D, col, ell_stride, P.n_ell, 0.5 * vmx_log(rr2), rrp * vmx_rsqrt(rr2), o) * ev;
                    oob[w] = oob[w] || o;
                }
            }
            if (P.radiation) {
                const bool resc = P.radiation == 2;
                const double drp = sc[S_DRP];
                const double rp = resc ? fma(sc[S_AP], rp0 + drp, drp) : rp0 + drp;
                const double rtr = resc ? sc[S_AT] * rt0 : rt0;
                const double rs2 = fma(rp, rp, rtr * rtr);
                const double irs = vmx_rsqrt(rs2);
                const double rs = rs2 * irs, ms = rp * irs;
                double xr = sc[S_RAD_S] * (irs * irs) * (1.0 - sc[S_RAD_A] * (1.0 - ms * ms));
                xr *= vmx_exp(-rs * fma(1.0 + ms, sc[S_RAD_IL], sc[S_RAD_ID]));
                xi += xr;
            }
            out[w] = G.presum ? fma(xi_member_factor(D, P, b), xi, out[w]) : xi;
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = b0 + w;
        if (b >= B) break;
        if (oob[w]) atomicOr(&D.status[b], VMX_STATUS_BOUNDS);
        D.xi[G.out_off + (size_t)b * G.n_pad + i] = out[w];
    }
}

// ... and the static-coordinate pipelines of an item (three basis values per bin and member), NW walkers per thread
template <int NW>
__global__ __launch_bounds__(256) void k_xi_bins_static_group(EngineDev D, const XiStaticGroup* groups, int B)
{
    const XiStaticGroup& G = groups[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G.n) return;
    const int b0 = blockIdx.z * NW;
    double out[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) out[w] = 0.0;
#pragma unroll 4
    for (int m = 0; m < G.n_members; ++m) {
        const XiMember& P = G.m[m];
        const size_t c = (size_t)P.coord_off + i;
        const double lnz = D.clnrelz[c], growth = D.cgrowth[c];
        const double lnz2 = P.split_evol ? D.clnrelz2[c] : 0.0;
        const double* y = D.poly_bins + P.poly_off + i;
        const double y0 = y[0], y1 = y[G.n_pad], y2 = y[2 * (size_t)G.n_pad];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int b = min(b0 + w, B - 1);
            const double* sc = D.scal + ((size_t)b * D.n_pipe + P.pipe) * VMX_NS;
            double c01 = sc[S_BIAS1], c02 = sc[S_BIAS2];
            const double c11 = sc[S_BB1], c12 = P.same_tracer ? sc[S_BB1] : sc[S_BB2];
            if (P.same_tracer) c02 = c01;
            const double a0 = c01 * c02, a1 = fma(c01, c12, c11 * c02), a2 = c11 * c12;
            double xi = fma(a2, y2, fma(a1, y1, a0 * y0));
            xi *= P.split_evol ? vmx_exp(fma(sc[S_EV1A], lnz, sc[S_EV2A] * lnz2)) : vmx_exp((sc[S_EV1A] + sc[S_EV2A]) * lnz);
            xi *= growth;
            out[w] = G.presum ? fma(xi_member_factor(D, P, b), xi, out[w]) : xi;
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
        if (b0 + w < B) D.xi[G.out_off + (size_t)(b0 + w) * G.n_pad + i] = out[w];
}

// k_xi_bins_static, NW walkers per thread: one walker per thread re-reads the three basis values and the two per-bin
// factors from L2 for every walker (8 loads per stored value, 2.3 GB of L2 reads per joint + metals step); here they are loaded
// once for NW walkers.  Pipelines with the standard bias evolution and no additive term (the host's list).
template <int NW>
__global__ __launch_bounds__(256) void k_xi_bins_static_nw(EngineDev D, XiLeanArgs A, int B)
{
    const XiLeanPipe& P = A.p[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    const int b0 = blockIdx.z * NW;
    const size_t c = (size_t)P.coord_off + i;
    const double lnz = D.clnrelz[c], growth = D.cgrowth[c];
    const double lnz2 = P.split_evol ? D.clnrelz2[c] : 0.0;
    const double* y = D.poly_bins + P.poly_off + i;
    const double y0 = y[0], y1 = y[P.n_pad], y2 = y[2 * (size_t)P.n_pad];
    double out[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int b = min(b0 + w, B - 1);
        const double* sc = D.scal + ((size_t)b * D.n_pipe + P.pipe) * VMX_NS;
        double c01 = sc[S_BIAS1], c02 = sc[S_BIAS2];
        const double c11 = sc[S_BB1], c12 = P.same_tracer ? sc[S_BB1] : sc[S_BB2];
        if (P.same_tracer) c02 = c01;
        const double a0 = c01 * c02, a1 = fma(c01, c12, c11 * c02), a2 = c11 * c12;
        double xi = fma(a2, y2, fma(a1, y1, a0 * y0));
        xi *= P.split_evol ? vmx_exp(fma(sc[S_EV1A], lnz, sc[S_EV2A] * lnz2)) : vmx_exp((sc[S_EV1A] + sc[S_EV2A]) * lnz);
        out[w] = xi * growth;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
        if (b0 + w < B) D.xi[P.xi_off + (size_t)(b0 + w) * P.n_pad + i] = out[w];
}

// chi2 = sum_items diff^T (C^-1 diff) + priors; sentinel on failure
// (1024 threads per walker: the kernel is a bandwidth-bound reduction with one block per walker)
__global__ __launch_bounds__(CHI2_THREADS) void k_chi2(EngineDev D, int B, SlabInfo slabs)
{
    __shared__ double red[CHI2_THREADS / 64];
    const int b = blockIdx.x;
    double acc = 0.0;
    if (D.gcinv) {
        for (int i = threadIdx.x; i < D.g_n; i += CHI2_THREADS) {
            double z = 0.0;
            for (int s = 0; s < slabs.g; ++s) z += D.gz[((size_t)s * B + b) * D.g_ld + i];
            acc = fma(D.gres[(size_t)b * D.g_ld + i], 2.0 * z, acc);         // gcinv holds the half form (see vegamx.hip)
        }
    } else {
        for (int q = 0; q < D.n_items; ++q) {
            const ItemDev& it = D.items[q];
            for (int i = threadIdx.x; i < it.n_masked; i += CHI2_THREADS) {
                const double rres = it.res[(size_t)b * it.n_masked_pad + i];
                double z;
                if (it.cinv) {
                    // up to 8 split-K slabs, requested together (a run-time trip count makes every load wait for the
                    // previous one) and added in slab order
                    const int ns = slabs.z[q];
                    double zs[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) zs[s] = it.z[((size_t)(s < ns ? s : 0) * B + b) * it.n_masked_pad + i];
                    z = zs[0];
#pragma unroll
                    for (int s = 1; s < 8; ++s) z += s < ns ? zs[s] : 0.0;
                    z *= 2.0;               // cinv holds the half form: r^T C^-1 r = 2 r^T (L r)
                } else z = rres;
                acc = fma(rres, z, acc);
            }
        }
    }
    // fixed-order reduction: within each wave by shuffles, then the 16 wave sums by one thread
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0;
        for (int w = 0; w < CHI2_THREADS / 64; ++w) c += red[w];
        const double* t = D.theta + (size_t)b * D.n_params;
        for (int q = 0; q < D.n_priors; ++q) {
            const double dlt = t[D.prior_slot[q]] - D.prior_mean[q];
            c += dlt * dlt / (D.prior_sigma[q] * D.prior_sigma[q]);
        }
        int st = D.status[b];
        if (!(c == c) || c > 1e300 || c < -1e300) { st |= VMX_STATUS_NONFINITE; D.status[b] = st; }
        D.chi2[b] = st ? 1e100 : c;
        if (D.chi2_host) D.chi2_host[b] = st ? 1e100 : c;
        if (D.status_host) D.status_host[b] = st;
        if (D.done_host) { __threadfence_system(); *D.done_host = D.done_seq; }      // (set for single-walker calls only)
        if (b == 0) {       // the next evaluation starts from an empty window and clean table flags
            D.k_live[2] = D.coef_win[0]; D.k_live[3] = D.coef_win[1]; D.coef_win[0] = 0x7fffffff; D.coef_win[1] = -1;
            xtab_key_store(D);
        }
    }
}
