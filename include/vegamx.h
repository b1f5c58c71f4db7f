/*
 * vegamx.h - C ABI of the MI355X-native Vega model + chi2 engine (libvegamx.so).
 *
 * The engine replaces, for batches of parameter points, the per-evaluation hot path of
 * andreicuceu/vega that sits behind
 *     VegaInterface.compute_model / chi2 / log_lik   (reference vega/vega_interface.py:208-387)
 *     Model.compute                                   (reference vega/model.py:157-187)
 * i.e. PowerSpectrum.compute (vega/power_spectrum.py:87-196), PktoXi.compute
 * (vega/pktoxi.py:99-163), CorrelationFunction.compute (vega/correlation_func.py:117-198),
 * Metals.compute (vega/metals.py:258-367), BroadbandPolynomials.compute
 * (vega/broadband_poly.py:74-198), the distortion-matrix product (vega/model.py:143-144)
 * and the Gaussian chi2 (vega/vega_interface.py:295-319).
 *
 * The reference is pure Python and has no FFI of its own; the binding a maintainer adds is the
 * ctypes layer shown in INTEGRATION.md (vega_amd/engine.py is that binding).
 *
 * Conventions
 *   - plain C types only; all host buffers are owned by the caller; the engine owns device memory;
 *   - every function returns 0 on success, a negative code on failure, and vmx_last_error()
 *     then describes the failure (thread-local string);
 *   - parameter points are rows of `theta` (row-major [B][n_params], fp64); a descriptor refers
 *     to a parameter by its column ("slot"); slot -1 means "absent" (Python None);
 *   - per-walker numerical failures (spline argument out of range = the reference's
 *     VegaBoundsError, NaN/Inf in the Arinyo term = VegaArinyoError) never fail the call: they
 *     set status[b] != 0 and chi2[b] = 1e100, the reference's sentinel
 *     (vega/vega_interface.py:268-279);
 *   - one engine handle = one HIP device + one stream; calls on a handle must be serialised by
 *     the caller; distinct handles are independent.  No global mutable state.
 */
#ifndef VEGAMX_H
#define VEGAMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmx_engine vmx_engine;

#define VMX_MAX_ELL 4          /* ell = 0, 2, 4, 6 (reference pktoxi.py:39,45) */
#define VMX_MAX_SMOOTH 3       /* Gaussian-type smoothing terms per pipeline */

enum { VMX_HCD_NONE = 0, VMX_HCD_ROGERS = 1, VMX_HCD_SINC = 2, VMX_HCD_FVOIGT = 3 };
enum { VMX_NL_NONE = 0, VMX_NL_ARINYO = 1, VMX_NL_MCDONALD = 2 };
enum { VMX_VD_NONE = 0, VMX_VD_GAUSS = 1, VMX_VD_LORENTZ = 2 };
enum { VMX_SCALE_UNIT = 0, VMX_SCALE_AP_AT = 1, VMX_SCALE_AISO_EPS = 2, VMX_SCALE_PHI_ALPHA = 3 };
enum { VMX_PKLIN_PEAK = 0, VMX_PKLIN_SMOOTH = 1, VMX_PKLIN_FULL = 2 };
enum { VMX_EVOL_STD = 0, VMX_EVOL_CROOM = 1 };
enum { VMX_MAT_DISTORTION = 0, VMX_MAT_INVCOV = 1, VMX_MAT_METAL = 2 };
enum { VMX_BB_PRE_MUL = 0, VMX_BB_PRE_ADD = 1, VMX_BB_POST_MUL = 2, VMX_BB_POST_ADD = 3 };
enum { VMX_BB_POLY = 0, VMX_BB_SKY = 1 };

/* status bits written per walker */
enum { VMX_STATUS_OK = 0, VMX_STATUS_BOUNDS = 1, VMX_STATUS_ARINYO = 2, VMX_STATUS_NONFINITE = 4,
       VMX_STATUS_NOT_CONSTANT = 8 };

/* A tracer's (bias, beta): two of the three slots must be present
 * (reference vega/utils.py:45-82 _tracer_bias_beta). */
typedef struct {
    int32_t bias_slot;
    int32_t bias_eta_slot;
    int32_t beta_slot;
    int32_t is_lya;        /* name == 'LYA': receives the UV / HCD effective bias */
    int32_t discrete;      /* tracer type 'discrete': receives the velocity-dispersion term */
    int32_t vd_sigma_slot; /* sigma_velo_disp_{gauss,lorentz}_<name>, -1 if not discrete */
    int32_t evol_kind;     /* VMX_EVOL_* (reference correlation_func.py:301-370) */
    int32_t alpha_slot;    /* alpha_<name> */
} vmx_tracer;

/* One P(k,mu) -> xi(r,mu) chain (a core peak / smooth component or one metal pair). */
typedef struct {
    vmx_tracer tracer[2];
    int32_t same_tracer;            /* tracer2 is tracer1 (utils.py:103-104) */
    int32_t growth_rate_slot;       /* 'growth_rate', -1 -> growth_rate_default */
    double  growth_rate_default;    /* 0.970386 (utils.py:60) or a fixed override */
    int32_t fast_metals;            /* Kaiser term without b1*b2 (power_spectrum.py:220-221) */
    int32_t pk_lin_kind;            /* VMX_PKLIN_* (model.py:177,182,184) */
    int32_t is_peak;                /* params['peak'] */

    int32_t uvb, heii;              /* power_spectrum.py:224-261 */
    int32_t bias_gamma_slot, bias_prim_slot, lambda_uv_slot, bias_gamma_e_slot, lambda_heii_slot;

    int32_t hcd_model;              /* VMX_HCD_*  (power_spectrum.py:263-358) */
    int32_t bias_hcd_slot, beta_hcd_slot, l0_hcd_slot;
    double  l0_default;             /* L0_sinc default 1 (power_spectrum.py:299) */

    int32_t nl_model;               /* VMX_NL_* after the skip-nl-model-in-peak rule */
    int32_t arinyo_slot[6];         /* q1, q2, kv, av, bv, kp  (q2 may be -1 -> 0) */
    double  arinyo_power;           /* 1 two Lya tracers, 0.5 one, 0 none (:472-477) */

    int32_t gk_table;               /* id from vmx_add_gk_table, -1: no G(k) */
    int32_t mock_los_slot;          /* -1, or the parameter p a mock's line-of-sight bin follows when p is sampled
                                     * (`mock-los-smoothing = growth | amplitude`, power_spectrum.py:143-160): one more
                                     * factor sinc(k_par mock_los_size (1 + theta[p]) / 2) per walker in the mu loop; the
                                     * static G table then holds the other binning factors only */
    double  mock_los_size;          /* the mock's bin size (`mock-bin-size`) */

    int32_t peak_nl;                /* apply compute_peak_nl (power_spectrum.py:382-417) */
    int32_t sigma_nl_par_slot, sigma_nl_per_slot;

    /* Gaussian smoothing terms: factor exp(-w (k_par^2 s_par^2 + k_perp^2 s_perp^2)) each
     * (power_spectrum.py:504-553, utils.py:396-420) */
    int32_t n_smooth;
    int32_t smooth_par_slot[VMX_MAX_SMOOTH], smooth_per_slot[VMX_MAX_SMOOTH];
    double  smooth_weight[VMX_MAX_SMOOTH];
    int32_t exp_par_slot, exp_per_slot;   /* exp smoothing (power_spectrum.py:560-586), -1 none */

    int32_t vd_kind;                /* VMX_VD_* (power_spectrum.py:588-636) */
    double  damping_scale;          /* <= 0: none (power_spectrum.py:192-194) */
    int32_t damping_power;

    int32_t n_ell;                  /* multipoles 0, 2, .. 2 (n_ell - 1) */
    int32_t scale_mode;             /* VMX_SCALE_* (scale_parameters.py:38-230) */
    int32_t scale_slot[2];
    int32_t drp_slot;               /* drp_<discrete tracer>, -1 none (correlation_func.py:65-69) */
    int32_t croom_slot[2];          /* croom_par0, croom_par1 */
    int32_t radiation;              /* QSO radiation term (correlation_func.py:446-489); 2: on the rescaled coordinates
                                     * (`rescale-coords-systematics`, :470-472) */
    int32_t rad_slot[4];            /* strength, asymmetry, lifetime, decrease */
    int32_t uv_shotnoise;           /* UV-background shot noise (correlation_func.py:649-686); 2: with the reference's
                                     * `rescale-coords-systematics` separation (:681-682) */
    int32_t uvsn_slot[3];           /* uv_shotnoise_amp, lambda_uv, bias_gamma (or bias_gamma_e) */
    int32_t single_ell;             /* -1, or ell / 2: return that multipole xi_ell(r') alone (pktoxi.py:122-155) */
    double  z_eff;
} vmx_pipe_desc;

/* One metal pair's contribution: bias1*bias2*factor * M . xi_pair (metals.py:286-334). */
typedef struct {
    int32_t pipeline;               /* id from vmx_add_pipeline */
    vmx_tracer tracer[2];           /* bias slots as seen by Metals.compute (after single-metal-beta) */
    int32_t same_tracer;
    int32_t growth_rate_slot;
    double  growth_rate_default;
    int32_t extra_bias_slot;        /* bias_<m1>_<m2> (separate-metal-auto-biases), -1 none */
    int32_t apply_bias;             /* fast_metal_bias: multiply by the bias product afterwards */
    double  multiplicity;           /* 2 for distinct metals in an auto-correlation (:238-239) */
    int32_t amplitude_slot;         /* -1, or a parameter multiplying the contribution: with `no-metal-decomp = False`
                                     * (model.py:120-123, :186) a pair enters twice, its smooth-spectrum pipeline as is
                                     * and its peak-spectrum pipeline times bao_amp */
    int32_t in_direct;              /* 1: the pair is part of a direct_pk evaluation (model.py:188-207 calls _compute_model with
                                     * the caller's spectrum: with `no-metal-decomp = False` the metal terms are computed on it,
                                     * :120-123 - the smooth-spectrum entries; with the default decomposition they are left out) */
} vmx_metal_desc;

typedef struct {
    int32_t n_model;                /* undistorted model bins */
    int32_t n_dist;                 /* distorted model bins (= output size) */
    int32_t pipe_peak, pipe_smooth; /* pipeline ids */
    int32_t bao_amp_slot;
} vmx_item_desc;

const char* vmx_last_error(void);
/* sizeof() of the structs as compiled (0 tracer, 1 pipe, 2 metal, 3 item, 4 vmx_fit_spec, 5 vmx_fit_options, 6 vmx_fit_result,
 * 7 vmx_fit_stats): lets a foreign binding verify its struct layout at load time. */
int vmx_struct_size(int32_t which);

int vmx_create(vmx_engine** out, int device);
void vmx_destroy(vmx_engine* e);

/* Template grids (vega_interface.py:690-696; power_spectrum.py:72-81; pktoxi.py:37,55).
 * pk_peak = pk_full - pk_smooth as formed by the caller (model.py:177);
 * delta2 = k^3 pk_fid / (2 pi^2) (power_spectrum.py:462).  n_mu = num_bins_muk (power_spectrum.py:52-58; default 1000): at
 * most 2048 - the mu tables of the P(k,mu) kernels live in LDS. */
int vmx_set_template(vmx_engine* e, int32_t nk, const double* k, const double* pk_peak,
                     const double* pk_smooth, const double* pk_full, const double* delta2,
                     int32_t n_mu);

/* The linear operator P_ell(k) -> cubic-spline coefficients of xi_ell on the uniform ln r knot
 * grid x0 + h*i: FFTLog (mcfit.P2xi call of pktoxi.py:141) followed by the not-a-knot spline of
 * pktoxi.py:144.  op is row-major [n_coef][nk]; n_coef = n_knots + 2. */
int vmx_set_fftlog(vmx_engine* e, int32_t ell_index, const double* op, int32_t n_coef,
                   double x0, double h, int32_t n_knots);
/* Evaluate the xi splines outside their knot range by polynomial extension instead of flagging the walker
 * (the reference's legacy old_fftlog path uses splev, which extrapolates; pktoxi.py:276-277). */
/* `fht_extrap = True` (reference vega/pktoxi.py:41,141: mcfit.P2xi(...)(pk_ell, extrap=True)): the FFTLog's input is padded
 * with power laws through its end segments instead of zeros - n_left samples F[0] (F[1] / F[0])^-t and n_right samples
 * F[n-1] (F[n-1] / F[n-2])^t, t = 1, 2, ...  Call BEFORE vmx_set_template; the operators of vmx_set_fftlog then have
 * nk + n_left + n_right columns, [samples | left pads t = 1.. | right pads t = 1..] (vega_amd/fftlog_op.xi_operator(extrap=True)),
 * and the engine forms the pad samples per walker, multipole and pipeline on the device.  0 / 0 end segments (a spectrum
 * smoothed to exact zeros at the template's last wavenumbers) give NaN there as in the reference: VMX_STATUS_NONFINITE. */
int vmx_set_fftlog_padding(vmx_engine* e, int32_t n_left, int32_t n_right);
int vmx_set_spline_extrapolation(vmx_engine* e, int32_t enabled);

/* Voigt-profile table of model-hcd = fvoigt: F(L0 k_par) by linear interpolation in (x, f), 1 below the table and
 * 0 above it (np.interp(..., left=1, right=0), power_spectrum.py:360-380).  x must be increasing. */
int vmx_set_fvoigt_table(vmx_engine* e, const double* x, const double* f, int32_t n);

/* G(k) binning table for one (bin_size_rp, bin_size_rt) pair (power_spectrum.py:481-502).
 * Returns the table id (>= 0). */
int vmx_add_gk_table(vmx_engine* e, double bin_size_rp, double bin_size_rt);
/* The same with the mock-binning factor of power_spectrum.py:143-160 folded in: G(bin sizes) * G(mock sizes);
 * a size of 0 leaves its sinc out (`mock-los-smoothing = only-los` -> mock_size_rt = 0). */
int vmx_add_gk_table_mock(vmx_engine* e, double bin_size_rp, double bin_size_rt, double mock_size_rp,
                      double mock_size_rt);

/* Returns the pipeline id (>= 0).  r, mu, z, rel_z_evol, xi_growth: [n] (correlation_func.py:46-80,252). */
int vmx_add_pipeline(vmx_engine* e, const vmx_pipe_desc* desc, int32_t n, const double* r,
                     const double* mu, const double* z, const double* rel_z_evol,
                     const double* xi_growth);

/* `new-bias-evolution` (correlation_func.py:238-299): in a cross-correlation the two tracers evolve with their own
 * redshifts z -/+ rp / (2 D_H(z)).  rel_z_1 / rel_z_2 [n] = (1 + z_tracer) / (1 + z_eff) per bin for the pipeline's
 * first / second tracer as given to vmx_add_pipeline; replaces the common rel_z_evol of that call. */
int vmx_pipeline_set_tracer_evolution(vmx_engine* e, int32_t pipeline, const double* rel_z_1,
                                      const double* rel_z_2, int32_t n);

/* Odd-multipole terms of a cross-correlation component (correlation_func.py:150-155, pktoxi.py:321-382):
 * coef [4][n_coef] = cubic B-spline coefficients (knots x0 + h i in ln r) of the Hamilton-FFTLog transforms of the
 * component's isotropic linear spectrum, in the order rel ell=1, rel ell=3, asy ell=0, asy ell=2;
 * slots = {Arel1, Arel3, Aasy0, Aasy2, Aasy3}. */
int vmx_pipeline_set_odd_terms(vmx_engine* e, int32_t pipeline, const double* coef, int32_t n_coef, double x0,
                               double h, int32_t relativistic, int32_t asymmetry, const int32_t* slots);

/* The same splines as a linear operator of the component's spectrum, for `direct_pk` (reference vega/model.py:188-207: the
 * caller's spectrum is then the pk_lin of the odd-multipole terms too, correlation_func.py:491-551): op[4][n_coef][nk], term
 * order as `coef` above, coef_term = op_term . pk (vega_amd/fftlog_op.hamilton_spline_operator).  vmx_set_direct_pk forms the
 * walkers' coefficient rows with it (one product per pipeline); without it a pipeline with odd terms refuses direct_pk. */
int vmx_pipeline_set_odd_operator(vmx_engine* e, int32_t pipeline, const double* op, int32_t n_coef, int32_t nk);

/* A(tau) table of the UV shot-noise term on the uniform grid tau0 + dtau i (correlation_func.py:597-647):
 * np.interp with left = a[0], right = 0. */
int vmx_set_shotnoise_table(vmx_engine* e, const double* a, int32_t n, double tau0, double dtau);

/* Returns the item id (>= 0). */
int vmx_add_item(vmx_engine* e, const vmx_item_desc* desc);
int vmx_item_add_metal(vmx_engine* e, int32_t item, const vmx_metal_desc* desc);
/* `fast_metals` (metals.py:144-169): a metal x metal correlation frozen at the first evaluation.  The metal is
 * added with desc.pipeline = -1 and contributes bias_product * xi[bin] with this static vector (already
 * multiplied by its metal matrix and multiplicity-free: desc.multiplicity still applies). */
int vmx_item_set_metal_static(vmx_engine* e, int32_t item, int32_t index, const double* xi, int32_t n_model);
/* Exact static form of a metal pair whose P(k,mu) is the bias-free Kaiser polynomial times static factors only
 * (power_spectrum.py:198-222 with fast_metals; no HCD / UV / non-linear / smoothing / velocity-dispersion term, fixed
 * coordinates): xi = Y0 + (beta1 + beta2) Y1 + beta1 beta2 Y2 with three static vectors (after the pair's metal matrix,
 * redshift evolution and growth).  basis = [3][n_model]; the metal is added with desc.pipeline = -1, its tracers
 * provide beta1, beta2 and the bias product per walker. */
int vmx_item_set_metal_basis(vmx_engine* e, int32_t item, int32_t index, const double* basis, int32_t n_model);
/* Set-up hook used to extract those vectors with the engine itself: while enabled, every tracer of a bias-free
 * (fast_metals) pipeline takes beta = `beta` instead of its parameters. */
int vmx_set_metal_beta_override(vmx_engine* e, int32_t enabled, double beta);

/* Parameter-level blinding (vega_interface.py:389-421 `_get_lcl_prms`, utils.py:375-393 `apply_blinding`): every
 * walker is mapped theta[p] -> scale[p] * theta[p] + shift[p] before anything reads it (model and priors alike).
 * The reference's blinding is shift = pi - exp(v^2) on the blinded names and (scale, shift) = (0, 1) on the
 * full-shape scale parameters it pins to 1.  scale, shift: [n_params] host arrays, or NULL, NULL to switch it off.
 * While a transform is set the device entry point stages the walkers through the engine's own buffer. */
int vmx_set_parameter_transform(vmx_engine* e, const double* scale, const double* shift);

/* Additive template of the non-peak component: v += amp * vec[bin] before the pre-distortion broadband
 * (DESI instrumental systematics, model.py:133-135, correlation_func.py:553-595).  amp = theta[slot], or
 * default_amp when slot = -1. */
int vmx_item_set_additive_template(vmx_engine* e, int32_t item, const double* vec, int32_t n_model,
                                   int32_t slot, double default_amp);

/* Broadband term (broadband_poly.py:119-198).
 *  VMX_BB_POLY: slots[n_coef] coefficient slots, basis [n_coef][n] = r1^i r2^j per coefficient;
 *  VMX_BB_SKY:  slots = {scale, sigma}, basis [2][n] = {rt, window(0/1)}.
 * n = n_model for pre terms, n_dist for post terms. */
int vmx_item_add_broadband(vmx_engine* e, int32_t item, int32_t position, int32_t func,
                           int32_t n_coef, const int32_t* slots, const double* basis, int32_t n);

/* Dense row-major matrices.  VMX_MAT_DISTORTION [n_dist][n_model] (model.py:143-144);
 * VMX_MAT_INVCOV [n_masked][n_masked] (vega_interface.py:316); VMX_MAT_METAL [n_model][n_pair]
 * with `index` = position of the metal in the order of vmx_item_add_metal (metals.py:338-367).
 * A matrix that is never set is the identity (data.py:77-78, :683-684). */
/* A metal matrix in Kronecker form, M = A (x) B on the bin order rt-fastest (index = rt + n_rt * rp): what
 * `new_metals` builds (metals.py:501-655: np.einsum('ij,kl->ikjl', rp_1d_dmat, rt_1d_dmat)), and with b_rt = NULL
 * (B = identity) the rp-only form (metals.py:354-358, :657-752).  The product M xi = A Xi B^T is applied as two small
 * products per walker instead of one [n_model]^2 product.  a_rp [n_rp][n_rp], b_rt [n_rt][n_rt], row-major;
 * n_rp * n_rt = n_model of the item = bins of the pair's pipeline. */
int vmx_item_set_metal_kron(vmx_engine* e, int32_t item, int32_t index, const double* a_rp, int32_t n_rp,
                            const double* b_rt, int32_t n_rt);
int vmx_item_set_matrix(vmx_engine* e, int32_t item, int32_t kind, int32_t index, int32_t rows,
                        int32_t cols, const double* dense);
/* The distortion matrix in CSR form, as the reference holds it (scipy.sparse.csr_array: data.py:342-346, :456-459; the
 * product of model.py:143-144): indptr [rows + 1] (int64), indices [nnz] (int32, STRICTLY ascending within a row: no duplicates - checked), values [nnz].
 * Replaces vmx_item_set_matrix(VMX_MAT_DISTORTION) for matrices sparse enough that streaming 12 bytes per non-zero beats
 * 8 bytes per entry; every batch size takes the CSR kernel then (8 walkers per pass over the matrix).  Before
 * vmx_finalize. */
int vmx_item_set_matrix_csr(vmx_engine* e, int32_t item, int32_t rows, int32_t cols, const int64_t* indptr,
                            const int32_t* indices, const double* values);
/* Indices (into the n_dist model bins) kept by the model mask (data.py:410). */
int vmx_item_set_mask(vmx_engine* e, int32_t item, const int32_t* idx, int32_t n_masked);
/* Masked data vector, or the current Monte-Carlo mock (vega_interface.py:311-315). */
int vmx_item_set_data(vmx_engine* e, int32_t item, const double* masked_data, int32_t n_masked);

/* Monte-Carlo fits of many mocks at once (vega/analysis.py:224-302 evaluates one mock at a time): a pool of
 * masked mock data vectors [n_mocks][n_masked] per item, and per walker the pool row it is compared with
 * (vmx_set_mock_index; -1 or a NULL index array = the item's own data vector).  Both may be called after
 * vmx_finalize; the index array applies to the following vmx_eval* calls. */
int vmx_item_set_mock_pool(vmx_engine* e, int32_t item, const double* pool, int32_t n_mocks, int32_t n_masked);
int vmx_set_mock_index(vmx_engine* e, const int32_t* index, int32_t B);
/* Mocks made on the device: the lower Cholesky factor L [n_masked][n_masked] of the (scaled) masked covariance and the fiducial
 * model on the masked bins - a mock is fiducial + L . (standard-normal draws), reference vega/data.py:737-753.  Kept until replaced;
 * used by vmx_fit_migrad with a mock stream.  vmx_item_get_mock_pool copies the first n_mocks rows of an item's pool to the host
 * (the MOCKS table of the result file, reference vega/output.py:442-520). */
int vmx_item_set_mock_factor(vmx_engine* e, int32_t item, const double* chol, const double* fiducial, int32_t n_masked);
int vmx_item_get_mock_pool(vmx_engine* e, int32_t item, double* pool, int32_t n_mocks, int32_t n_masked);
/* Page-locked host memory for buffers the library reads asynchronously - the `draws` of a vmx_mock_stream above all: from
 * pageable memory every wave's upload is staged (and, behind some runtimes, pinned page by page while the producer is still
 * first-touching the buffer: 20 ms of a 290-ms run).  Owned by the engine: vmx_host_free, or vmx_destroy at the latest. */
int vmx_host_alloc(vmx_engine* e, void** out, int64_t bytes);
int vmx_host_free(vmx_engine* e, void* p);

/* Global-covariance mode (vega_interface.py:295-304): inverse of the masked global covariance over
 * the concatenation of all items' masked bins, in item order. */
int vmx_set_global_invcov(vmx_engine* e, const double* invcov, int32_t n);

int vmx_add_prior(vmx_engine* e, int32_t slot, double mean, double sigma);

/* Allocate the per-batch workspace.  After this the engine is immutable except for
 * vmx_item_set_data / vmx_item_set_matrix(INVCOV) / vmx_set_global_invcov. */
int vmx_finalize(vmx_engine* e, int32_t n_params, int32_t max_batch);
/* Before vmx_finalize.  enabled = 0: pipelines whose P(k,mu) is the Kaiser polynomial times static factors keep their per-walker
 * P(k) -> xi path instead of the static spline-coefficient basis derived from the template's spectra (default 1: the basis).
 * Needed for direct_pk evaluations that include such pipelines (vmx_metal_desc::in_direct): a caller-supplied spectrum has no
 * static basis; vmx_set_direct_pk refuses them otherwise. */
int vmx_set_static_poly(vmx_engine* e, int32_t enabled);

/* Total output size per walker: sum of n_dist over items, in item order. */
int vmx_model_size(vmx_engine* e);
/* Column of a pipeline in the P_ell(k) / spline-coefficient stage buffers (vmx_debug_read 0 and 2, laid out
 * [ell][walker * n_columns + column][k]); pipelines whose multipoles are never formed per walker (Kaiser polynomial
 * times static factors: they carry a static coefficient basis) have none: the value is then -3 - n_columns. */
int vmx_pipeline_column(vmx_engine* e, int32_t pipeline);

/* Evaluate B parameter points.  theta [B][n_params] host memory.
 * chi2 [B] (may be NULL), model [B][vmx_model_size] (may be NULL), status [B] (may be NULL).
 * Synchronous: outputs are valid on return. */
int vmx_eval(vmx_engine* e, const double* theta, int32_t B, double* chi2, double* model,
             int32_t* status);

/* Same with theta / chi2 already in device memory (no host copies, no synchronisation: the work
 * is enqueued on the engine stream; use vmx_sync).  d_model may be NULL. */
int vmx_eval_device(vmx_engine* e, const double* d_theta, int32_t B, double* d_chi2,
                    double* d_model, int32_t* d_status);
int vmx_sync(vmx_engine* e);
/* vmx_eval_device for walkers that bring their own Monte-Carlo mock rows: d_mock_index [B] (device memory) holds, per walker, the
 * row of the items' mock pools it is compared with (-1: the item's data vector) - what vmx_set_mock_index states from the host
 * for the calls that follow, stated per call from HBM here, so that two batches with different rows can be in flight (lanes).
 * chi2 only; always eager launches (no captured graph: the batch size is expected to change from call to call).  The rows must
 * lie inside the pools (the caller wrote them; the fit driver below checks its own on the host). */
int vmx_eval_device_mocks(vmx_engine* e, const double* d_theta, int32_t B, double* d_chi2, int32_t* d_status,
                          const int32_t* d_mock_index);

/* Fits where the walkers live.  The reference minimises chi2 one MIGRAD after the other - vega/minimizer.py:66-97 per fit
 * (iminuit.Minuit(chi2, ...).migrad(ncall): a bias-only pre-fit, then the full fit from its result), vega/analysis.py:224-308 per
 * Monte-Carlo mock, bin/run_vega_mc_mpi.py:52-71 per rank.  vmx_fit_migrad runs n_fits such fits together ON THE DEVICE: every
 * fit is a resumable MIGRAD state machine (vega_amd/csrc/vmx_migrad.h - Minuit2 strategy 1 restated decision for decision:
 * internal coordinates, seed, two-point gradient, line search, Davidon update, HESSE, iminuit's `iterate` re-runs - tested on the
 * CPU under sanitizers against the per-fit reference coroutine of vega_amd/migrad.py) advanced by one thread; a round = one
 * kernel that consumes the chi2 values of every fit's last request and runs its bookkeeping up to the next one, a scan, one
 * kernel that writes the requested points as parameter rows (Minuit's transforms in the kernel), and the engine's chain over
 * those rows in chunks on alternating lanes.  Parameter rows, chi2 values and the fits' states never leave HBM; the host reads
 * ONE integer per round (the number of rows) and the results at the end.  A fit's sequence of function values is Minuit's own;
 * the fits advance at their own pace (a fit that converges goes on to HESSE / its next object while others iterate).
 *
 *   spec      the Minuit objects of a fit (1 or 2 stages; stage 1 starts from stage 0's result): free parameter columns, limits,
 *             step sizes; errordef `up`, tolerance, call limit, iminuit's `iterate` (reference: 1.0, 0.1, 100000, 5)
 *   theta0    [n_fits][n_params] host: start values and the values of the columns held fixed, per fit
 *   mock_row  [n_fits] host or NULL: the pool row (vmx_item_set_mock_pool) fit f is fitted to; NULL: the items' data
 *   opt       const_hint = vmx_set_constant_nl_hint's level for the rows of a round, or -1: derived here (a column varies when a
 *             stage frees it or the fits' rows differ in it - what vmx_eval derives from host walkers); NULL: -1, 512, 2;
 *             chunk = rows per engine call (0: 512); lanes = batches in flight (0: 2); mocks: see vmx_mock_stream - NULL: the
 *             pools hold the mocks already
 *   results   [n_stages] host arrays the caller owns: x [n_fits][n] internal minimum, ext [n_fits][n] its external values,
 *             V [n_fits][n][n] internal error matrix, fval, edm, flags (VMX_FIT_* bits), nfcn (function calls of the stage, re-runs
 *             included), n_iter
 * Synchronous; calls on a handle stay serialised by the caller.  Every argument is checked before anything runs (-1 and
 * vmx_last_error, the engine untouched).  A run that fails later (-2: a HIP error, a stalled producer of a mock stream) leaves the
 * engine usable with its hint / lanes restored; `results` and, with a mock stream, the items' mock pools then hold unspecified rows. */
#define VMX_FIT_MAXN 32
#define VMX_FIT_MAX_STAGES 2
enum { VMX_FIT_VALID = 1, VMX_FIT_HESSE_FAILED = 2, VMX_FIT_ACCURATE = 4, VMX_FIT_AT_CALL_LIMIT = 8, VMX_FIT_MADE_POSDEF = 16 };
typedef struct {
    int32_t n;
    int32_t col[VMX_FIT_MAXN];
    int32_t has_lo[VMX_FIT_MAXN], has_hi[VMX_FIT_MAXN];
    double lo[VMX_FIT_MAXN], hi[VMX_FIT_MAXN];
    double err[VMX_FIT_MAXN];
} vmx_fit_stage;
typedef struct {
    int32_t n_stages, n_params, iterate, maxfcn;
    double up, tol;
    vmx_fit_stage stage[VMX_FIT_MAX_STAGES];
} vmx_fit_spec;
/* Mocks made while their fits run (vmx_fit_migrad): the normal draws arrive from a producer on the host - the reference draws them
 * from NumPy's legacy global generator, mock after mock, a mock's correlations in turn (vega/data.py:748-757), and that stream is
 * sequential by construction - while everything else of a mock happens on the device, wave by wave: mock = fiducial + L . draws
 * with the chain's product kernels (L, fiducial: vmx_item_set_mock_factor), its row of every item's pool, its rows of the
 * quadratic form's linear terms.  Wave w (mocks [w wave, (w + 1) wave)) joins the fits at round w - a fixed schedule: what a round
 * evaluates never depends on how fast the draws arrive; the host waits for the producer when it is behind. */
typedef struct {
    int32_t n_mocks;                    /* = n_fits: fit f is fitted to mock f */
    int32_t wave;                       /* mocks per wave (0: 64) */
    const double* draws;                /* host [n_mocks][stride]: a mock's standard-normal draws, its items' side by side in item order */
    int64_t stride;
    const volatile int32_t* n_drawn;    /* host: mocks whose draws are complete (the producer counts up, release order) */
    double timeout_seconds;             /* give up when the producer stalls (0: 600) */
} vmx_mock_stream;
typedef struct { int32_t const_hint, chunk, lanes, reserved; const vmx_mock_stream* mocks; } vmx_fit_options;
typedef struct {
    double* x; double* ext; double* V; double* fval; double* edm;
    int32_t* flags; int64_t* nfcn; int32_t* n_iter;
} vmx_fit_result;
typedef struct {
    int64_t rounds, evaluations, engine_calls, fits_unfinished;
    int64_t calls_by_batch[8];          /* engine calls with 1, 2-4, 5-16, 17-64, 65-256, 257-1024, 1025-4096, more rows */
    int64_t evaluations_by_batch[8];
    double seconds, seconds_setup, seconds_rounds;
    double seconds_host_waiting;        /* of seconds_rounds: the host blocked on the stream (the GPU working) */
    double gpu_idle_seconds_between_rounds;     /* HIP events around the host's turn of every round: the stream empty */
    double seconds_waiting_for_draws;   /* mock stream: the host waiting for the producer at a wave's round */
    double seconds_enqueuing_waves;     /* mock stream: host time of the waves' copies and launches */
    double seconds_enqueuing_rounds;    /* host time of the rounds' own launches (advance, scan, emit) */
    double seconds_enqueuing_calls;     /* host time of the engine calls' launches */
} vmx_fit_stats;
int vmx_fit_migrad(vmx_engine* e, const vmx_fit_spec* spec, int32_t n_fits, const double* theta0, const int32_t* mock_row,
                   const vmx_fit_options* opt, vmx_fit_result* results, vmx_fit_stats* stats);
/* direct_pk (vega_interface.py:208-248 -> model.py:188-207): while set, every item's model is its smooth pipeline
 * (no peak component, no metals with the default no-metal-decomp; additive broadband terms enter once) evaluated with
 * the linear spectrum pk[b][nk] of walker b (host pointer, e.g. the output of a Boltzmann code per parameter point)
 * instead of the fiducial template.  pk = NULL returns to the template. */
int vmx_set_direct_pk(vmx_engine* e, const double* pk, int32_t B, int32_t nk);

/* `Model.compute(pars, pk_full, pk_smooth)` (model.py:157-187) takes the linear spectra per call: replace the
 * template's spectra (pk_peak = pk_full - pk_smooth as formed by the caller, model.py:177) after vmx_finalize.  The
 * Arinyo term keeps the fiducial Delta^2(k) of vmx_set_template (power_spectrum.py:72-73, :462).  nk must equal the
 * template's. */
int vmx_set_linear_spectra(vmx_engine* e, const double* pk_peak, const double* pk_smooth, const double* pk_full,
                           int32_t nk);

/* Small-scale marginalisation coefficients (vega_interface.py:546-579 `compute_marg_coeff`; returned by
 * chi2 / log_lik(..., return_marg_coeff=True), :282-325, which is what the PolyChord adapter calls,
 * samplers/polychord.py:106-113): coeff = M . (data - model[mask]) with the static matrix
 * M = `marg_diff2coeff_matrix` [n_templates][n_masked] (data.py:762-828).  Set before vmx_finalize;
 * vmx_marg_coeff applies it to the residuals of the LAST evaluation (B = its batch size) with the engine's product
 * kernels and copies out [B][n_templates] (host pointer; synchronous). */
int vmx_item_set_marg_matrix(vmx_engine* e, int32_t item, const double* m, int32_t n_templates, int32_t n_masked);
int vmx_marg_coeff(vmx_engine* e, int32_t item, double* out, int32_t B);

/* The mu sums of the P(k,mu) stage.  The reference sums P(k,mu) L_ell(mu) over 1000 midpoints in mu
 * (power_spectrum.py:76-77, pktoxi.py:138).  node_rule != 0 (the default): for wavenumbers up to 24 / (largest bin size)
 * the engine evaluates the first 48 and the last 48 of those midpoints and 82 fixed nodes in between - two 32-point
 * Gauss-Legendre panels for the integral plus ONE one-sided nine-point finite-difference stencil per end for the h^2, h^4
 * and h^6 end corrections of the Euler-Maclaurin formula - which reproduces the 1000-point sums, not the integral, to
 * <= 1e-13 of the largest k^3 P_ell over the whole parameter box of vmx_set_mu_rule_box (tests/test_mu_quadrature.py: the
 * reference's prior limits and wider, corners included); larger wavenumbers, and the model options
 * that are not smooth in mu (exponential smoothing, Voigt / sinc HCD, McDonald), keep the plain loop.  node_rule = 0:
 * the 1000-point loop everywhere (VMX_EXACT_MU in the environment does the same).  Returns the setting in effect. */
int vmx_set_mu_quadrature(vmx_engine* e, int32_t node_rule);
/* Applicability guard of that rule (before vmx_finalize).  The rule is validated on a parameter box - the prior limits
 * of the reference's vega/parameters/default_values.txt for every parameter that shapes P(k,mu) other than polynomially,
 * widened where the tests go further (tests/test_mu_quadrature.py: >= 100 draws incl. corners, 1e-12).  A walker with
 * theta[slots[i]] outside [lo[i], hi[i]] (or NaN) is not trusted to it: its P(k,mu) blocks run the reference's 1000-point
 * loop itself, decided per walker on the device (k_prologue), so a sampler that wanders off gets slower, never different.
 * vmx_debug_read(what = 4)[7] counts such walkers since vmx_finalize.  n = 0: no guard. */
int vmx_set_mu_rule_box(vmx_engine* e, int32_t n, const int32_t* slots, const double* lo, const double* hi);
/* The extra nodes of that rule as the engine built them: mu[n], w[n] (weights in units of one midpoint); returns n
 * (also when the buffers are NULL or too small, without writing). */
int vmx_get_mu_nodes(vmx_engine* e, double* mu, double* w, int32_t capacity);

/* Lanes.  lanes = 2: chi2-only vmx_eval_device calls (model == NULL, 64 walkers or more, the quadratic form in use)
 * alternate between the engine's own per-batch workspace and a second one - a clone that BORROWS every static tensor
 * (matrices, tables, operators; no copy) and owns its workspace, its per-batch tables and its stream - so that two
 * independent batches are in flight and the kernels of one fill the partly idle first / last block rounds of the other's
 * (an ensemble sampler's half-ensembles, Monte-Carlo realisations, nested-sampling threads; the reference's
 * bin/run_vega_mc_mpi.py:54-65 gives every rank such independent work).  Per batch the arithmetic, and therefore every bit
 * of chi2, is that of one lane.  A call returns once its kernels are enqueued, as before; vmx_sync waits for both lanes;
 * vmx_last_stream is the stream of the last vmx_eval_device (to order a consumer after it with an event; vmx_stream stays
 * the first lane's).  Any call that changes what an evaluation computes (data, mocks, covariances, parameter transform,
 * quadrature, ...) or needs the first lane's buffers (vmx_eval, model output) waits for / retires the second lane - it is
 * re-made on demand.  lanes = 1 (the default): one batch in flight. */
int vmx_set_lanes(vmx_engine* e, int32_t lanes);
void* vmx_last_stream(vmx_engine* e);

/* chi2-only evaluations (model == NULL) as a static quadratic form.  Without a multiplicative post-distortion
 * broadband the model on the fitted bins is linear in x' = [pre-distortion vector ; additive post-distortion broadband
 * coefficients], model = S DM' x' (model.py:143-149), so
 *     chi2 = r0^T C^-1 r0 - 2 dx^T (DM'^T S^T C^-1 r0) + dx^T (DM'^T S^T C^-1 S DM') dx,   dx = x' - x0',  r0 = data - S DM' x0'
 * with ONE static symmetric matrix per item in place of the distortion product (model.py:143-144) and the C^-1 product
 * (vega_interface.py:316): about half the matrix work of an evaluation, identical results to rounding.  theta_ref
 * [n_params] is the expansion point (x0' = its vector; any point the model can be evaluated at - the configured
 * parameter values - keeps the three terms small next to the signal); NULL switches the form off.  The tensors are
 * built with the engine's own kernels at the next chi2-only evaluation and refreshed when data, mocks or inverse
 * covariances change.  Not used with a global covariance, a multiplicative or non-polynomial post-distortion
 * broadband, direct_pk, or when a model output is requested (those run the full chain).  Returns 1 when the
 * configuration is eligible, 0 when it is not (the call is then a no-op). */
int vmx_set_quadratic_form(vmx_engine* e, const double* theta_ref);
/* Which form of that quadratic form the chi2-only evaluations take.  With C^-1 = U^T U (the library factors the inverse
 * covariance it was given) the same chi2 is || U r0 - F dx ||^2 with the static F = U S DM' [n_masked][nq]: 2 n_masked nq flops per
 * walker instead of the nq^2 of the half-form Q'.  kind = 0 (default): the cheaper one per engine - Q' for model grids that are the
 * data grid (2500 / 5000 bins against 1590 / 3180 fitted ones), the factored form when the model grid is much finer (a
 * `distortion-file` with COEFMOD >= 2, reference vega/data.py:441-473: nq = 10 000 against 1590); 1: Q'; 2: the factored form.
 * The two agree to rounding (the sum of squares has no cancellation at all); vmx_debug_read(what = 4)[8] reports the form the last
 * evaluation took (0: full chain, 1: Q', 2: factored).  After vmx_finalize; the tensors are rebuilt at the next chi2-only call. */
int vmx_set_quadratic_form_kind(vmx_engine* e, int32_t kind);

/* Samplers usually keep the Arinyo non-linear parameters fixed.  When they are identical for every walker of a
 * batch the factor D_NL(k,mu) G(k,mu) is tabulated once per batch instead of being exponentiated per walker and
 * grid point (level 1).  When the parameters of every Gaussian factor - smoothing (power_spectrum.py:526-556), peak
 * broadening (:382-417), Gaussian velocity dispersion (:504-524) - are shared as well, those factors join the table and a
 * second table serves the peak component (level 2): the HCD term is then the only exponential left per walker and
 * (k, mu).  A table is kept across evaluations while its parameters do not change.  vmx_eval (host theta) detects the
 * level by itself; for vmx_eval_device the caller states it here: enabled = 0 none, 1 level 1, 2 level 2.
 * A walker whose table parameters differ from the first walker's while the hint is on gets
 * status VMX_STATUS_NOT_CONSTANT and chi2 = 1e100 - never a silently wrong value. */
int vmx_set_constant_nl_hint(vmx_engine* e, int32_t enabled);
/* The engine's HIP stream (hipStream_t as an opaque pointer), so that a caller can order its own work - e.g. an
 * RCCL collective on the chi2 buffer - after vmx_eval_device without a host synchronisation. */
void* vmx_stream(vmx_engine* e);

/* Stage taps for parity tests: copy an internal buffer of the last evaluation to the host.
 * what: 0 = P_ell(k) [n_ell_max][B*n_columns][nk_pad] (vmx_pipeline_column); 1 = xi per pipeline [B][n] (index = pipeline);
 * 2 = spline coefficients; 3 = metal correlation after its metal matrix [B][pad32(n_model)] (index = metal, in the
 * global order of vmx_item_add_metal).
 * 4 = {live wavenumbers of the P(k,mu) stage in the last evaluation, k up to which the mu node rule applies (0: off),
 * nodes per wavenumber of that rule, leading wavenumbers whose tiles took the rule in the last evaluation, table level of
 * the last evaluation (vmx_set_constant_nl_hint), first and last spline-coefficient row the last evaluation's bins read,
 * walkers that left the mu rule's box since vmx_finalize, form of the last evaluation (vmx_set_quadratic_form_kind)}.
 * Returns the number of doubles written (<= capacity) or a negative error. */
int64_t vmx_debug_read(vmx_engine* e, int32_t what, int32_t index, double* out, int64_t capacity);

/* Stand-alone distortion-style product y[b] = A x[b] through the same kernels the evaluation
 * uses (bench / roofline measurement of the distortion-matrix step).  d_A row-major [rows][cols] with
 * cols a multiple of 32 (zero padded), d_x [B][cols], d_y [B][pad32(rows)], all device pointers (B <= 8 streams
 * the matrix once; larger B takes the MFMA path, split-K partial sums added in fixed order);
 * enqueued on the engine stream. */
int vmx_matvec_device(vmx_engine* e, const double* d_A, int32_t rows, int32_t cols,
                      const double* d_x, int32_t B, double* d_y);

/* Host-pointer convenience around the same product: Y[b] = A X[b] for row-major host arrays A [rows][cols],
 * X [B][cols], Y [B][rows] (padding, upload and download inside; synchronous).  Used once per Monte-Carlo run for
 * mock = fiducial + cholesky(C) . randn (data.py:751-753) on all mocks at once. */
int vmx_matmul_host(vmx_engine* e, const double* A, int32_t rows, int32_t cols, const double* X, int32_t B,
                    double* Y);

/* Restrict the event pairs of vmx_set_profiling to the kernel classes whose bit (1 << class index, the index
 * vmx_kernel_name enumerates) is set: timing one class perturbs a timed run far less than timing all of them.
 * vmx_set_profiling(e, 1) resets the mask to all classes.  Bits 28-31 of a mask other than 0xffffffff hold a sampling
 * stride minus one: n > 0 times every (n + 1)-th launch of the selected classes only (an event pair costs the queue a few
 * microseconds - several per cent of a 0.3 ms step when one sits in every step). */
int vmx_set_profiling_mask(vmx_engine* e, uint32_t kernel_class_mask);

/* Per-kernel timing with HIP events on the engine stream.  When enabled every kernel launch of
 * vmx_eval* is bracketed by events; vmx_get_timings returns the accumulated milliseconds and
 * launch counts per kernel class since the last reset. */
#define VMX_N_KERNELS 13
int vmx_set_profiling(vmx_engine* e, int32_t enabled);
int vmx_get_timings(vmx_engine* e, double* ms, int64_t* launches, int32_t reset);
const char* vmx_kernel_name(int32_t kernel_class);

#ifdef __cplusplus
}
#endif
#endif /* VEGAMX_H */
