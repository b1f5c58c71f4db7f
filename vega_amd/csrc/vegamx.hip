// libvegamx.so - host side of the C ABI declared in include/vegamx.h (gfx950 only).
//
// One engine handle owns one HIP device, one stream, the static tensors (template grids, G(k)
// tables, FFTLog+spline operators, coordinates, distortion / metal / inverse-covariance matrices)
// and a per-batch workspace sized at vmx_finalize.  vmx_eval* enqueues the kernel chain of
// vmx_device.h on the engine stream.
#include "vmx_device.h"
#include "vmx_fit.h"

#include <atomic>
#include <cmath>
#include <chrono>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <utility>
#include <string>
#include <vector>

#ifndef VMX_MAX_LANES
#define VMX_MAX_LANES 2             // (three lanes measured at the end of round 4 with an experiment build: no gain over two)
#endif

static thread_local std::string g_err;

const char* vmx_last_error(void) { return g_err.c_str(); }

static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_OK(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return fail(-2, std::string(#call) + ": " + hipGetErrorString(err__));           \
    } while (0)

#define REQUIRE(cond, msg)                                                                   \
    do { if (!(cond)) return fail(-1, std::string("invalid argument: ") + msg); } while (0)

namespace {

static const uint64_t VMX_PART_SENTINEL = 0x7ff8dead0000beefull;      // a NaN payload no arithmetic produces (k_gemv1 MODE 2 slots)


enum KernelClass {
    KC_PROLOGUE = 0, KC_PK, KC_FFTLOG, KC_XI, KC_METAL, KC_ASSEMBLE, KC_DISTORTION, KC_POST,
    KC_INVCOV, KC_CHI2, KC_MATVEC, KC_OTHER, KC_QUAD
};
const char* kKernelNames[VMX_N_KERNELS] = {
    "prologue", "pk_multipoles", "fftlog_spline_product", "xi_bins", "metal_matrix_product",
    "assemble", "distortion_product", "post", "invcov_product", "chi2", "matvec_api", "other",
    "quadratic_form_product"};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool borrowed = false;      // a copy of a DevBuf is a non-owning view (the second lane of an engine shares the static tensors)
    DevBuf() = default;
    DevBuf(const DevBuf& o) : p(o.p), n(o.n), borrowed(true) {}
    DevBuf& operator=(const DevBuf& o) { if (this != &o) { release(); p = o.p; n = o.n; borrowed = true; } return *this; }
    void forget() { if (borrowed) { p = nullptr; n = 0; borrowed = false; } }
    int alloc(size_t count, bool zero = true) {
        release();
        if (count == 0) count = 1;
        hipError_t err = hipMalloc((void**)&p, count * sizeof(T));
        if (err != hipSuccess) return fail(-2, std::string("hipMalloc: ") + hipGetErrorString(err));
        n = count;
        if (zero) {
            err = hipMemset(p, 0, count * sizeof(T));
            if (err != hipSuccess) return fail(-2, std::string("hipMemset: ") + hipGetErrorString(err));
        }
        return 0;
    }
    int upload(const T* host, size_t count) {
        if (alloc(count, false)) return -2;
        hipError_t err = hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice);
        if (err != hipSuccess) return fail(-2, std::string("hipMemcpy: ") + hipGetErrorString(err));
        return 0;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; n = 0; borrowed = false; }
    ~DevBuf() { release(); }
};

// dense row-major [rows][cols] host matrix -> device [rows][ld] with zero padded columns
static int upload_padded(DevBuf<double>& buf, const double* host, int rows, int cols, int ld)
{
    if (buf.alloc((size_t)rows * ld, true)) return -2;
    hipError_t err = hipMemcpy2D(buf.p, (size_t)ld * sizeof(double), host, (size_t)cols * sizeof(double),
                                 (size_t)cols * sizeof(double), rows, hipMemcpyHostToDevice);
    if (err != hipSuccess) return fail(-2, std::string("hipMemcpy2D: ") + hipGetErrorString(err));
    return 0;
}

// Inverse covariances are stored in "half form": L[i][j] = (C[i][j] + C[j][i]) / 2 below the diagonal, C[i][i] / 2 on
// it and 0 above, so that r^T C^-1 r = 2 r^T (L r) exactly in exact arithmetic - the product then needs the lower
// triangle only (half the flops, and half the bytes for a single walker).
static std::vector<double> half_form(const double* c, int n)
{
    std::vector<double> out((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) out[(size_t)i * n + j] = 0.5 * (c[(size_t)i * n + j] + c[(size_t)j * n + i]);
        out[(size_t)i * n + i] = 0.5 * c[(size_t)i * n + i];
    }
    return out;
}

struct MetalHost {
    MetalDev dev{};
    DevBuf<double> mat, svec, basis, kron_a, kron_b;
    int rows = 0, cols = 0;
};

struct ItemHost {
    ItemDev dev{};
    std::vector<MetalHost*> metals;
    DevBuf<double> dm, cinv, data, vec, dist, res, z, mock_pool, add_vec;
    DevBuf<double> marg, marg_out;      // marg_diff2coeff matrix [n_templates][n_masked_pad], coefficients [max_batch][pad]
    // quadratic form of chi2 (vmx_device.h): W = DM'^T S^T C^-1 [nq][n_masked_pad] is kept so that new data / mocks only
    // redo the linear terms
    DevBuf<double> q_mat, q_w, q_lin, q_c0, q_x0, q_x, q_z;
    // ... in its factored form (vmx_set_quadratic_form_kind): U with C^-1 = U^T U [n_masked][n_masked_pad], F = U S DM'
    // [n_masked][nq_pad], u0 = U r0 per data vector / mock [rows][n_masked_pad], slabs of F dx [slab_rows][n_masked_pad]
    DevBuf<double> q_u, q_f, q_u0, q_y;
    // mocks made on the device (vmx_item_set_mock_factor, vmx_fit_migrad with a mock stream): Cholesky factor [n_masked][pad] and
    // fiducial [pad]; the full C^-1 and the masked reference model of the quadratic form, kept for the per-wave linear terms;
    // per-wave scratch (draws, noise, residual rows, C^-1 rows)
    DevBuf<double> mc_chol, mc_fid, q_cfull, q_m0, mc_z, mc_noise, mc_r0, mc_t;
    bool has_factor = false;
    std::vector<double> h_q_c0;         // host copy of q_c0 (the single-walker chain adds the constants on the host)
    DevBuf<int64_t> q_basis_off;
    int q_rows = 0;                     // rows of q_lin / q_c0 (1 + mocks)
    int n_templates = 0;
    int n_mocks = 0;
    DevBuf<int32_t> inv_mask;
    DevBuf<int64_t> csr_ptr; DevBuf<int32_t> csr_idx; DevBuf<double> csr_val;      // CSR distortion matrix
    int64_t csr_nnz = 0; bool has_csr = false;
    std::vector<int32_t> mask_idx;
    bool has_dm = false, has_cinv = false, has_mask = false, has_data = false;
    bool lean_pair = false;             // k_xi_quad_plain applies: plain_pair, or that but for a radiation term on the smooth component
};

// Device-resident fits (vmx_fit_migrad): buffers kept between calls, grown on demand
struct FitWorkspace {
    DevBuf<double> state;               // [F] vmx_migrad::FitStateT<N> (N: vmx_fit_migrad picks the capacity)
    DevBuf<vmx_migrad::Spec> spec;
    DevBuf<double> base, theta, chi2;
    DevBuf<int32_t> mock_row, count, offset, done, mock, status;
    DevBuf<double> ox[vmx_migrad::MAX_STAGES], oext[vmx_migrad::MAX_STAGES], oV[vmx_migrad::MAX_STAGES], ofval[vmx_migrad::MAX_STAGES], oedm[vmx_migrad::MAX_STAGES];
    DevBuf<int32_t> oflags[vmx_migrad::MAX_STAGES], oiter[vmx_migrad::MAX_STAGES];
    DevBuf<int64_t> onfcn[vmx_migrad::MAX_STAGES];
    int32_t* pin_word = nullptr; int32_t* dpin_word = nullptr;       // mapped host memory: rows of the round, fits still running
    hipEvent_t ev_lane = nullptr;
    std::vector<hipEvent_t> ev_gap;                                   // pairs around the host's turn of a round (GPU idle time)
    ~FitWorkspace() {
        if (pin_word) (void)hipHostFree(pin_word);
        if (ev_lane) (void)hipEventDestroy(ev_lane);
        for (auto& ev : ev_gap) (void)hipEventDestroy(ev);
    }
};
template <typename T>
static int ensure(DevBuf<T>& b, size_t count) { return b.n >= count && b.p ? 0 : b.alloc(count, false); }

}  // namespace

struct vmx_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    // correlation items are independent between xi_bins and chi2: item q > 0 runs on aux[q-1] (fork / join events)
    hipStream_t cur = nullptr;
    std::vector<hipStream_t> aux;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    bool finalized = false;
    // Second lane (vmx_set_lanes): a clone that borrows every static tensor and owns its per-batch workspace and stream;
    // chi2-only device evaluations alternate between the two, so that independent batches overlap on the GPU.
    std::vector<vmx_engine*> lanes;        // lanes 1 .. n_lanes - 1 (lane 0 is this engine), made on demand
    int n_lanes = 1;
    int64_t lane_calls = 0;
    hipStream_t last_stream = nullptr;      // the stream the last vmx_eval_device ran on
    const int32_t* call_mock = nullptr;     // per-call mock rows of the walkers (device pointer; vmx_eval_device_mocks, vmx_fit_migrad)
    FitWorkspace* fitws = nullptr;

    int nk = 0, nkp = 0, n_mu = 0;
    int n_rows = 0, n_extra = 0, mu_lo = 0, mu_hi = 0;     // node rule of the mu sums (vmx_set_mu_quadrature)
    DevBuf<double> node_w, mu_img, mu_img_w;
    std::vector<int32_t> rule_slot; std::vector<double> rule_lo, rule_hi;     // vmx_set_mu_rule_box
    DevBuf<int32_t> d_rule_slot; DevBuf<double> d_rule_lo, d_rule_hi;
    double k_node_max = 0.0; bool mu_nodes_on = true;
    DevBuf<double> k, pklin, delta2, mu, sq1mmu2, lnmu, wl, gk, gk_mom, fv_x, fv_f, xtab;
    std::vector<int32_t> const_slots;        // parameters a level-1 table depends on (the Arinyo set)
    std::vector<int32_t> const_slots2;       // ... a level-2 table: + everything that enters a Gaussian factor
    DevBuf<int32_t> d_const_slots, d_const_slots2, d_xtab_pipe, d_xtab_partner;
    DevBuf<double> xtab_k;
    std::vector<int32_t> xi_static_taps, xi_static_bins;     // k_xi_bins<true> / k_xi_bins_static launch lists (set by poly_basis_build)
    DevBuf<int32_t> d_xi_static_taps, d_xi_static_bins;
    DevBuf<unsigned long long> gemm_trace;   // VMX_GEMM_TRACE=<file>: block timeline of the last FFTLog product, written by vmx_sync
    size_t gemm_trace_blocks = 0;
    DevBuf<unsigned long long> pk_trace;     // VMX_PK_TRACE=<file>: block timeline of the last k_pk_tab2 launch, written by vmx_sync
    size_t pk_trace_blocks = 0;
    std::vector<int32_t> w_groups;           // shared-W groups (indices into pk_groups): k_pk_w for large batches
    DevBuf<int32_t> d_w_groups;
    bool no_pk_w = false;                    // VMX_NO_PK_W: the shared-W groups stay in k_pk_multipoles
    std::vector<Tab2Group> tab2_groups;      // the groups with tables, as k_pk_tab2 takes them (cross groups first)
    DevBuf<double> xtab_key;
    int n_xtab = 0;
    int const_hint = 0;              // vmx_set_constant_nl_hint: table level the caller vouches for (device-resident theta)
    int fv_n = 0;
    struct GkSpec { double rp, rt, mock_rp, mock_rt; };   // G(rp, rt) * G(mock_rp, mock_rt); a zero size = factor 1
    std::vector<GkSpec> gk_tables;
    std::vector<double> h_k, h_mu;

    int n_coef = 0, ncp = 0, n_knots = 0;
    DevBuf<double> op;          // [VMX_MAX_ELL][ncp][nkp]
    bool op_set[VMX_MAX_ELL] = {false, false, false, false};
    bool extrapolate = false;
    double x0[VMX_MAX_ELL] = {0}, h[VMX_MAX_ELL] = {0};

    std::vector<PipeDev> pipes;
    std::vector<double> h_r, h_mu_c, h_z, h_relz, h_lnrelz, h_lnrelz2, h_growth;
    DevBuf<double> cr, cmu, crp, crt, cz, crelz, clnrelz, clnrelz2, cgrowth;
    DevBuf<PipeDev> d_pipes;
    std::vector<PkGroup> pk_groups;
    DevBuf<PkGroup> d_pk_groups;
    std::vector<int32_t> pk_members, pk_poly, pk_static;     // pk_static: polynomial pipelines with a static coefficient basis
    DevBuf<int32_t> d_pk_members, d_pk_poly, d_pk_static, d_pipe_active;
    DevBuf<double> poly_coef, poly_bins;
    int64_t poly_bins_total = 0;
    int n_active = 0;
    bool poly_dirty = true;

    std::vector<ItemHost*> items;
    std::vector<MetalHost*> metals;
    DevBuf<ItemDev> d_items;
    DevBuf<MetalDev> d_metals;
    std::vector<double> h_bb, h_odd, h_odd_op;
    std::map<int, int64_t> odd_op_off;        // pipeline -> its odd-multipole operator in odd_op (vmx_pipeline_set_odd_operator)
    DevBuf<double> bb_basis, odd_coef, odd_op, odd_dyn, sn_a;
    int sn_n = 0; double sn_tau0 = 0.0, sn_dtau = 1.0;

    std::vector<int32_t> prior_slot;
    std::vector<double> prior_mean, prior_sigma;
    DevBuf<int32_t> d_prior_slot;
    DevBuf<double> d_prior_mean, d_prior_sigma;

    DevBuf<double> gcinv, gres, gz;
    int g_n = 0, g_ld = 0;

    int n_params = 0, max_batch = 0, model_size = 0, slab_rows = 0;
    int pad_l = 0, pad_r = 0;           // vmx_set_fftlog_padding
    int fact_slab_rows = 0;          // rows of the factored form's slabs of F dx (small rows: up to 4 K splits of a full batch)
    std::map<int, std::vector<int>> group_splits;     // K splits of the grouped launches per (stage, batch size)
    // tapes of the quadratic-form launches per number of walker tiles: the blocks' entries, their queues, the partial-sum slots
    struct QuadList { DevBuf<GemmWork> work; DevBuf<int32_t> queue, nt_off; DevBuf<double> part; int n_blocks = 0; int n_entries = 0; };
    std::map<int, QuadList*> quad_lists;     // by number of walker tiles
    std::vector<void*> host_allocs;          // vmx_host_alloc: pinned host buffers handed to the caller, freed with the engine at the latest
    std::map<int, QuadList*> cinv_lists;     // the same tape over the inverse covariances (chi2 of the full chain), by walker tiles
    DevBuf<double> zero_row;                 // zeros, as long as the longest padded residual: that contraction has no linear term
    int quad_blocks = 0;             // persistent blocks of the quadratic-form launch: 2 per CU
    std::vector<double> host_key, pending_key;   // vmx_eval: shared parameters the level-2 tables hold / seen in the last call
    bool host_key_valid = false, skip_xtab_once = false;
    bool fft_ring = true, fft_ring_attr = false;     // (false: the device refused the ring's 128 KB of LDS - the two-buffer kernel then)
    // pre-summed bins (xi_sum_plan): groups of the lean / static-coordinate pipelines of an item, and what assemble_bin is told
    bool xi_sums = true;             // VMX_NO_XI_SUMS: one array per pipeline, as the general kernels write them
    bool sums_dirty = true;
    std::vector<XiLeanGroup> lean_groups; std::vector<XiStaticGroup> static_groups;
    std::vector<int32_t> static_single;              // static-coordinate pipelines that keep their own array
    DevBuf<XiLeanGroup> d_lean_groups; DevBuf<XiStaticGroup> d_static_groups; DevBuf<ItemSums> d_item_sums;
    DevBuf<int32_t> d_static_single;
    bool sums_any = false;
    // (members per array: four lean, sixteen static-coordinate pipelines - measured at B = 512: groups of 2 / 3 lean members
    // 462k / 477k evaluations / s against 474k, 4 / 8 static members 472k / 473k)
    bool xi_lean = true;             // VMX_NO_XI_LEAN: every per-walker pipeline's bins by the general k_xi_bins
    std::vector<int32_t> xi_lean_pipes, xi_rest_pipes;      // the active pipelines k_xi_bins_lean serves / the others
    DevBuf<int32_t> d_xi_rest_pipes;
    bool ring_allowed = true;        // one batch in flight only: a 128 KB block leaves the other lane's kernels no room on its CU
    int last_tab_level = 0;          // table level of the last chain (vmx_debug_read what = 4)
    bool no_tab2 = false;            // VMX_NO_TAB2: level-1 tables only (the Gaussian factors stay in the mu loop)
    bool no_cinv_tape = false;       // VMX_NO_CINV_TAPE: chi2 of the full chain by the C^-1 products + k_chi2 at every batch size
    bool pk_small_attr = false;      // the single-walker P(k) shape asked for its > 64 KB of LDS
    bool kron_attr = false;          // k_metal_kron asked for its dynamic LDS
    DevBuf<double> mv_part;          // split-K slabs of the stand-alone product
    int64_t xi_total = 0, xim_total = 0;
    DevBuf<double> theta, scal, metal_bias, pl, coef, xi, xim, model, chi2;
    DevBuf<int32_t> status, mock_index, k_live, coef_win;
    DevBuf<double> pk_direct;        // [max_batch][nkp] per-walker linear spectra of the direct_pk mode
    bool direct = false;
    std::vector<int32_t> h_mock_index;
    int last_B = 0;
    bool last_full = false;          // the last evaluation ran the full chain (model and residuals are valid)
    bool last_taps = true;           // ... wrote the per-pipeline bins (vmx_debug_read what = 1)
    // quadratic form of chi2: used when only chi2 is asked for (see vmx_set_quadratic_form)
    std::vector<double> theta_ref;
    bool quad_eligible = false, quad_mat_dirty = true, quad_lin_dirty = true, no_fuse = false;
    bool static_poly = true;         // vmx_set_static_poly
    int quad_kind = 0;               // vmx_set_quadratic_form_kind: 0 = the cheaper form, 1 = Q', 2 = factored
    bool quad_factored = false;      // the form the tensors were built for
    int last_form = 0;               // form of the last evaluation: 0 full chain, 1 Q', 2 factored (vmx_debug_read 4 [8])
    // VMX_TRACE_HOST: host-side phases of vmx_eval accumulated in nanoseconds (staging, enqueue, wait), printed at destroy
    bool trace_host = false; double host_ns[3] = {0, 0, 0}; int64_t host_calls = 0;
    EngineDev dev{};

    // host path: pinned staging buffers and one captured graph per batch size
    double* pin_theta = nullptr; double* pin_chi2 = nullptr; int32_t* pin_status = nullptr;
    std::vector<double> blind_scale, blind_shift;       // parameter-level blinding (empty = off); device copy [2][n_params]
    DevBuf<double> d_blind;
    double* dpin_theta = nullptr; double* dpin_chi2 = nullptr; int32_t* dpin_status = nullptr;   // device views
    int64_t* pin_done = nullptr; int64_t* dpin_done = nullptr; int64_t done_seq = 0;          // completion word of single-walker calls
    // single-walker quadratic form: one number per block of the last products (k_gemv1 MODE 2), added up by the host
    double* pin_part = nullptr; double* dpin_part = nullptr;
    int host_reduce_items = 0, host_reduce_blocks[VMX_MAX_GROUP] = {0};
    bool no_host_reduce = false;     // VMX_NO_HOST_REDUCE
    std::map<int, hipGraphExec_t> graphs;
    bool use_graphs = true;

    // profiling
    bool profiling = false;
    uint32_t prof_mask = 0xffffffffu;     // kernel classes that get event pairs while profiling
    int prof_stride = 1; int64_t prof_count[32] = {};      // ... every prof_stride-th launch of each of them (vmx_set_profiling_mask)
    struct Span { hipEvent_t a, b; int kc; };
    std::vector<Span> spans;
    size_t span_used = 0;
    double ms[VMX_N_KERNELS] = {0};
    int64_t launches[VMX_N_KERNELS] = {0};

    ~vmx_engine() {
        for (auto* l : lanes) { (void)hipStreamSynchronize(l->stream); delete l; }
        lanes.clear();
        for (auto* it : items) delete it;
        for (auto* m : metals) delete m;
        for (auto& s : spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
        for (auto& q : quad_lists) delete q.second;
        for (auto& q : cinv_lists) delete q.second;
        for (void* p : host_allocs) (void)hipHostFree(p);
        delete fitws;
        if (pin_theta) (void)hipHostFree(pin_theta);
        if (pin_chi2) (void)hipHostFree(pin_chi2);
        if (pin_status) (void)hipHostFree(pin_status);
        if (pin_done) (void)hipHostFree(pin_done);
        if (pin_part) (void)hipHostFree(pin_part);
        for (auto& a : aux) (void)hipStreamDestroy(a);
        for (auto& ev : ev_join) (void)hipEventDestroy(ev);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

// Anything that changes what an evaluation computes retires the second lane (it borrows the static tensors and copies
// the host-side state at the moment it is made); the next two-lane evaluation makes a fresh one.
static void drop_lane(vmx_engine* e)
{
    if (!e) return;
    for (auto* l : e->lanes) { (void)hipStreamSynchronize(l->stream); delete l; }
    e->lanes.clear();
}
static void wait_lane(vmx_engine* e) { if (e) for (auto* l : e->lanes) (void)hipStreamSynchronize(l->stream); }

struct ScopedTimer {
    vmx_engine* e; int idx = -1;
    ScopedTimer(vmx_engine* eng, int kc) : e(eng) {
        if (!e->profiling || !((e->prof_mask >> kc) & 1u)) return;
        if (e->prof_stride > 1 && (e->prof_count[kc]++ % e->prof_stride) != 0) return;
        if (e->span_used == e->spans.size()) {
            vmx_engine::Span s{};
            if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
            e->spans.push_back(s);
        }
        idx = (int)e->span_used++;
        e->spans[idx].kc = kc;
        (void)hipEventRecord(e->spans[idx].a, e->cur);
    }
    ~ScopedTimer() { if (idx >= 0) (void)hipEventRecord(e->spans[idx].b, e->cur); }
};

static void collect_spans(vmx_engine* e)
{
    for (size_t i = 0; i < e->span_used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e->spans[i].a, e->spans[i].b) == hipSuccess) {
            e->ms[e->spans[i].kc] += t;
            e->launches[e->spans[i].kc] += 1;
        }
    }
    e->span_used = 0;
}

// true when `peak` evaluates the same P(k,mu) factors as `smooth` apart from the peak-only Gaussian
// broadening (and the k-only linear spectrum): then both are produced by one pass of k_pk_multipoles
static bool pk_stage_compatible(const vmx_pipe_desc& smooth, const vmx_pipe_desc& peak)
{
    vmx_pipe_desc a = smooth, b = peak;
    if (a.is_peak || !b.is_peak || a.peak_nl || !b.peak_nl) return false;
    // neutralise the fields that are allowed to differ, then compare everything the P(k) stage reads
    b.is_peak = a.is_peak; b.pk_lin_kind = a.pk_lin_kind; b.peak_nl = a.peak_nl;
    b.sigma_nl_par_slot = a.sigma_nl_par_slot; b.sigma_nl_per_slot = a.sigma_nl_per_slot;
    b.scale_mode = a.scale_mode; b.scale_slot[0] = a.scale_slot[0]; b.scale_slot[1] = a.scale_slot[1];
    b.radiation = a.radiation;
    return std::memcmp(&a, &b, sizeof(vmx_pipe_desc)) == 0;
}

// true when two pipelines without amplitude-dependent mu structure share the factor W(k, mu) (G table, Gaussian
// exponent, velocity dispersion): they may differ in the tracers' bias / beta, the linear spectrum and the xi stage
static bool w_stage_compatible(const vmx_pipe_desc& a, const vmx_pipe_desc& b)
{
    if (a.gk_table != b.gk_table || a.is_peak != b.is_peak || a.peak_nl != b.peak_nl) return false;
    if (a.mock_los_slot >= 0 || b.mock_los_slot >= 0) return false;        // (a per-walker factor of the mu loop)
    if (a.peak_nl && (a.sigma_nl_par_slot != b.sigma_nl_par_slot || a.sigma_nl_per_slot != b.sigma_nl_per_slot ||
                      a.growth_rate_slot != b.growth_rate_slot || a.growth_rate_default != b.growth_rate_default))
        return false;
    if (a.n_smooth != b.n_smooth || a.exp_par_slot != b.exp_par_slot || a.exp_per_slot != b.exp_per_slot) return false;
    for (int i = 0; i < a.n_smooth; ++i)
        if (a.smooth_par_slot[i] != b.smooth_par_slot[i] || a.smooth_per_slot[i] != b.smooth_per_slot[i] ||
            a.smooth_weight[i] != b.smooth_weight[i]) return false;
    if (a.vd_kind != b.vd_kind) return false;
    for (int q = 0; q < 2; ++q)
        if (a.vd_kind != VMX_VD_NONE && (a.tracer[q].discrete != b.tracer[q].discrete ||
                                         (a.tracer[q].discrete && a.tracer[q].vd_sigma_slot != b.tracer[q].vd_sigma_slot)))
            return false;
    return a.hcd_model == VMX_HCD_NONE && b.hcd_model == VMX_HCD_NONE && a.nl_model == VMX_NL_NONE &&
           b.nl_model == VMX_NL_NONE;
}

// compile-time specialisation of the mu loop that matches a pipeline (PKV_GENERIC when none does)
static int pk_variant(const vmx_pipe_desc& d, bool paired)
{
    const bool rare = d.hcd_model == VMX_HCD_SINC || d.hcd_model == VMX_HCD_FVOIGT || d.nl_model == VMX_NL_MCDONALD || d.exp_par_slot >= 0 ||
                      d.mock_los_slot >= 0 ||
                      (d.fast_metals && (d.tracer[0].is_lya || d.tracer[1].is_lya) &&
                       (d.uvb || d.heii || d.hcd_model != VMX_HCD_NONE));
    if (rare) return PKV_GENERIC;
    // no mu-dependent factor besides the Kaiser polynomial and the static G table: closed form in the moments
    if (!rare && !paired && !d.is_peak && !d.peak_nl && d.hcd_model == VMX_HCD_NONE && d.nl_model == VMX_NL_NONE &&
        d.n_smooth == 0 && d.exp_par_slot < 0 && d.vd_kind == VMX_VD_NONE)
        return PKV_POLY;
    const bool rogers = d.hcd_model == VMX_HCD_ROGERS;
    const bool hcd1 = rogers && d.tracer[0].is_lya, hcd2 = rogers && d.tracer[1].is_lya;
    const bool arinyo = d.nl_model == VMX_NL_ARINYO;
    const bool lorentz = d.vd_kind == VMX_VD_LORENTZ;
    const bool vd1 = lorentz && d.tracer[0].discrete, vd2 = lorentz && d.tracer[1].discrete;
    if (vd1 || d.gk_table < 0) return PKV_GENERIC;
    if (d.same_tracer) {
        if (hcd1 && arinyo && paired && !vd2) return PKV_AUTO_CORE;
        if (!hcd1 && !arinyo && !paired && !vd2) return PKV_PLAIN_SAME;
        return PKV_GENERIC;
    }
    if (hcd1 && !hcd2 && arinyo && paired && vd2) return PKV_CROSS_CORE;
    if (!hcd1 && !hcd2 && !arinyo && !paired) return vd2 ? PKV_PLAIN_PAIR_VD : PKV_PLAIN_PAIR;
    return PKV_GENERIC;
}

// the single-walker streaming kernel keeps x in LDS
static bool gemv1_applies(int N, int K) { return N == 1 && K <= 5120 && (size_t)K * sizeof(double) <= 48 * 1024; }

// CSR distortion product of one item for B walkers -> it->dist (one slab)
static void launch_csr(vmx_engine* e, ItemHost* it, int B)
{
    const ItemDev& d = it->dev;
    ScopedTimer timer(e, KC_DISTORTION);
    const int rows = d.d.n_dist;
#define VMX_CSR(NB) hipLaunchKernelGGL(k_csr_spmm<NB>, dim3((rows + 3) / 4, (B + NB - 1) / NB), dim3(256), 0, e->cur, it->csr_ptr.p, \
                                       it->csr_idx.p, it->csr_val.p, rows, it->vec.p, d.n_model_pad, B, it->dist.p, d.n_dist_pad)
    if (B == 1) VMX_CSR(1);
    else if (B == 2) VMX_CSR(2);
    else if (B <= 4) VMX_CSR(4);
    else VMX_CSR(8);
#undef VMX_CSR
}

// persistent blocks of the single-walker streaming kernel (2 per CU), rows strided over them
static int gemv1_blocks(int M)
{
    int blocks = 512;
    while ((M + blocks - 1) / blocks > GEMV1_MAX_ROWS) blocks *= 2;
    return blocks > M ? M : blocks;
}

// Tiling of one MFMA product: fills the tile / split fields of `g`, returns the number of K slabs and, in *per_xcd,
// the blocks each XCD runs for it.  `other_tiles`: tiles of the other problems of the same launch.
static int plan_gemm(vmx_engine* e, GemmArgs& g, int nbatch, int slab_rows_avail, const int32_t* k_limit, bool tri,
                     int other_tiles, int* per_xcd, int forced_split = 0)
{
    constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_BK;
    const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
    const int tm_eff = tri ? (tm + 1) / 2 : tm;       // a triangular matrix pairs its row tiles (k_gemm_nt)
    const int tiles = tm_eff * tn * nbatch + other_tiles;
    // split K (1, 2, 4 or 8 ways: a split belongs to whole XCDs) until the launch has at least 512 blocks (2 per CU);
    // partial sums go to separate slabs, which the consumer kernels re-read: more splits cost there
    int nsplit = 1;
    while (nsplit < 8 && tiles * nsplit < 512) nsplit *= 2;
    if (forced_split > 0) nsplit = forced_split;
    while (nsplit > 1 && (int64_t)nsplit * g.N > slab_rows_avail) nsplit /= 2;
    int klen = ((g.K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
    while (nsplit > 1 && (int64_t)klen * (nsplit - 1) >= g.K) { nsplit /= 2; klen = ((g.K + nsplit - 1) / nsplit + BK - 1) / BK * BK; }
    g.nsplit = nsplit; g.klen = klen; g.d_slab = (int64_t)g.N * g.ldd;
    g.tm = tm; g.tn = tn; g.k_limit = k_limit; g.tri = tri ? 1 : 0;
    const int ngroups = 8 / nsplit;
    *per_xcd = ((tm_eff + ngroups - 1) / ngroups) * tn;
    if (g.m_window && nsplit == 1) *per_xcd = tm_eff * ((tn + 7) / 8);     // walker tiles over the XCDs (k_gemm_nt44: n_major)
    return nsplit;
}

// K splits of the problems of one grouped launch.  A CU works through its blocks one after the other (the SIMDs issue
// from the oldest wave first, so the second resident block only fills gaps), which makes a launch a list-scheduling
// problem: blocks of `stages / split` K stages (+ a fixed start / end cost), handed in launch order to the first free of
// the 256 CUs.  Every combination of 1 / 2 / 4 / 8-way splits is simulated (up to three problems; beyond that the
// block-count rule of plan_gemm applies) and the shortest makespan wins, extra slabs charged with their write + re-read
// at ~3 TB/s.
// (vmx_plan::choose_group_splits, vmx_plan.h: both models, run on the CPU by tests/test_planner_host.py)
using vmx_plan::SplitProblem;
using vmx_plan::choose_group_splits;

static int gemm_tiles(int M, int N, bool tri)
{
    const int tm = (M + GEMM_BM - 1) / GEMM_BM, tn = (N + GEMM_BN - 1) / GEMM_BN;
    return (tri ? (tm + 1) / 2 : tm) * tn;
}

static void launch_gemm_group(vmx_engine* e, int kc, const GemmGroup& G, int per_xcd_total, int nbatch)
{
    constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_BK;
    dim3 grid(8 * per_xcd_total, nbatch), block(256);
    // the four-block 4x4x4 MFMA kernel runs every full product (distortion and metal matrices, FFTLog o spline,
    // stand-alone products); the short triangular C^-1 products are faster on the 16x16x4 kernel, whose two resident
    // blocks hide each other's start and end (0.093 against 0.103 - 0.126 ms: two short passes per block)
    if (kc != KC_INVCOV) {
        block = dim3(GEMM44_THREADS);
        // (a windowed FFTLog launch carries its batch - the multipoles - inside grid.x: GemmGroup::batch_in_x)
        GemmGroup Gx = G;
        dim3 gridx = grid;
        // (when the launch has the chip to itself; with several batches in flight the late starters fill the other lane's gaps)
        const bool batch_x = e->ring_allowed;
        if (batch_x && kc == KC_FFTLOG && G.n == 1 && G.p[0].m_window && G.work == nullptr && nbatch > 1) { Gx.batch_in_x = nbatch; gridx = dim3(grid.x * nbatch, 1); }
        switch (kc) {
            case KC_QUAD: hipLaunchKernelGGL((k_gemm_nt44<KC_QUAD>), grid, block, 0, e->cur, G); break;
            case KC_DISTORTION: hipLaunchKernelGGL((k_gemm_nt44<KC_DISTORTION>), grid, block, 0, e->cur, G); break;
            case KC_METAL: hipLaunchKernelGGL((k_gemm_nt44<KC_METAL>), grid, block, 0, e->cur, G); break;
            case KC_FFTLOG:
                if (getenv("VMX_GEMM_TRACE")) {
                    e->gemm_trace_blocks = (size_t)2 * grid.x * grid.y;       // (the 64 x 32 tiling has up to twice the blocks)
                    if (e->gemm_trace.n < 4 * e->gemm_trace_blocks) (void)e->gemm_trace.alloc(4 * e->gemm_trace_blocks, true);
                    else (void)hipMemsetAsync(e->gemm_trace.p, 0, 4 * e->gemm_trace_blocks * sizeof(unsigned long long), e->cur);
                    const_cast<GemmGroup&>(G).trace = e->gemm_trace.p;
                }
                // The four-stage ring pays when the launch has about one live tile per CU: a lone block then streams at its own
                // pace (B = 256: 49 -> 44 us).  With fewer tiles its slower dispatch (128 KB of LDS per block) costs more than
                // it gains (B = 64: 32 -> 35 us); with more, two resident two-buffer blocks hide each other's latencies better
                // (B = 1024: 103 -> 138 us).  The live rows are device data: a third of the operator's rows is the estimate.
                if (G.n == 1 && G.p[0].m_window && G.p[0].nsplit == 1 && !G.p[0].tri) {
                    // about one live 64 x 64 tile per CU (the operator's live rows are device data: a third of them is the
                    // estimate): 64 x 32 tiles instead - twice the blocks, two per CU, each covering the other's bubbles
                    const int64_t est = (int64_t)G.p[0].tn * nbatch * ((G.p[0].tm * 3 + 9) / 10);
                    if (est >= 96 && est <= 384) {
                        GemmGroup G2 = G;
                        const int tn32 = (G.p[0].N + 31) / 32;
                        G2.p[0].tn = tn32;
                        const dim3 grid2 = batch_x ? dim3(8 * G.p[0].tm * ((tn32 + 7) / 8) * nbatch, 1) : dim3(8 * G.p[0].tm * ((tn32 + 7) / 8), nbatch);
                        G2.batch_in_x = batch_x ? nbatch : 0;         // (the live blocks of all multipoles first in launch order)
                        // (+ 24 KB of unused dynamic LDS: 72 KB per block = two per CU.  With its own 48 KB the dispatcher packs three
                        // blocks on a CU before it moves on and leaves a third of the CUs empty.)
                        const size_t pad = 24 * 1024;
                        hipLaunchKernelGGL((k_gemm_nt44<KC_FFTLOG, 2, 32>), grid2, block, pad, e->cur, G2);
                        break;
                    }
                }
                if (e->fft_ring && e->ring_allowed && G.n == 1 && G.p[0].m_window) {
                    const int64_t est = (int64_t)G.p[0].tn * nbatch * ((G.p[0].tm * 3 + 9) / 10);
                    if (est < 160 || est > 320) { hipLaunchKernelGGL((k_gemm_nt44<KC_FFTLOG>), gridx, block, 0, e->cur, Gx); break; }
                    constexpr size_t ring_bytes = (size_t)4 * (GEMM_BM + GEMM_BN) * GEMM_BK * sizeof(double);
                    if (!e->fft_ring_attr) {
                        if (hipFuncSetAttribute((const void*)k_gemm_nt44<KC_FFTLOG, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_bytes) != hipSuccess) {
                            (void)hipGetLastError();
                            e->fft_ring = false;
                            hipLaunchKernelGGL((k_gemm_nt44<KC_FFTLOG>), gridx, block, 0, e->cur, Gx);
                            break;
                        }
                        e->fft_ring_attr = true;
                    }
                    hipLaunchKernelGGL((k_gemm_nt44<KC_FFTLOG, 4>), gridx, block, ring_bytes, e->cur, Gx);
                } else hipLaunchKernelGGL((k_gemm_nt44<KC_FFTLOG>), gridx, block, 0, e->cur, Gx);
                break;
            default: hipLaunchKernelGGL((k_gemm_nt44<KC_OTHER>), grid, block, 0, e->cur, G); break;
        }
        return;
    }
    hipLaunchKernelGGL((k_gemm_nt<BM, BN, BK, KC_INVCOV>), grid, block, 0, e->cur, G);
}

// D[n][m] = sum_k A[m][k] X[n][k]; returns the number of K slabs written (consumer sums them).
static int launch_product(vmx_engine* e, int kc, const double* A, int lda, int64_t a_batch, int M, int K,
                          const double* X, int ldx, int64_t x_batch, int N, double* D, int ldd,
                          int64_t d_batch, int nbatch, int slab_rows_avail, const int32_t* k_limit = nullptr,
                          int fused_item = -1, bool tri = false, const int32_t* m_window = nullptr)
{
    GemmArgs g{};
    g.A = A; g.lda = lda; g.a_batch = a_batch;
    g.X = X; g.ldx = ldx; g.x_batch = x_batch;
    g.D = D; g.ldd = ldd; g.d_batch = d_batch;
    g.M = M; g.N = N; g.K = K; g.tri = tri ? 1 : 0; g.m_window = m_window;
    ScopedTimer timer(e, kc);
    if (gemv1_applies(N, K)) {
        // persistent streaming kernel: 2 blocks per CU, rows strided over blocks
        g.nsplit = 1; g.klen = K; g.d_slab = 0;
        const int blocks = gemv1_blocks(M);
        dim3 grid(blocks, 1, nbatch), block(256);
        const size_t shmem = (size_t)K * sizeof(double);
        if (fused_item >= 0) {
            if (K <= 2560) hipLaunchKernelGGL((k_gemv1<5, 1>), grid, block, shmem, e->cur, g, e->dev, fused_item);
            else hipLaunchKernelGGL((k_gemv1<10, 1>), grid, block, shmem, e->cur, g, e->dev, fused_item);
        } else {
            if (K <= 2560) hipLaunchKernelGGL((k_gemv1<5, 0>), grid, block, shmem, e->cur, g, e->dev, 0);
            else hipLaunchKernelGGL((k_gemv1<10, 0>), grid, block, shmem, e->cur, g, e->dev, 0);
        }
        return 1;
    }
    if (N <= 8) {
        g.nsplit = 1; g.klen = K; g.d_slab = 0;
        dim3 grid((M + 3) / 4, 1, nbatch), block(256);
        switch (N) {
            case 1: hipLaunchKernelGGL(k_gemv<1>, grid, block, 0, e->cur, g); break;
            case 2: hipLaunchKernelGGL(k_gemv<2>, grid, block, 0, e->cur, g); break;
            case 3: case 4: hipLaunchKernelGGL(k_gemv<4>, grid, block, 0, e->cur, g); break;
            default: hipLaunchKernelGGL(k_gemv<8>, grid, block, 0, e->cur, g); break;
        }
        return 1;
    }
    GemmGroup G{};
    int per_xcd = 0;
    const int nsplit = plan_gemm(e, g, nbatch, slab_rows_avail, k_limit, tri, 0, &per_xcd);
    G.p[0] = g; G.n = 1; G.seq_end[0] = per_xcd;
    launch_gemm_group(e, kc, G, per_xcd, nbatch);
    return nsplit;
}

}  // namespace

// capacity N of the fit states and fits per block T: N = the smallest of 4 / 8 / 16 / 32 that holds the stages; T x the state's
// size fits a CU's LDS (vmx_fit.h).  One fit per block: the threads of a wave are in different phases of their fits and a wave
// executes the union of its threads' paths (vmx_fit.h)
template <int N, int T>
static int fit_launch_round(const FitDev& D, hipStream_t st, size_t emit_lds, bool set_attr)
{
    constexpr size_t lds = (size_t)T * fit_lds_stride<N>() * sizeof(double);
    static_assert(lds <= 160 * 1024 - 1024, "fit states of a block must fit a CU's LDS");
    if (set_attr && lds > 64 * 1024)
        HIP_OK(hipFuncSetAttribute((const void*)k_fit_advance<N, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (set_attr) return 0;
    hipLaunchKernelGGL((k_fit_advance<N, T>), dim3((D.F + T - 1) / T), dim3(FIT_THREADS), lds, st, D);
    hipLaunchKernelGGL(k_fit_scan, dim3(1), dim3(1024), 0, st, D);
    hipLaunchKernelGGL((k_fit_emit<N>), dim3(D.F), dim3(64), emit_lds, st, D);
    return 0;
}
static int fit_round(int cap, const FitDev& D, hipStream_t st, size_t emit_lds, bool set_attr)
{
    switch (cap) {
    case 4: return fit_launch_round<4, 1>(D, st, emit_lds, set_attr);
    case 8: return fit_launch_round<8, 1>(D, st, emit_lds, set_attr);
    case 16: return fit_launch_round<16, 1>(D, st, emit_lds, set_attr);
    default: return fit_launch_round<32, 1>(D, st, emit_lds, set_attr);
    }
}
static size_t fit_state_bytes(int cap)
{
    switch (cap) {
    case 4: return sizeof(vmx_migrad::FitStateT<4>);
    case 8: return sizeof(vmx_migrad::FitStateT<8>);
    case 16: return sizeof(vmx_migrad::FitStateT<16>);
    default: return sizeof(vmx_migrad::FitStateT<32>);
    }
}

extern "C" {

int vmx_struct_size(int32_t which)
{
    switch (which) {
        case 0: return (int)sizeof(vmx_tracer);
        case 1: return (int)sizeof(vmx_pipe_desc);
        case 2: return (int)sizeof(vmx_metal_desc);
        case 3: return (int)sizeof(vmx_item_desc);
        case 4: return (int)sizeof(vmx_fit_spec);
        case 5: return (int)sizeof(vmx_fit_options);
        case 6: return (int)sizeof(vmx_fit_result);
        case 7: return (int)sizeof(vmx_fit_stats);
        default: return -1;
    }
}

int vmx_create(vmx_engine** out, int device)
{
    REQUIRE(out != nullptr, "out is null");
    int count = 0;
    HIP_OK(hipGetDeviceCount(&count));
    if (count <= 0) return fail(-3, "no HIP device available: the vegamx engine has no CPU fallback");
    REQUIRE(device >= 0 && device < count, "device index out of range");
    HIP_OK(hipSetDevice(device));
    auto* e = new vmx_engine();
    e->device = device;
    hipError_t err = hipStreamCreate(&e->stream);
    if (err != hipSuccess) { delete e; return fail(-2, std::string("hipStreamCreate: ") + hipGetErrorString(err)); }
    e->cur = e->stream;
    *out = e;
    return 0;
}

void vmx_destroy(vmx_engine* e)
{
    if (!e) return;
    if (e->trace_host && e->host_calls)
        std::fprintf(stderr, "[vegamx] host phases of vmx_eval over %lld calls: staging %.2f us, enqueue %.2f us, wait %.2f us\n",
                     (long long)e->host_calls, e->host_ns[0] / e->host_calls * 1e-3, e->host_ns[1] / e->host_calls * 1e-3,
                     e->host_ns[2] / e->host_calls * 1e-3);
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    delete e;
}

int vmx_set_template(vmx_engine* e, int32_t nk, const double* k, const double* pk_peak,
                     const double* pk_smooth, const double* pk_full, const double* delta2, int32_t n_mu)
{
    REQUIRE(e && !e->finalized, "engine is null or already finalized");
    REQUIRE(nk > 8 && n_mu > 0, "bad template sizes");
    REQUIRE(n_mu <= 2048, "num_bins_muk above 2048: the mu tables of the P(k,mu) kernels no longer fit a CU's LDS (the reference's default is 1000)");
    HIP_OK(hipSetDevice(e->device));
    e->nk = nk; e->nkp = vmx_pad(nk + e->pad_l + e->pad_r); e->n_mu = n_mu;      // (the rows of P_ell carry the FFTLog's power-law pads behind the samples)
    const int nkp = e->nkp;
    std::vector<double> buf(nkp, 0.0);
    e->h_k.assign(k, k + nk);
    std::copy(k, k + nk, buf.begin());
    for (int i = nk; i < nkp; ++i) buf[i] = k[nk - 1];
    if (e->k.upload(buf.data(), nkp)) return -2;
    std::vector<double> pl(3 * (size_t)nkp, 0.0);
    std::copy(pk_peak, pk_peak + nk, pl.begin());
    std::copy(pk_smooth, pk_smooth + nk, pl.begin() + nkp);
    std::copy(pk_full, pk_full + nk, pl.begin() + 2 * (size_t)nkp);
    if (e->pklin.upload(pl.data(), pl.size())) return -2;
    std::fill(buf.begin(), buf.end(), 0.0);
    std::copy(delta2, delta2 + nk, buf.begin());
    if (e->delta2.upload(buf.data(), nkp)) return -2;

    // mu grid: midpoint rule on [0, 1] (power_spectrum.py:76-77); Legendre weights
    // L_ell(mu) (2 ell + 1) / n_mu (pktoxi.py:37,55,138)
    // Node rule of the mu sums.  The reference sums P(k,mu) L_ell(mu) over n_mu midpoints (power_spectrum.py:76-77,
    // pktoxi.py:138).  For an integrand f that is smooth on [a, b] = [mu_lo, n_mu - mu_hi] / n_mu the midpoint sum over
    // that range is, by the Euler-Maclaurin formula (D = d/dmu, h = 1 / n_mu),
    //     sum_j f(mu_j) = (1/h) int_a^b f - (h/24) [Df(b) - Df(a)] + (7 h^3 / 5760) [D^3 f(b) - D^3 f(a)] - O(h^5 D^5 f),
    // so the engine keeps the first mu_lo and the last mu_hi midpoints (where sharp features of the model sit: the
    // Gaussian smoothing / non-linear broadening, the HCD exponential and mu^bv of the Arinyo term at 0; a smoothing that
    // is stronger across than along the line of sight at 1) and replaces the middle by two 32-point Gauss-Legendre
    // panels plus one-sided finite-difference stencils for the two derivative terms: 84 extra nodes with fixed weights.
    // Checked against the reference's own sums over the parameter ranges of the tests: <= 1.2e-13 of the largest
    // k^3 P_ell (tests/test_mu_quadrature.py); used for wavenumbers up to k_node_max (smooth enough for the bin size),
    // the plain midpoint loop above it and for the rarely used model options that are not smooth in mu.
    std::vector<double> node_mu, node_w;
    if (n_mu == 1000) {
        // (round 3: 48 + 48 kept midpoints, two 32-point panels and ONE nine-point one-sided stencil per end that carries the
        // first, third and fifth derivative terms of the Euler-Maclaurin formula at once - 178 nodes, 7e-14 of the largest
        // k^3 M_n over the guard's whole parameter box; round 2's 96 + 96 + 84 with five-point stencils reached 1e-13)
        const int lo = 48, hi = 48, panels = 2, ngl = 32, npts = 9;
        const double h = 1.0 / n_mu, a = lo * h, b = (n_mu - hi) * h, eps = 1e-3;
        std::vector<double> gx(ngl), gw(ngl);
        for (int i = 0; i < ngl; ++i) {            // Gauss-Legendre nodes by Newton's iteration on P_ngl
            double x = std::cos(M_PI * (i + 0.75) / (ngl + 0.5)), dp = 1.0;
            for (int it = 0; it < 100; ++it) {
                double p0 = 1.0, p1 = x;
                for (int n = 2; n <= ngl; ++n) { const double p2 = ((2 * n - 1) * x * p1 - (n - 1) * p0) / n; p0 = p1; p1 = p2; }
                dp = ngl * (x * p1 - p0) / (x * x - 1.0);
                const double dx = p1 / dp;
                x -= dx;
                if (std::fabs(dx) < 1e-16) break;
            }
            gx[i] = x; gw[i] = 2.0 / ((1.0 - x * x) * dp * dp);
        }
        for (int pnl = 0; pnl < panels; ++pnl) {
            const double pa = a + (b - a) * pnl / panels, pb = a + (b - a) * (pnl + 1) / panels;
            for (int i = 0; i < ngl; ++i) { node_mu.push_back(0.5 * (pb - pa) * gx[i] + 0.5 * (pa + pb)); node_w.push_back(0.5 * (pb - pa) * gw[i] / h); }
        }
        // Finite-difference weights of the derivatives 0 .. 5 at x = 0 on the points 0, 1, .. npts - 1 (Fornberg's recursion):
        // D^k f(x) ~ sum_i c[i][k] f(x + i e) / e^k
        constexpr int MD = 5;
        std::vector<std::vector<double>> c(npts, std::vector<double>(MD + 1, 0.0));
        {
            double c1 = 1.0, c4 = 0.0;
            c[0][0] = 1.0;
            for (int i = 1; i < npts; ++i) {
                const int mn = std::min(i, MD);
                double c2 = 1.0;
                const double c5 = c4;
                c4 = (double)i;
                for (int j = 0; j < i; ++j) {
                    const double c3 = (double)(i - j);
                    c2 *= c3;
                    if (j == i - 1) {
                        for (int k = mn; k >= 1; --k) c[i][k] = c1 * (k * c[i - 1][k - 1] - c5 * c[i - 1][k]) / c2;
                        c[i][0] = -c1 * c5 * c[i - 1][0] / c2;
                    }
                    for (int k = mn; k >= 1; --k) c[j][k] = (c4 * c[j][k] - k * c[j][k - 1]) / c3;
                    c[j][0] = c4 * c[j][0] / c3;
                }
                c1 = c2;
            }
        }
        // midpoint sum = (1/h) int - (h/24) [Df] + (7 h^3/5760) [D^3 f] - (31 h^5/967680) [D^5 f],  [g] = g(b) - g(a);
        // backward stencil at b: D^k f(b) ~ -sum_i c[i][k] f(b - i e) / e^k for odd k
        const double coef[3] = {-h / 24.0, 7.0 * h * h * h / 5760.0, -31.0 * h * h * h * h * h / 967680.0};
        const int order[3] = {1, 3, 5};
        for (int i = 0; i < npts; ++i) {
            double wb = 0.0, wa = 0.0;
            for (int q = 0; q < 3; ++q) {
                const double scaled = c[i][order[q]] / std::pow(eps, order[q]);
                wb += coef[q] * (-scaled);
                wa += -coef[q] * scaled;
            }
            node_mu.push_back(b - i * eps); node_w.push_back(wb);
            node_mu.push_back(a + i * eps); node_w.push_back(wa);
        }
        e->mu_lo = lo; e->mu_hi = hi;
    }
    e->n_extra = (int)node_mu.size();
    e->n_rows = n_mu + e->n_extra;
    if (e->node_w.upload(node_w.data(), node_w.size())) return -2;
    const int n_rows = e->n_rows;
    std::vector<double> mu(n_rows), sq(n_rows), lnm(n_rows), wl(4 * (size_t)n_mu);
    for (int j = n_mu; j < n_rows; ++j) {
        const double m = node_mu[j - n_mu];
        mu[j] = m; sq[j] = std::sqrt(std::max(1.0 - m * m, 0.0)); lnm[j] = std::log(m);
    }
    for (int j = 0; j < n_mu; ++j) {
        const double m = (j + 0.5) / n_mu, m2 = m * m;
        mu[j] = m; sq[j] = std::sqrt(1.0 - m2); lnm[j] = std::log(m);
        const double dmu = 1.0 / n_mu;
        wl[j] = dmu * 1.0 * 1.0;
        wl[n_mu + j] = dmu * (0.5 * (3.0 * m2 - 1.0)) * 5.0;
        wl[2 * (size_t)n_mu + j] = dmu * (0.125 * ((35.0 * m2 - 30.0) * m2 + 3.0)) * 9.0;
        wl[3 * (size_t)n_mu + j] = dmu * (0.0625 * (((231.0 * m2 - 315.0) * m2 + 105.0) * m2 - 5.0)) * 13.0;
    }
    e->h_mu = mu;
    {
        // the LDS image of k_pk_tab2's node tables: {mu^2, mu^4} of the midpoints, {mu, mu^2, mu^4, w} of the extra nodes
        std::vector<double> img(2 * (size_t)n_mu + 4 * (size_t)e->n_extra);
        for (int j = 0; j < n_mu; ++j) { const double m2 = mu[j] * mu[j]; img[2 * (size_t)j] = m2; img[2 * (size_t)j + 1] = m2 * m2; }
        for (int j = 0; j < e->n_extra; ++j) {
            const double m = mu[n_mu + j], m2 = m * m;
            double* q = &img[2 * (size_t)n_mu + 4 * (size_t)j];
            q[0] = m; q[1] = m2; q[2] = m2 * m2; q[3] = node_w[j];
        }
        if (e->mu_img.upload(img.data(), img.size())) return -2;
        // ... and of k_pk_w's: the same midpoints, {mu^2, mu^4, mu^6, w} of the extra nodes
        for (int j = 0; j < e->n_extra; ++j) {
            const double m = mu[n_mu + j], m2 = m * m;
            double* q = &img[2 * (size_t)n_mu + 4 * (size_t)j];
            q[0] = m2; q[1] = m2 * m2; q[2] = m2 * m2 * m2; q[3] = node_w[j];
        }
        if (e->mu_img_w.upload(img.data(), img.size())) return -2;
    }
    if (e->mu.upload(mu.data(), n_rows) || e->sq1mmu2.upload(sq.data(), n_rows) || e->lnmu.upload(lnm.data(), n_rows) || e->wl.upload(wl.data(), wl.size())) return -2;
    return 0;
}

int vmx_set_fftlog_padding(vmx_engine* e, int32_t n_left, int32_t n_right)
{
    REQUIRE(e && !e->finalized && e->nk == 0, "vmx_set_fftlog_padding (before vmx_set_template)");
    REQUIRE(n_left >= 0 && n_right >= 0 && n_left + n_right <= 16384, "vmx_set_fftlog_padding: pad lengths");
    e->pad_l = n_left; e->pad_r = n_right;
    return 0;
}

int vmx_set_fftlog(vmx_engine* e, int32_t ell_index, const double* op, int32_t n_coef, double x0, double h,
                   int32_t n_knots)
{
    REQUIRE(e && !e->finalized && e->nk > 0, "set the template first");
    REQUIRE(ell_index >= 0 && ell_index < VMX_MAX_ELL, "ell_index out of range");
    REQUIRE(n_coef == n_knots + 2 && n_knots >= 4, "n_coef must be n_knots + 2");
    HIP_OK(hipSetDevice(e->device));
    const int ncp = vmx_pad(n_coef);
    if (e->op.p == nullptr) {
        e->n_coef = n_coef; e->ncp = ncp; e->n_knots = n_knots;
        if (e->op.alloc((size_t)VMX_MAX_ELL * ncp * e->nkp, true)) return -2;
    }
    REQUIRE(n_coef == e->n_coef, "all multipoles must share the knot count");
    const size_t width = (size_t)(e->nk + e->pad_l + e->pad_r) * sizeof(double);       // samples [| left pads | right pads]
    HIP_OK(hipMemcpy2D(e->op.p + (size_t)ell_index * ncp * e->nkp, (size_t)e->nkp * sizeof(double), op, width, width, n_coef,
                       hipMemcpyHostToDevice));
    e->x0[ell_index] = x0; e->h[ell_index] = h; e->op_set[ell_index] = true;
    return 0;
}

int vmx_set_spline_extrapolation(vmx_engine* e, int32_t enabled)
{
    REQUIRE(e && !e->finalized, "vmx_set_spline_extrapolation");
    e->extrapolate = enabled != 0;
    return 0;
}

int vmx_set_fvoigt_table(vmx_engine* e, const double* x, const double* f, int32_t n)
{
    REQUIRE(e && !e->finalized && x && f && n >= 2, "vmx_set_fvoigt_table");
    for (int i = 1; i < n; ++i) REQUIRE(x[i] > x[i - 1], "the fvoigt table abscissae must be increasing");
    HIP_OK(hipSetDevice(e->device));
    if (e->fv_x.upload(x, n) || e->fv_f.upload(f, n)) return -2;
    e->fv_n = n;
    return 0;
}

int vmx_add_gk_table_mock(vmx_engine* e, double bin_size_rp, double bin_size_rt, double mock_size_rp, double mock_size_rt)
{
    if (!e || e->finalized || e->nk == 0) return fail(-1, "invalid argument: set the template first");
    for (size_t i = 0; i < e->gk_tables.size(); ++i) {
        const auto& g = e->gk_tables[i];
        if (g.rp == bin_size_rp && g.rt == bin_size_rt && g.mock_rp == mock_size_rp && g.mock_rt == mock_size_rt) return (int)i;
    }
    e->gk_tables.push_back({bin_size_rp, bin_size_rt, mock_size_rp, mock_size_rt});
    return (int)e->gk_tables.size() - 1;
}

int vmx_add_gk_table(vmx_engine* e, double bin_size_rp, double bin_size_rt)
{
    return vmx_add_gk_table_mock(e, bin_size_rp, bin_size_rt, 0.0, 0.0);
}

int vmx_add_pipeline(vmx_engine* e, const vmx_pipe_desc* desc, int32_t n, const double* r, const double* mu,
                     const double* z, const double* rel_z_evol, const double* xi_growth)
{
    if (!e || e->finalized || !desc || n <= 0) return fail(-1, "invalid argument: vmx_add_pipeline");
    if (desc->n_ell < 1 || desc->n_ell > VMX_MAX_ELL) return fail(-1, "invalid argument: n_ell");
    if (desc->gk_table >= (int)e->gk_tables.size()) return fail(-1, "invalid argument: gk_table id");
    if (desc->mock_los_slot >= 0 && !(desc->mock_los_size > 0.0)) return fail(-1, "invalid argument: mock_los_size");
    if (desc->n_smooth < 0 || desc->n_smooth > VMX_MAX_SMOOTH) return fail(-1, "invalid argument: n_smooth");
    PipeDev p{};
    p.d = *desc;
    p.poly_basis = -1; p.col = -1; p.poly_bins_off = -1; p.odd_dyn_off = -1;
    // The P(k) stage is symmetric in the two tracers: keep the Lya-like tracer first (and a discrete
    // tracer last) so that the specialised mu loops see one canonical order.
    if (!p.d.same_tracer && ((!p.d.tracer[0].is_lya && p.d.tracer[1].is_lya) ||
                             (p.d.tracer[0].is_lya == p.d.tracer[1].is_lya && p.d.tracer[0].discrete &&
                              !p.d.tracer[1].discrete))) {
        std::swap(p.d.tracer[0], p.d.tracer[1]);
        p.tracers_swapped = 1;
    }
    p.n = n;
    p.coord_off = (int64_t)e->h_r.size();
    e->h_r.insert(e->h_r.end(), r, r + n);
    e->h_mu_c.insert(e->h_mu_c.end(), mu, mu + n);
    e->h_z.insert(e->h_z.end(), z, z + n);
    e->h_relz.insert(e->h_relz.end(), rel_z_evol, rel_z_evol + n);
    for (int i = 0; i < n; ++i) { e->h_lnrelz.push_back(std::log(rel_z_evol[i])); e->h_lnrelz2.push_back(std::log(rel_z_evol[i])); }
    // extremes of |rp| and rt over the bins with r != 0: k_prologue bounds the rescaled separations with them
    p.rp_absmin = 1e300; p.rp_absmax = 0.0; p.rt_min = 1e300; p.rt_max = 0.0;
    for (int i = 0; i < n; ++i) {
        if (r[i] == 0.0) continue;
        const double rp = std::fabs(r[i] * mu[i]), rt = r[i] * std::sqrt(std::max(0.0, 1.0 - mu[i] * mu[i]));
        p.rp_absmin = std::min(p.rp_absmin, rp); p.rp_absmax = std::max(p.rp_absmax, rp);
        p.rt_min = std::min(p.rt_min, rt); p.rt_max = std::max(p.rt_max, rt);
    }
    if (p.rp_absmin > p.rp_absmax) { p.rp_absmin = 0.0; p.rt_min = 0.0; }       // no bin with r != 0
    e->h_growth.insert(e->h_growth.end(), xi_growth, xi_growth + n);
    e->pipes.push_back(p);
    return (int)e->pipes.size() - 1;
}

int vmx_pipeline_set_tracer_evolution(vmx_engine* e, int32_t pipeline, const double* rel_z_1, const double* rel_z_2, int32_t n)
{
    REQUIRE(e && !e->finalized && rel_z_1 && rel_z_2, "vmx_pipeline_set_tracer_evolution");
    REQUIRE(pipeline >= 0 && pipeline < (int)e->pipes.size(), "pipeline id");
    PipeDev& p = e->pipes[pipeline];
    REQUIRE(n == p.n, "tracer evolution size");
    REQUIRE(p.d.tracer[0].evol_kind == VMX_EVOL_STD && p.d.tracer[1].evol_kind == VMX_EVOL_STD,
            "Croom model is not supported with new bias evol");
    // vmx_add_pipeline may have swapped the tracers into its canonical order
    const bool swapped = p.tracers_swapped != 0;
    for (int i = 0; i < n; ++i) {
        e->h_lnrelz[(size_t)p.coord_off + i] = std::log((swapped ? rel_z_2 : rel_z_1)[i]);
        e->h_lnrelz2[(size_t)p.coord_off + i] = std::log((swapped ? rel_z_1 : rel_z_2)[i]);
    }
    p.split_evol = 1;
    return 0;
}

int vmx_pipeline_set_odd_terms(vmx_engine* e, int32_t pipeline, const double* coef, int32_t n_coef, double x0,
                               double h, int32_t relativistic, int32_t asymmetry, const int32_t* slots)
{
    REQUIRE(e && !e->finalized && coef && slots, "vmx_pipeline_set_odd_terms");
    REQUIRE(pipeline >= 0 && pipeline < (int)e->pipes.size(), "pipeline id");
    REQUIRE(n_coef >= 4 && h > 0.0, "spline description");
    PipeDev& p = e->pipes[pipeline];
    p.odd_rel = relativistic != 0; p.odd_asy = asymmetry != 0; p.odd_ncoef = n_coef;
    p.odd_x0 = x0; p.odd_h = h;
    for (int i = 0; i < 5; ++i) {
        const bool needed = (i < 2) ? p.odd_rel : p.odd_asy;
        REQUIRE(!needed || slots[i] >= 0, "amplitude slot of an odd-multipole term");
        p.odd_slot[i] = slots[i];
    }
    p.odd_off = (int64_t)e->h_odd.size();
    p.odd_dyn_off = -1; p.odd_dyn_ld = 0;
    e->h_odd.insert(e->h_odd.end(), coef, coef + (size_t)4 * n_coef);
    return 0;
}

int vmx_pipeline_set_odd_operator(vmx_engine* e, int32_t pipeline, const double* op, int32_t n_coef, int32_t nk)
{
    REQUIRE(e && !e->finalized && op, "vmx_pipeline_set_odd_operator (before vmx_finalize)");
    REQUIRE(pipeline >= 0 && pipeline < (int)e->pipes.size(), "pipeline id");
    PipeDev& p = e->pipes[pipeline];
    REQUIRE((p.odd_rel || p.odd_asy) && n_coef == p.odd_ncoef && nk == e->nk, "vmx_pipeline_set_odd_operator: after vmx_pipeline_set_odd_terms, same shapes");
    // rows [4 n_coef] of nkp doubles (zero tails): the operand layout of the product kernels
    const size_t rows = (size_t)4 * n_coef;
    e->odd_op_off[pipeline] = (int64_t)e->h_odd_op.size();
    e->h_odd_op.resize(e->h_odd_op.size() + rows * e->nkp, 0.0);
    double* dst = e->h_odd_op.data() + e->odd_op_off[pipeline];
    for (size_t r = 0; r < rows; ++r) std::copy(op + r * nk, op + (r + 1) * nk, dst + r * e->nkp);
    return 0;
}

int vmx_set_shotnoise_table(vmx_engine* e, const double* a, int32_t n, double tau0, double dtau)
{
    REQUIRE(e && !e->finalized && a && n >= 2 && dtau > 0.0, "vmx_set_shotnoise_table");
    HIP_OK(hipSetDevice(e->device));
    if (e->sn_a.upload(a, n)) return -2;
    e->sn_n = n; e->sn_tau0 = tau0; e->sn_dtau = dtau;
    return 0;
}

int vmx_item_set_additive_template(vmx_engine* e, int32_t item, const double* vec, int32_t n_model, int32_t slot,
                                   double default_amp)
{
    REQUIRE(e && !e->finalized && vec, "vmx_item_set_additive_template");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(n_model == it->dev.d.n_model, "template size");
    HIP_OK(hipSetDevice(e->device));
    if (it->add_vec.upload(vec, n_model)) return -2;
    it->dev.add_vec = it->add_vec.p; it->dev.add_slot = slot; it->dev.add_default = default_amp;
    return 0;
}

int vmx_add_item(vmx_engine* e, const vmx_item_desc* desc)
{
    if (!e || e->finalized || !desc) return fail(-1, "invalid argument: vmx_add_item");
    const int np = (int)e->pipes.size();
    if (desc->pipe_peak < 0 || desc->pipe_peak >= np || desc->pipe_smooth < 0 || desc->pipe_smooth >= np)
        return fail(-1, "invalid argument: pipeline id");
    if (e->pipes[desc->pipe_peak].n != desc->n_model || e->pipes[desc->pipe_smooth].n != desc->n_model)
        return fail(-1, "invalid argument: core pipelines must have n_model bins");
    if (desc->bao_amp_slot < 0) return fail(-1, "invalid argument: bao_amp slot");
    auto* it = new ItemHost();
    it->dev.d = *desc;
    it->dev.n_model_pad = vmx_pad(desc->n_model);
    it->dev.n_dist_pad = vmx_pad(desc->n_dist);
    it->dev.metal_begin = (int)e->metals.size();
    it->dev.add_vec = nullptr; it->dev.add_slot = -1; it->dev.add_default = 0.0;
    e->items.push_back(it);
    return (int)e->items.size() - 1;
}

int vmx_item_add_metal(vmx_engine* e, int32_t item, const vmx_metal_desc* desc)
{
    REQUIRE(e && !e->finalized && desc, "vmx_item_add_metal");
    REQUIRE(item == (int)e->items.size() - 1, "metals must be added to the most recent item");
    REQUIRE(desc->pipeline >= -1 && desc->pipeline < (int)e->pipes.size(), "metal pipeline id");
    ItemHost* it = e->items[item];
    REQUIRE((int)it->metals.size() < VMX_MAX_METALS, "too many metals");
    auto* m = new MetalHost();
    m->dev.d = *desc;
    m->dev.mat_off = -1;
    m->dev.svec = nullptr;
    m->dev.basis = nullptr;
    it->metals.push_back(m);
    e->metals.push_back(m);
    it->dev.n_metals = (int)it->metals.size();
    return (int)it->metals.size() - 1;
}

int vmx_item_set_metal_static(vmx_engine* e, int32_t item, int32_t index, const double* xi, int32_t n_model)
{
    REQUIRE(e && !e->finalized && xi, "vmx_item_set_metal_static");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(index >= 0 && index < (int)it->metals.size(), "metal index");
    REQUIRE(n_model == it->dev.d.n_model, "static metal correlation size");
    MetalHost* m = it->metals[index];
    REQUIRE(m->dev.d.pipeline == -1, "a static correlation replaces the pipeline: add the metal with pipeline = -1");
    HIP_OK(hipSetDevice(e->device));
    if (m->svec.upload(xi, (size_t)n_model)) return -2;
    m->dev.svec = m->svec.p;
    return 0;
}

int vmx_item_set_metal_basis(vmx_engine* e, int32_t item, int32_t index, const double* basis, int32_t n_model)
{
    REQUIRE(e && !e->finalized && basis, "vmx_item_set_metal_basis");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(index >= 0 && index < (int)it->metals.size(), "metal index");
    REQUIRE(n_model == it->dev.d.n_model, "metal basis size");
    MetalHost* m = it->metals[index];
    REQUIRE(m->dev.d.pipeline == -1, "a static basis replaces the pipeline: add the metal with pipeline = -1");
    HIP_OK(hipSetDevice(e->device));
    if (upload_padded(m->basis, basis, 3, n_model, vmx_pad(n_model))) return -2;
    m->dev.basis = m->basis.p;
    return 0;
}

int vmx_set_parameter_transform(vmx_engine* e, const double* scale, const double* shift)
{
    REQUIRE(e && e->finalized, "vmx_set_parameter_transform");
    drop_lane(e);
    REQUIRE((scale == nullptr) == (shift == nullptr), "scale and shift are given together");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->blind_scale.clear(); e->blind_shift.clear();
    if (!scale) return 0;
    e->blind_scale.assign(scale, scale + e->n_params);
    e->blind_shift.assign(shift, shift + e->n_params);
    if (e->d_blind.n < (size_t)2 * e->n_params && e->d_blind.alloc((size_t)2 * e->n_params)) return -2;
    HIP_OK(hipMemcpy(e->d_blind.p, scale, (size_t)e->n_params * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(e->d_blind.p + e->n_params, shift, (size_t)e->n_params * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int vmx_set_metal_beta_override(vmx_engine* e, int32_t enabled, double beta)
{
    REQUIRE(e && e->finalized, "vmx_set_metal_beta_override");
    drop_lane(e);
    e->dev.beta_override_on = enabled ? 1 : 0;
    e->dev.beta_override = beta;
    // captured graphs hold the previous value: drop them
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second);
    e->graphs.clear();
    return 0;
}

int vmx_item_add_broadband(vmx_engine* e, int32_t item, int32_t position, int32_t func, int32_t n_coef,
                           const int32_t* slots, const double* basis, int32_t n)
{
    REQUIRE(e && !e->finalized, "vmx_item_add_broadband");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    REQUIRE(position >= 0 && position < 4, "position");
    ItemHost* it = e->items[item];
    REQUIRE(it->dev.n_bb[position] < VMX_MAX_BB, "too many broadband terms");
    const bool pre = position == VMX_BB_PRE_MUL || position == VMX_BB_PRE_ADD;
    REQUIRE(n == (pre ? it->dev.d.n_model : it->dev.d.n_dist), "broadband basis size");
    REQUIRE(func == VMX_BB_POLY || func == VMX_BB_SKY, "broadband func");
    REQUIRE(n_coef > 0 && n_coef <= 16, "at most 16 coefficients per broadband term");
    REQUIRE(func != VMX_BB_SKY || n_coef == 2, "sky term takes {scale, sigma}");
    BBTermDev& t = it->dev.bb[position][it->dev.n_bb[position]++];
    t.func = func; t.n_coef = n_coef;
    for (int i = 0; i < n_coef; ++i) { REQUIRE(slots[i] >= 0, "broadband slot"); t.slot[i] = slots[i]; }
    t.basis_off = (int64_t)e->h_bb.size();
    e->h_bb.insert(e->h_bb.end(), basis, basis + (size_t)n_coef * n);
    return 0;
}

int vmx_item_set_matrix(vmx_engine* e, int32_t item, int32_t kind, int32_t index, int32_t rows, int32_t cols,
                        const double* dense)
{
    REQUIRE(e && dense, "vmx_item_set_matrix");
    drop_lane(e);
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    HIP_OK(hipSetDevice(e->device));
    ItemHost* it = e->items[item];
    if (kind == VMX_MAT_DISTORTION) {
        REQUIRE(!e->finalized, "distortion matrix must be set before vmx_finalize");
        REQUIRE(rows == it->dev.d.n_dist && cols == it->dev.d.n_model, "distortion matrix shape");
        REQUIRE(!it->has_csr, "the item already has a CSR distortion matrix");
        if (upload_padded(it->dm, dense, rows, cols, it->dev.n_model_pad)) return -2;
        it->has_dm = true;
    } else if (kind == VMX_MAT_INVCOV) {
        REQUIRE(it->has_mask, "set the mask before the inverse covariance");
        REQUIRE(rows == it->dev.n_masked && cols == rows, "inverse covariance shape");
        const std::vector<double> half = half_form(dense, rows);
        e->quad_mat_dirty = true;
        if (e->finalized) {
            REQUIRE(it->has_cinv, "an identity inverse covariance cannot be replaced after vmx_finalize");
            HIP_OK(hipStreamSynchronize(e->stream));
            HIP_OK(hipMemcpy2D(it->cinv.p, (size_t)it->dev.n_masked_pad * sizeof(double), half.data(),
                               (size_t)cols * sizeof(double), (size_t)cols * sizeof(double), rows,
                               hipMemcpyHostToDevice));
        } else {
            if (upload_padded(it->cinv, half.data(), rows, cols, it->dev.n_masked_pad)) return -2;
            it->has_cinv = true;
        }
    } else if (kind == VMX_MAT_METAL) {
        REQUIRE(!e->finalized, "metal matrices must be set before vmx_finalize");
        REQUIRE(index >= 0 && index < (int)it->metals.size(), "metal index");
        MetalHost* m = it->metals[index];
        REQUIRE(m->dev.d.pipeline >= 0, "a static metal correlation takes no matrix");
        REQUIRE(rows == it->dev.d.n_model && cols == e->pipes[m->dev.d.pipeline].n, "metal matrix shape");
        if (upload_padded(m->mat, dense, rows, cols, vmx_pad(cols))) return -2;
        m->rows = rows; m->cols = cols;
        m->dev.mat_off = 0;
        m->dev.mat_ld = vmx_pad(cols);
    } else return fail(-1, "invalid argument: matrix kind");
    return 0;
}

int vmx_item_set_matrix_csr(vmx_engine* e, int32_t item, int32_t rows, int32_t cols, const int64_t* indptr,
                            const int32_t* indices, const double* values)
{
    REQUIRE(e && !e->finalized && indptr && indices && values, "vmx_item_set_matrix_csr (before vmx_finalize)");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(rows == it->dev.d.n_dist && cols == it->dev.d.n_model, "distortion matrix shape");
    REQUIRE(!it->has_dm, "the item already has a dense distortion matrix");
    // validity and canonical form (vmx_plan.h: the set-up of the quadratic form scatters the rows - last write wins - where
    // the product adds them, so duplicates would disagree)
    if (const char* why = vmx_plan::csr_problem(rows, cols, indptr, indices); why[0]) return fail(-1, std::string("invalid argument: CSR matrix: ") + why);
    const int64_t nnz = indptr[rows];
    HIP_OK(hipSetDevice(e->device));
    if (it->csr_ptr.upload(indptr, (size_t)rows + 1) || it->csr_idx.upload(indices, (size_t)std::max<int64_t>(nnz, 1)) ||
        it->csr_val.upload(values, (size_t)std::max<int64_t>(nnz, 1))) return -2;
    it->csr_nnz = nnz; it->has_csr = true;
    return 0;
}

int vmx_item_set_metal_kron(vmx_engine* e, int32_t item, int32_t index, const double* a_rp, int32_t n_rp,
                            const double* b_rt, int32_t n_rt)
{
    REQUIRE(e && !e->finalized && a_rp, "vmx_item_set_metal_kron (before vmx_finalize)");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(index >= 0 && index < (int)it->metals.size(), "metal index");
    MetalHost* m = it->metals[index];
    REQUIRE(m->dev.d.pipeline >= 0, "a static metal correlation takes no matrix");
    REQUIRE(n_rp > 0 && n_rt > 0 && n_rp * n_rt == it->dev.d.n_model && n_rp * n_rt == e->pipes[m->dev.d.pipeline].n,
            "Kronecker factors: n_rp * n_rt must be the item's (and the pair's) bin count");
    REQUIRE((size_t)(2 * n_rp * n_rt + std::max(n_rp * n_rp, n_rt * n_rt)) * sizeof(double) <= 160 * 1024,
            "Kronecker factors too large for the LDS kernel: pass the dense matrix instead");
    HIP_OK(hipSetDevice(e->device));
    if (m->kron_a.upload(a_rp, (size_t)n_rp * n_rp)) return -2;
    if (b_rt && m->kron_b.upload(b_rt, (size_t)n_rt * n_rt)) return -2;
    m->dev.kron_a = m->kron_a.p; m->dev.kron_b = b_rt ? m->kron_b.p : nullptr;
    m->dev.kron_nrp = n_rp; m->dev.kron_nrt = n_rt;
    m->dev.mat_off = 0;                 // "has a matrix": its product lives in the metal-product buffer
    return 0;
}

int vmx_item_set_mask(vmx_engine* e, int32_t item, const int32_t* idx, int32_t n_masked)
{
    REQUIRE(e && !e->finalized && idx, "vmx_item_set_mask");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(n_masked > 0 && n_masked <= it->dev.d.n_dist, "mask size");
    HIP_OK(hipSetDevice(e->device));
    std::vector<int32_t> inv(it->dev.d.n_dist, -1);
    for (int i = 0; i < n_masked; ++i) {
        REQUIRE(idx[i] >= 0 && idx[i] < it->dev.d.n_dist && inv[idx[i]] < 0, "mask index");
        inv[idx[i]] = i;
    }
    it->mask_idx.assign(idx, idx + n_masked);
    if (it->inv_mask.upload(inv.data(), inv.size())) return -2;
    it->dev.n_masked = n_masked;
    it->dev.n_masked_pad = vmx_pad(n_masked);
    it->has_mask = true;
    return 0;
}

int vmx_item_set_data(vmx_engine* e, int32_t item, const double* masked_data, int32_t n_masked)
{
    REQUIRE(e && masked_data, "vmx_item_set_data");
    drop_lane(e);
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(it->has_mask && n_masked == it->dev.n_masked, "data size must match the mask");
    HIP_OK(hipSetDevice(e->device));
    e->quad_lin_dirty = true;
    if (it->has_data) {
        HIP_OK(hipStreamSynchronize(e->stream));
        HIP_OK(hipMemcpy(it->data.p, masked_data, (size_t)n_masked * sizeof(double), hipMemcpyHostToDevice));
    } else {
        if (it->data.upload(masked_data, n_masked)) return -2;
        it->has_data = true;
    }
    return 0;
}

int vmx_item_set_mock_pool(vmx_engine* e, int32_t item, const double* pool, int32_t n_mocks, int32_t n_masked)
{
    REQUIRE(e && e->finalized && pool, "vmx_item_set_mock_pool (after vmx_finalize)");
    drop_lane(e);
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(n_mocks > 0 && n_masked == it->dev.n_masked, "mock pool shape");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (it->mock_pool.upload(pool, (size_t)n_mocks * n_masked)) return -2;
    it->n_mocks = n_mocks;
    e->quad_lin_dirty = true;
    it->dev.mock_pool = it->mock_pool.p;
    // the item table lives in device memory: patch this item's entry
    HIP_OK(hipMemcpy(e->d_items.p + item, &it->dev, sizeof(ItemDev), hipMemcpyHostToDevice));
    return 0;
}

int vmx_item_set_mock_factor(vmx_engine* e, int32_t item, const double* chol, const double* fiducial, int32_t n_masked)
{
    REQUIRE(e && e->finalized && chol && fiducial, "vmx_item_set_mock_factor (after vmx_finalize)");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(n_masked == it->dev.n_masked, "mock factor: [n_masked][n_masked]");
    HIP_OK(hipSetDevice(e->device));
    wait_lane(e);
    HIP_OK(hipStreamSynchronize(e->stream));
    const int nmp = it->dev.n_masked_pad;
    if (upload_padded(it->mc_chol, chol, n_masked, n_masked, nmp) || upload_padded(it->mc_fid, fiducial, 1, n_masked, nmp)) return -2;
    it->has_factor = true;
    return 0;
}

int vmx_item_get_mock_pool(vmx_engine* e, int32_t item, double* pool, int32_t n_mocks, int32_t n_masked)
{
    REQUIRE(e && e->finalized && pool, "vmx_item_get_mock_pool");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(n_mocks > 0 && n_mocks <= it->n_mocks && n_masked == it->dev.n_masked && it->mock_pool.p, "mock pool shape");
    HIP_OK(hipSetDevice(e->device));
    wait_lane(e);
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(pool, it->mock_pool.p, (size_t)n_mocks * n_masked * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int vmx_host_alloc(vmx_engine* e, void** out, int64_t bytes)
{
    REQUIRE(e && out && bytes > 0, "vmx_host_alloc");
    HIP_OK(hipSetDevice(e->device));
    void* p = nullptr;
    HIP_OK(hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault));
    e->host_allocs.push_back(p);
    *out = p;
    return 0;
}

int vmx_host_free(vmx_engine* e, void* p)
{
    REQUIRE(e && p, "vmx_host_free");
    auto it = std::find(e->host_allocs.begin(), e->host_allocs.end(), p);
    REQUIRE(it != e->host_allocs.end(), "vmx_host_free: not a buffer of vmx_host_alloc on this engine");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));       // (no copy out of it may still be queued)
    wait_lane(e);
    e->host_allocs.erase(it);
    HIP_OK(hipHostFree(p));
    return 0;
}

int vmx_set_mock_index(vmx_engine* e, const int32_t* index, int32_t B)
{
    REQUIRE(e && e->finalized, "vmx_set_mock_index (after vmx_finalize)");
    drop_lane(e);
    REQUIRE(B >= 0 && B <= e->max_batch, "batch exceeds max_batch");
    HIP_OK(hipSetDevice(e->device));
    for (int b = 0; b < e->max_batch; ++b) e->h_mock_index[b] = -1;
    if (index) {
        for (int b = 0; b < B; ++b) {
            for (auto* it : e->items) REQUIRE(index[b] < 0 || index[b] < it->n_mocks, "mock index exceeds the pool");
            e->h_mock_index[b] = index[b];
        }
    }
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy(e->mock_index.p, e->h_mock_index.data(), (size_t)e->max_batch * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

int vmx_set_global_invcov(vmx_engine* e, const double* invcov, int32_t n)
{
    REQUIRE(e && invcov && n > 0, "vmx_set_global_invcov");
    drop_lane(e);
    HIP_OK(hipSetDevice(e->device));
    const std::vector<double> half = half_form(invcov, n);
    if (e->finalized) {
        REQUIRE(e->g_n == n, "global inverse covariance size changed");
        HIP_OK(hipStreamSynchronize(e->stream));
        HIP_OK(hipMemcpy2D(e->gcinv.p, (size_t)e->g_ld * sizeof(double), half.data(), (size_t)n * sizeof(double),
                           (size_t)n * sizeof(double), n, hipMemcpyHostToDevice));
        return 0;
    }
    e->g_n = n; e->g_ld = vmx_pad(n);
    return upload_padded(e->gcinv, half.data(), n, n, e->g_ld);
}

int vmx_add_prior(vmx_engine* e, int32_t slot, double mean, double sigma)
{
    REQUIRE(e && !e->finalized && slot >= 0 && sigma != 0.0, "vmx_add_prior");
    e->prior_slot.push_back(slot); e->prior_mean.push_back(mean); e->prior_sigma.push_back(sigma);
    return 0;
}

static int poly_basis_build(vmx_engine* e);
static int xi_sum_plan(vmx_engine* e);

int vmx_finalize(vmx_engine* e, int32_t n_params, int32_t max_batch)
{
    REQUIRE(e && !e->finalized, "vmx_finalize");
    REQUIRE(n_params > 0 && max_batch > 0, "n_params / max_batch");
    REQUIRE(e->nk > 0 && !e->pipes.empty() && !e->items.empty(), "template, pipelines and items are required");
    REQUIRE(e->items.size() <= 16, "at most 16 correlation items");
    REQUIRE(e->pipes.size() * sizeof(PipeDev) <= 48 * 1024, "too many pipelines for the prologue's descriptor stage");
    for (int i = 0; i < VMX_MAX_ELL; ++i) REQUIRE(e->op_set[i], "all four FFTLog operators are required");
    HIP_OK(hipSetDevice(e->device));
    const int Bm = max_batch;
    e->n_params = n_params; e->max_batch = Bm;
    // split-K slabs are re-read by the consumer kernels: at most 1024 walker rows of slabs per product (measured
    // best in the full chain, where the items overlap on separate streams and fill the chip anyway)
    e->slab_rows = Bm > 1024 ? Bm : 1024;
    e->fact_slab_rows = std::max(1024, std::min(4 * Bm, 8192));
    if (getenv("VMX_NO_GRAPH")) e->use_graphs = false;
    if (getenv("VMX_TRACE_HOST")) e->trace_host = true;
    {
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        e->quad_blocks = 2 * cus;
    }
    if (getenv("VMX_NO_TAB2")) e->no_tab2 = true;
    if (getenv("VMX_NO_CINV_TAPE")) e->no_cinv_tape = true;
    if (getenv("VMX_NO_PK_W")) e->no_pk_w = true;
    if (getenv("VMX_NO_HOST_REDUCE")) e->no_host_reduce = true;
    if (getenv("VMX_NO_XI_LEAN")) e->xi_lean = false;
    if (getenv("VMX_NO_XI_SUMS")) e->xi_sums = false;

    // every slot must index a theta column and the combinations the kernels rely on must be present
    auto slot_ok = [&](int s) { return s < n_params; };
    for (auto& p : e->pipes) {
        const vmx_pipe_desc& d = p.d;
        const int slots[] = {d.tracer[0].bias_slot, d.tracer[0].bias_eta_slot, d.tracer[0].beta_slot,
                             d.tracer[0].vd_sigma_slot, d.tracer[0].alpha_slot, d.tracer[1].bias_slot,
                             d.tracer[1].bias_eta_slot, d.tracer[1].beta_slot, d.tracer[1].vd_sigma_slot,
                             d.tracer[1].alpha_slot, d.growth_rate_slot, d.bias_gamma_slot, d.bias_prim_slot,
                             d.lambda_uv_slot, d.bias_gamma_e_slot, d.lambda_heii_slot, d.bias_hcd_slot,
                             d.beta_hcd_slot, d.l0_hcd_slot, d.sigma_nl_par_slot, d.sigma_nl_per_slot,
                             d.exp_par_slot, d.exp_per_slot, d.scale_slot[0], d.scale_slot[1], d.drp_slot,
                             d.croom_slot[0], d.croom_slot[1], d.mock_los_slot};
        for (int s : slots) REQUIRE(slot_ok(s), "pipeline slot exceeds n_params");
        for (int i = 0; i < 6; ++i) REQUIRE(slot_ok(d.arinyo_slot[i]), "arinyo slot exceeds n_params");
        for (int i = 0; i < d.n_smooth; ++i)
            REQUIRE(d.smooth_par_slot[i] >= 0 && d.smooth_per_slot[i] >= 0 && slot_ok(d.smooth_par_slot[i]) &&
                    slot_ok(d.smooth_per_slot[i]), "smoothing slot");
        for (int q = 0; q < 2; ++q) {
            const vmx_tracer& t = d.tracer[q];
            const int have = (t.bias_slot >= 0) + (t.bias_eta_slot >= 0) + (t.beta_slot >= 0);
            REQUIRE(have >= 2, "each tracer needs two of (bias, bias_eta, beta)");
            REQUIRE(t.evol_kind == VMX_EVOL_CROOM ? (d.croom_slot[0] >= 0 && d.croom_slot[1] >= 0) : t.alpha_slot >= 0,
                    "bias-evolution slot");
            REQUIRE(d.vd_kind == VMX_VD_NONE || !t.discrete || t.vd_sigma_slot >= 0, "velocity dispersion slot");
        }
        REQUIRE(!d.uvb || (d.bias_gamma_slot >= 0 && d.bias_prim_slot >= 0 && d.lambda_uv_slot >= 0), "UVB slots");
        REQUIRE(!d.heii || (d.bias_gamma_e_slot >= 0 && d.bias_prim_slot >= 0 && d.lambda_heii_slot >= 0), "HeII slots");
        REQUIRE(d.hcd_model == VMX_HCD_NONE || (d.bias_hcd_slot >= 0 && d.beta_hcd_slot >= 0), "HCD slots");
        REQUIRE(d.hcd_model != VMX_HCD_ROGERS || d.l0_hcd_slot >= 0, "L0_hcd slot");
        REQUIRE(d.hcd_model != VMX_HCD_FVOIGT || e->fv_n >= 2, "model-hcd = fvoigt needs vmx_set_fvoigt_table");
        REQUIRE(d.nl_model != VMX_NL_ARINYO || (d.arinyo_slot[0] >= 0 && d.arinyo_slot[2] >= 0 && d.arinyo_slot[3] >= 0 &&
                                                d.arinyo_slot[4] >= 0 && d.arinyo_slot[5] >= 0), "Arinyo slots");
        REQUIRE(!d.peak_nl || d.sigma_nl_par_slot >= 0 || d.sigma_nl_per_slot >= 0, "sigmaNL slots");
        REQUIRE((d.exp_par_slot >= 0) == (d.exp_per_slot >= 0), "exp smoothing slots");
        REQUIRE(d.scale_mode == VMX_SCALE_UNIT || (d.scale_slot[0] >= 0 && d.scale_slot[1] >= 0), "scale slots");
        if (d.radiation) for (int i = 0; i < 4; ++i) REQUIRE(d.rad_slot[i] >= 0 && slot_ok(d.rad_slot[i]), "radiation slots");
        if (d.uv_shotnoise) {
            for (int i = 0; i < 3; ++i) REQUIRE(d.uvsn_slot[i] >= 0 && slot_ok(d.uvsn_slot[i]), "UV shot-noise slots");
            REQUIRE(e->sn_n >= 2, "UV shot noise needs vmx_set_shotnoise_table");
        }
    }
    for (auto* m : e->metals) {
        const vmx_metal_desc& d = m->dev.d;
        for (int q = 0; q < 2; ++q) {
            const vmx_tracer& t = d.tracer[q];
            REQUIRE(slot_ok(t.bias_slot) && slot_ok(t.bias_eta_slot) && slot_ok(t.beta_slot), "metal slot exceeds n_params");
            REQUIRE(!d.apply_bias || (t.bias_slot >= 0) + (t.bias_eta_slot >= 0) + (t.beta_slot >= 0) >= 2,
                    "each metal tracer needs two of (bias, bias_eta, beta)");
        }
        REQUIRE(slot_ok(d.growth_rate_slot) && slot_ok(d.extra_bias_slot), "metal slot exceeds n_params");
    }
    for (auto* it : e->items) {
        REQUIRE(slot_ok(it->dev.d.bao_amp_slot), "bao_amp slot exceeds n_params");
        REQUIRE(slot_ok(it->dev.add_slot), "additive-template slot exceeds n_params");
        for (int pos = 0; pos < 4; ++pos)
            for (int q = 0; q < it->dev.n_bb[pos]; ++q)
                for (int c = 0; c < it->dev.bb[pos][q].n_coef; ++c)
                    REQUIRE(slot_ok(it->dev.bb[pos][q].slot[c]), "broadband slot exceeds n_params");
    }

    // G(k) tables
    const size_t gk_stride = (size_t)e->n_rows * e->nkp;
    if (!e->gk_tables.empty()) {
        // (+ 16 rows: k_pk_w requests its table rows four steps ahead without checking for the end)
        if (e->gk.alloc(gk_stride * e->gk_tables.size() + (size_t)16 * e->nkp, true)) return -2;
        for (size_t t = 0; t < e->gk_tables.size(); ++t) {
            dim3 grid((e->nkp + 255) / 256, e->n_rows), block(256);
            hipLaunchKernelGGL(k_gk_table, grid, block, 0, e->stream, e->gk.p + t * gk_stride, e->k.p, e->mu.p,
                               e->nk, e->nkp, e->n_rows, e->gk_tables[t].rp, e->gk_tables[t].rt, e->gk_tables[t].mock_rp, e->gk_tables[t].mock_rt);
        }
        HIP_OK(hipGetLastError());
    }
    {
        const size_t nt = e->gk_tables.size();
        if (e->gk_mom.alloc((nt + 1) * 6 * (size_t)e->nkp, true)) return -2;
        for (size_t t = 0; t <= nt; ++t)
            hipLaunchKernelGGL(k_gk_moments, dim3((e->nkp + 63) / 64), dim3(64), 0, e->stream,
                               e->gk_mom.p + t * 6 * (size_t)e->nkp, t < nt ? e->gk.p + t * gk_stride : nullptr,
                               e->mu.p, e->nkp, e->n_mu);
        HIP_OK(hipGetLastError());
    }

    // coordinates and pipelines
    {
        std::vector<double> rp(e->h_r.size()), rt(e->h_r.size());
        for (size_t i = 0; i < rp.size(); ++i) {
            const double r = e->h_r[i], m = e->h_mu_c[i];
            rp[i] = r * m; rt[i] = r * std::sqrt(1.0 - m * m);        // correlation_func.py:216-217
        }
        if (e->crp.upload(rp.data(), rp.size()) || e->crt.upload(rt.data(), rt.size())) return -2;
    }
    if (e->cr.upload(e->h_r.data(), e->h_r.size()) || e->cmu.upload(e->h_mu_c.data(), e->h_mu_c.size()) ||
        e->cz.upload(e->h_z.data(), e->h_z.size()) || e->crelz.upload(e->h_relz.data(), e->h_relz.size()) ||
        e->clnrelz.upload(e->h_lnrelz.data(), e->h_lnrelz.size()) || e->clnrelz2.upload(e->h_lnrelz2.data(), e->h_lnrelz2.size()) ||
        e->cgrowth.upload(e->h_growth.data(), e->h_growth.size())) return -2;
    int64_t xi_off = 0;
    for (auto& p : e->pipes) { p.n_pad = vmx_pad(p.n); p.xi_off = xi_off; xi_off += (int64_t)Bm * p.n_pad; }
    e->xi_total = xi_off;
    const int n_pipe = (int)e->pipes.size();
    {
        // direct_pk with odd-multipole terms: where a pipeline's per-walker coefficient rows live (EngineDev::odd_dyn)
        int64_t off = 0;
        for (auto& kv : e->odd_op_off) {
            PipeDev& p = e->pipes[kv.first];
            p.odd_dyn_ld = vmx_pad(4 * p.odd_ncoef);
            p.odd_dyn_off = off;
            off += (int64_t)Bm * p.odd_dyn_ld;
        }
    }
    if (e->d_pipes.upload(e->pipes.data(), e->pipes.size())) return -2;

    // P(k,mu) work groups: an item's peak component rides along with its smooth component when the two
    // differ only by the peak broadening
    {
        std::vector<char> taken(e->pipes.size(), 0);
        e->pk_groups.clear();
        for (auto* it : e->items) {
            const int ps = it->dev.d.pipe_smooth, pk = it->dev.d.pipe_peak;
            if (ps != pk && !taken[ps] && !taken[pk] && pk_stage_compatible(e->pipes[ps].d, e->pipes[pk].d)) {
                e->pk_groups.push_back({ps, pk, 0, 0, 0, -1});
                taken[ps] = taken[pk] = 1;
            }
        }
        // unpaired pipelines: those without any amplitude-dependent mu structure (no HCD / UV-in-the-loop / NL)
        // that share the amplitude-free factor W are evaluated by one mu loop per shared-W group
        e->pk_members.clear();
        e->pk_poly.clear();
        for (int p = 0; p < (int)e->pipes.size(); ++p) {
            if (taken[p]) continue;
            const int variant = pk_variant(e->pipes[p].d, false);
            const bool plain = variant == PKV_PLAIN_SAME || variant == PKV_PLAIN_PAIR || variant == PKV_PLAIN_PAIR_VD;
            if (variant == PKV_POLY) { e->pk_poly.push_back(p); taken[p] = 1; continue; }
            if (!plain) { e->pk_groups.push_back({p, -1, variant, 0, 0, -1}); taken[p] = 1; continue; }
            PkGroup g{p, -1, PKV_SHARED_W, 0, (int32_t)e->pk_members.size(), -1};
            for (int q = p; q < (int)e->pipes.size(); ++q) {
                if (taken[q]) continue;
                const int vq = pk_variant(e->pipes[q].d, false);
                const bool plain_q = vq == PKV_PLAIN_SAME || vq == PKV_PLAIN_PAIR || vq == PKV_PLAIN_PAIR_VD;
                if (plain_q && w_stage_compatible(e->pipes[p].d, e->pipes[q].d)) {
                    e->pk_members.push_back(q); taken[q] = 1; ++g.n_members;
                }
            }
            e->pk_groups.push_back(g);
        }
        e->n_xtab = 0;
        e->const_slots.clear();
        e->const_slots2.clear();
        auto add_slot = [](std::vector<int32_t>& v, int slot) {
            if (slot >= 0 && std::find(v.begin(), v.end(), slot) == v.end()) v.push_back(slot);
        };
        for (auto& g : e->pk_groups) {
            if (g.peak_partner >= 0) g.variant = pk_variant(e->pipes[g.pipe].d, true);
            // core groups with the Arinyo term can run against per-batch tables (EngineDev::xtab_level)
            if (g.variant == PKV_AUTO_CORE || g.variant == PKV_CROSS_CORE) {
                g.xtab = e->n_xtab++;
                for (int i = 0; i < 6; ++i) add_slot(e->const_slots, e->pipes[g.pipe].d.arinyo_slot[i]);
                // level 2: whatever k_prologue forms ga / gb from, for the pipeline and its peak partner
                for (int pq : {g.pipe, g.peak_partner}) {
                    if (pq < 0) continue;
                    const vmx_pipe_desc& d = e->pipes[pq].d;
                    if (d.peak_nl) {
                        add_slot(e->const_slots2, d.sigma_nl_par_slot); add_slot(e->const_slots2, d.sigma_nl_per_slot);
                        if (d.sigma_nl_par_slot < 0 || d.sigma_nl_per_slot < 0) add_slot(e->const_slots2, d.growth_rate_slot);
                    }
                    for (int i = 0; i < d.n_smooth; ++i) { add_slot(e->const_slots2, d.smooth_par_slot[i]); add_slot(e->const_slots2, d.smooth_per_slot[i]); }
                    for (int q = 0; q < 2; ++q)
                        if (d.vd_kind == VMX_VD_GAUSS && d.tracer[q].discrete) add_slot(e->const_slots2, d.tracer[q].vd_sigma_slot);
                }
            }
        }
        for (int slot : e->const_slots) add_slot(e->const_slots2, slot);
        // two tables per group (level 2: the pipeline's and its peak partner's); + 4 x 32 rows: the mu loop requests its table
        // rows four steps ahead without checking for the end
        if (e->n_xtab > 0 && (e->xtab.alloc(((size_t)e->n_xtab * 2 * e->n_rows + 128) * e->nkp, true) ||
                              e->xtab_k.alloc((size_t)e->n_xtab * 4 * e->nkp, true))) return -2;
        {
            std::vector<int32_t> xp((size_t)e->n_xtab + 1, -1), xq((size_t)e->n_xtab + 1, -1);
            for (auto& g : e->pk_groups) if (g.xtab >= 0) { xp[g.xtab] = g.pipe; xq[g.xtab] = g.peak_partner; }
            e->tab2_groups.clear();
            for (int cross = 1; cross >= 0; --cross)
                for (auto& g : e->pk_groups) {
                    if (g.xtab < 0 || (g.variant == PKV_CROSS_CORE) != (cross == 1)) continue;
                    const vmx_pipe_desc& d = e->pipes[g.pipe].d;
                    Tab2Group t{};
                    t.pipe = g.pipe; t.partner = g.peak_partner; t.xtab = g.xtab; t.cross = cross;
                    t.kind_s = d.pk_lin_kind; t.kind_q = e->pipes[g.peak_partner].d.pk_lin_kind;
                    t.uvb = d.uvb; t.heii = d.heii; t.lya1 = d.tracer[0].is_lya; t.lya2 = d.tracer[1].is_lya;
                    t.damping_power = d.damping_power; t.damping_scale = d.damping_scale;
                    e->tab2_groups.push_back(t);          // (col_s / col_q: once the active columns are numbered)
                }
            std::vector<double> key((size_t)2 * e->n_xtab * VMX_XTAB_KEY + 1, std::nan(""));     // [held][seen by the running evaluation]
            if (e->d_xtab_pipe.upload(xp.data(), xp.size()) || e->d_xtab_partner.upload(xq.data(), xq.size()) ||
                e->xtab_key.upload(key.data(), key.size())) return -2;
            e->host_key_valid = false; e->pending_key.clear();
        }
        e->const_slots.push_back(-1);
        e->const_slots2.push_back(-1);
        if (e->d_const_slots.upload(e->const_slots.data(), e->const_slots.size()) ||
            e->d_const_slots2.upload(e->const_slots2.data(), e->const_slots2.size())) return -2;
        e->const_slots.pop_back();
        e->const_slots2.pop_back();
        e->pk_members.push_back(-1);
        if (e->d_pk_members.upload(e->pk_members.data(), e->pk_members.size())) return -2;
        // polynomial pipelines without any k-dependent walker term: their spline coefficients are linear in the Kaiser
        // coefficients with static vectors (k_poly_basis + the FFTLog operator, once) - no P(k,mu), no FFTLog column per
        // walker.  Every other pipeline gets a column of the P_ell / coefficient buffers.
        {
            std::vector<int32_t> keep;
            e->pk_static.clear();
            for (int p : e->pk_poly) {
                const vmx_pipe_desc& d = e->pipes[p].d;
                // (not with fht_extrap: the pads are not linear in P_ell)
                if (!d.uvb && !d.heii && !(d.damping_scale > 0.0) && e->static_poly && e->pad_l + e->pad_r == 0 && !getenv("VMX_NO_STATIC_POLY")) {
                    PipeDev& pd = e->pipes[p];
                    pd.poly_basis = (int32_t)e->pk_static.size();
                    e->pk_static.push_back(p);
                    // static coordinates as well (no rescaling, no delta_rp, no odd-multipole terms): the basis is
                    // evaluated on the bins once (k_poly_bins)
                    if (d.scale_mode == VMX_SCALE_UNIT && d.drp_slot < 0 && !pd.odd_rel && !pd.odd_asy && d.radiation != 2 &&
                        d.uv_shotnoise != 2 && !getenv("VMX_NO_STATIC_BINS")) {
                        pd.poly_bins_off = e->poly_bins_total;
                        e->poly_bins_total += (int64_t)3 * vmx_pad(pd.n);
                    }
                } else keep.push_back(p);
            }
            e->pk_poly.swap(keep);
            e->n_active = 0;
            for (auto& pd : e->pipes) {
                if (pd.poly_basis >= 0) pd.col = -1;
                else pd.col = e->n_active++;
            }
            for (auto& t : e->tab2_groups) { t.col_s = e->pipes[t.pipe].col; t.col_q = e->pipes[t.partner].col; }
            if (e->d_pipes.upload(e->pipes.data(), e->pipes.size())) return -2;
            e->pk_static.push_back(-1);
            if (e->d_pk_static.upload(e->pk_static.data(), e->pk_static.size())) return -2;
            e->pk_static.pop_back();
            std::vector<int32_t> active;
            for (int p = 0; p < (int)e->pipes.size(); ++p) if (e->pipes[p].col >= 0) active.push_back(p);
            // ... of which k_xi_bins_lean takes those that are a spline sum with the standard evolution (+ radiation)
            e->xi_lean_pipes.clear(); e->xi_rest_pipes.clear();
            for (int p : active) {
                const PipeDev& P = e->pipes[p];
                const vmx_pipe_desc& d = P.d;
                const bool lean = e->xi_lean && !e->extrapolate && d.tracer[0].evol_kind == VMX_EVOL_STD && d.tracer[1].evol_kind == VMX_EVOL_STD &&
                                  !d.uv_shotnoise && !P.odd_rel && !P.odd_asy && d.single_ell < 0 && (!d.radiation || !d.is_peak) &&      // (the radiating pair's peak: faster by the general kernel)
                                  (int)e->xi_lean_pipes.size() < VMX_XI_LEAN_MAX;
                (lean ? e->xi_lean_pipes : e->xi_rest_pipes).push_back(p);
            }
            {
                std::vector<int32_t> rest = e->xi_rest_pipes;
                rest.push_back(-1);
                if (e->d_xi_rest_pipes.upload(rest.data(), rest.size())) return -2;
            }
            active.push_back(-1);
            if (e->d_pipe_active.upload(active.data(), active.size())) return -2;
        }
        e->pk_poly.push_back(-1);
        if (e->d_pk_poly.upload(e->pk_poly.data(), e->pk_poly.size())) return -2;
        e->pk_poly.pop_back();
        if (e->d_pk_groups.upload(e->pk_groups.data(), e->pk_groups.size())) return -2;
        {
            e->w_groups.clear();
            for (size_t gi = 0; gi < e->pk_groups.size(); ++gi) if (e->pk_groups[gi].variant == PKV_SHARED_W) e->w_groups.push_back((int32_t)gi);
            std::vector<int32_t> wl = e->w_groups;
            wl.push_back(-1);
            if (e->d_w_groups.upload(wl.data(), wl.size())) return -2;
        }
    }

    // metals
    int64_t xim_off = 0;
    std::vector<MetalDev> metals;
    for (auto* it : e->items)
        for (auto* m : it->metals) {
            REQUIRE(m->dev.d.pipeline >= 0 || m->dev.svec || m->dev.basis, "metal without pipeline and without static correlation");
            if (m->dev.mat_off >= 0) { m->dev.xim_off = xim_off; xim_off += (int64_t)Bm * it->dev.n_model_pad; }
            metals.push_back(m->dev);
        }
    e->xim_total = xim_off;
    metals.push_back(MetalDev{});
    if (e->d_metals.upload(metals.data(), metals.size())) return -2;

    // items
    int64_t model_off = 0, masked_off = 0;
    std::vector<ItemDev> items;
    for (auto* it : e->items) {
        REQUIRE(it->has_mask && it->has_data, "every item needs a mask and a data vector");
        ItemDev& d = it->dev;
        d.model_off = model_off; model_off += d.d.n_dist;
        d.masked_off = masked_off; masked_off += d.n_masked;
        REQUIRE(it->has_dm || it->has_csr || d.d.n_dist == d.d.n_model, "identity distortion needs n_dist == n_model");
        if (it->vec.alloc((size_t)Bm * d.n_model_pad, true)) return -2;
        if (it->res.alloc((size_t)Bm * d.n_masked_pad, true)) return -2;
        if ((it->has_dm || it->has_csr) && it->dist.alloc((size_t)e->slab_rows * d.n_dist_pad, true)) return -2;
        if (it->has_cinv && it->z.alloc((size_t)e->slab_rows * d.n_masked_pad, true)) return -2;
        d.dm = it->has_dm ? it->dm.p : nullptr; d.dm_ld = d.n_model_pad;
        d.dm_ptr = it->has_csr ? it->csr_ptr.p : nullptr; d.dm_idx = it->csr_idx.p; d.dm_val = it->csr_val.p;
        d.cinv = it->has_cinv ? it->cinv.p : nullptr; d.cinv_ld = d.n_masked_pad;
        d.inv_mask = it->inv_mask.p; d.data = it->data.p;
        d.vec = it->vec.p; d.dist = it->dist.p; d.res = it->res.p; d.z = it->z.p;
        items.push_back(d);
    }
    e->model_size = (int)model_off;
    if (e->d_items.upload(items.data(), items.size())) return -2;
    if (e->gcinv.p) {
        REQUIRE(e->g_n == (int)masked_off, "global inverse covariance size must equal the total masked size");
        if (e->gres.alloc((size_t)Bm * e->g_ld, true) || e->gz.alloc((size_t)e->slab_rows * e->g_ld, true)) return -2;
    }

    if (e->bb_basis.upload(e->h_bb.data(), e->h_bb.size() ? e->h_bb.size() : 0)) return -2;
    if (e->odd_coef.upload(e->h_odd.data(), e->h_odd.size())) return -2;
    if (!e->h_odd_op.empty()) {
        // direct_pk with odd-multipole terms: the operators, and the walkers' coefficient rows [pipelines with one][Bm][ld]
        if (e->odd_op.upload(e->h_odd_op.data(), e->h_odd_op.size())) return -2;
        int64_t total = 0;
        for (auto& kv : e->odd_op_off) total += (int64_t)Bm * e->pipes[kv.first].odd_dyn_ld;       // (offsets: set with the descriptors above)
        if (e->odd_dyn.alloc((size_t)total, true)) return -2;
    }
    for (auto& p : e->pipes)
        for (int i = 0; i < 5; ++i) REQUIRE(p.odd_slot[i] < n_params, "odd-multipole slot exceeds n_params");
    if (!e->prior_slot.empty()) {
        for (int s : e->prior_slot) REQUIRE(s < n_params, "prior slot exceeds n_params");
        if (e->d_prior_slot.upload(e->prior_slot.data(), e->prior_slot.size()) ||
            e->d_prior_mean.upload(e->prior_mean.data(), e->prior_mean.size()) ||
            e->d_prior_sigma.upload(e->prior_sigma.data(), e->prior_sigma.size())) return -2;
    }

    // workspace (pad regions are zeroed once here and never written afterwards)
    const size_t ncols = (size_t)Bm * n_pipe, acols = (size_t)Bm * std::max(e->n_active, 1);
    if (e->theta.alloc((size_t)Bm * n_params) || e->scal.alloc(ncols * VMX_NS) ||
        e->metal_bias.alloc((size_t)Bm * 3 * (e->metals.size() + 1)) ||
        e->pl.alloc((size_t)VMX_MAX_ELL * acols * e->nkp) || e->coef.alloc((size_t)VMX_MAX_ELL * acols * e->ncp) ||
        e->xi.alloc((size_t)e->xi_total) || e->xim.alloc((size_t)e->xim_total) ||
        e->model.alloc((size_t)Bm * e->model_size) || e->chi2.alloc(Bm) || e->status.alloc(Bm) || e->k_live.alloc(8)) return -2;
    {
        const int32_t empty_window[2] = {0x7fffffff, -1};
        if (e->coef_win.upload(empty_window, 2)) return -2;
    }
    e->h_mock_index.assign(Bm, -1);
    if (e->mock_index.upload(e->h_mock_index.data(), Bm)) return -2;

    EngineDev& D = e->dev;
    D = EngineDev{};
    D.nk = e->nk; D.nkp = e->nkp; D.n_mu = e->n_mu; D.n_ell = VMX_MAX_ELL;
    D.pad_l = e->pad_l; D.pad_r = e->pad_r; D.pipe_active = e->d_pipe_active.p;
    D.n_rows = e->n_rows; D.n_extra = e->n_extra; D.mu_lo = e->mu_lo; D.mu_hi = e->mu_hi; D.node_w = e->node_w.p; D.mu_img = e->mu_img.p; D.mu_img_w = e->mu_img_w.p;
    {
        // the node rule needs the integrand smooth on the scale of its panels: the binning sincs oscillate with k x bin
        // size, so wavenumbers beyond 24 / (largest bin size) [6 h/Mpc for 4 Mpc/h bins] keep the midpoint loop
        double size = 4.0;
        for (auto& g : e->gk_tables) size = std::max(std::max(size, g.rp), std::max(g.rt, std::max(g.mock_rp, g.mock_rt)));
        e->k_node_max = 24.0 / size;
        if (getenv("VMX_EXACT_MU") || e->n_extra == 0) e->mu_nodes_on = false;
        D.k_node_max = e->mu_nodes_on ? e->k_node_max : 0.0;
    }
    for (int sl : e->rule_slot) REQUIRE(sl < n_params, "mu-rule box slot exceeds n_params");
    if (!e->rule_slot.empty() && (e->d_rule_slot.upload(e->rule_slot.data(), e->rule_slot.size()) ||
                                  e->d_rule_lo.upload(e->rule_lo.data(), e->rule_lo.size()) ||
                                  e->d_rule_hi.upload(e->rule_hi.data(), e->rule_hi.size()))) return -2;
    D.rule_slot = e->d_rule_slot.p; D.rule_lo = e->d_rule_lo.p; D.rule_hi = e->d_rule_hi.p; D.n_rule = (int)e->rule_slot.size();
    D.k = e->k.p; D.pklin = e->pklin.p; D.delta2 = e->delta2.p; D.mu = e->mu.p; D.sq1mmu2 = e->sq1mmu2.p; D.lnmu = e->lnmu.p;
    D.wl = e->wl.p; D.fv_x = e->fv_x.p; D.fv_f = e->fv_f.p; D.fv_n = e->fv_n; D.gk = e->gk.p; D.gk_mom = e->gk_mom.p; D.xtab = e->xtab.p; D.const_slots = e->d_const_slots.p; D.n_const_slots = 0; D.xtab_pipe = e->d_xtab_pipe.p; D.n_xtab = e->n_xtab; D.xtab_key = e->xtab_key.p; D.xtab_partner = e->d_xtab_partner.p; D.xtab_k = e->xtab_k.p; D.xtab_level = 0; D.n_gk = (int)e->gk_tables.size();
    D.n_coef = e->n_coef; D.ncp = e->ncp; D.extrapolate = e->extrapolate ? 1 : 0;
    for (int i = 0; i < VMX_MAX_ELL; ++i) {
        D.x0[i] = e->x0[i]; D.h[i] = e->h[i]; D.inv_h[i] = 1.0 / e->h[i]; D.xlast[i] = e->x0[i] + e->h[i] * (e->n_knots - 1);
    }
    D.same_grid = 1;
    for (int i = 1; i < VMX_MAX_ELL; ++i)
        if (e->op_set[i] && (!e->op_set[0] || D.x0[i] != D.x0[0] || D.inv_h[i] != D.inv_h[0] || D.xlast[i] != D.xlast[0])) D.same_grid = 0;
#ifdef VMX_EXP_NO_SAME_GRID           // (experiment build: the per-multipole instances on a common grid - scripts/gpu_same_grid.py checks the bits)
    D.same_grid = 0;
#endif
    D.n_pipe = n_pipe; D.pipes = e->d_pipes.p;
    D.n_active = e->n_active; D.n_static = (int)e->pk_static.size();
    if (!e->pk_static.empty() && e->poly_coef.alloc((size_t)VMX_MAX_ELL * e->pk_static.size() * 3 * e->ncp, true)) return -2;
    D.poly_coef = e->poly_coef.p;
    if (e->poly_bins_total > 0 && e->poly_bins.alloc((size_t)e->poly_bins_total, true)) return -2;
    D.poly_bins = e->poly_bins.p;
    D.cr = e->cr.p; D.cmu = e->cmu.p; D.crp = e->crp.p; D.crt = e->crt.p; D.cz = e->cz.p; D.crelz = e->crelz.p; D.clnrelz = e->clnrelz.p; D.clnrelz2 = e->clnrelz2.p; D.cgrowth = e->cgrowth.p;
    D.n_items = (int)e->items.size(); D.items = e->d_items.p;
    D.metals = e->d_metals.p; D.n_metals_total = (int)e->metals.size();
    D.bb_basis = e->bb_basis.p;
    D.odd_coef = e->odd_coef.p; D.odd_dyn = e->odd_dyn.p;
    D.sn_a = e->sn_a.p; D.sn_n = e->sn_n; D.sn_tau0 = e->sn_tau0; D.sn_dtau = e->sn_dtau;
    D.n_priors = (int)e->prior_slot.size();
    D.prior_slot = e->d_prior_slot.p; D.prior_mean = e->d_prior_mean.p; D.prior_sigma = e->d_prior_sigma.p;
    D.n_params = n_params;
    D.theta = e->theta.p; D.scal = e->scal.p; D.metal_bias = e->metal_bias.p; D.pl = e->pl.p; D.coef = e->coef.p;
    D.xi = e->xi.p; D.xim = e->xim.p; D.model = e->model.p; D.chi2 = e->chi2.p; D.status = e->status.p; D.k_live = e->k_live.p; D.coef_win = e->coef_win.p; D.pk_trace = nullptr; D.mock_index = e->mock_index.p;
    D.model_size = e->model_size;
    D.gcinv = e->gcinv.p; D.g_n = e->g_n; D.g_ld = e->g_ld; D.gres = e->gres.p; D.gz = e->gz.p;

    for (size_t q = 1; q < e->items.size(); ++q) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        HIP_OK(hipStreamCreate(&st));
        HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->aux.push_back(st); e->ev_join.push_back(ev);
    }
    HIP_OK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));

    HIP_OK(hipHostMalloc((void**)&e->pin_theta, (size_t)Bm * n_params * sizeof(double), hipHostMallocMapped));
    HIP_OK(hipHostMalloc((void**)&e->pin_chi2, (size_t)Bm * sizeof(double), hipHostMallocMapped));
    HIP_OK(hipHostMalloc((void**)&e->pin_status, (size_t)Bm * sizeof(int32_t), hipHostMallocMapped));
    HIP_OK(hipHostMalloc((void**)&e->pin_done, sizeof(int64_t), hipHostMallocMapped));
    *e->pin_done = 0;
    if (hipHostMalloc((void**)&e->pin_part, (size_t)VMX_MAX_GROUP * 1024 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&e->dpin_part, e->pin_part, 0) != hipSuccess) { (void)hipGetLastError(); e->dpin_part = nullptr; }
    if (getenv("VMX_NO_DONE_WORD") || hipHostGetDevicePointer((void**)&e->dpin_done, e->pin_done, 0) != hipSuccess) { (void)hipGetLastError(); e->dpin_done = nullptr; }
    if (!getenv("VMX_NO_ZERO_COPY") &&
        (hipHostGetDevicePointer((void**)&e->dpin_theta, e->pin_theta, 0) != hipSuccess ||
         hipHostGetDevicePointer((void**)&e->dpin_chi2, e->pin_chi2, 0) != hipSuccess ||
         hipHostGetDevicePointer((void**)&e->dpin_status, e->pin_status, 0) != hipSuccess)) {
        (void)hipGetLastError();
        e->dpin_theta = nullptr; e->dpin_chi2 = nullptr; e->dpin_status = nullptr;     // staging copies instead
    }

    // the quadratic form of chi2 needs a model that is linear in [x ; additive post-distortion coefficients]
    e->quad_eligible = e->gcinv.p == nullptr && e->items.size() <= 16;
    for (auto* it : e->items) {
        int na = 0;
        if (it->dev.n_bb[VMX_BB_POST_MUL]) e->quad_eligible = false;
        for (int q = 0; q < it->dev.n_bb[VMX_BB_POST_ADD]; ++q) {
            if (it->dev.bb[VMX_BB_POST_ADD][q].func != VMX_BB_POLY) e->quad_eligible = false;
            na += it->dev.bb[VMX_BB_POST_ADD][q].n_coef;
        }
        if (na > VMX_MAX_QUAD_COEF) e->quad_eligible = false;
    }
    if (getenv("VMX_NO_QUAD")) e->quad_eligible = false;

    HIP_OK(hipStreamSynchronize(e->stream));
    e->finalized = true;
    if (poly_basis_build(e) || xi_sum_plan(e)) { e->finalized = false; return -2; }
    return 0;
}

int vmx_model_size(vmx_engine* e) { return e ? e->model_size : -1; }

int vmx_pipeline_column(vmx_engine* e, int32_t pipeline)
{
    REQUIRE(e && e->finalized && pipeline >= 0 && pipeline < (int)e->pipes.size(), "vmx_pipeline_column");
    return e->pipes[pipeline].col >= 0 ? e->pipes[pipeline].col : -3 - e->n_active;     // (< -2: no column; the count is -3 - value)
}

// The quadratic-form launch as a persistent tape ("stream-K"): the cut, the slot numbering and the block queues are planned by
// vmx_plan::plan_quad_tape (vmx_plan.h - plain C++, run under sanitizers by tests/test_planner_host.py), a pure function of
// (problem shapes, walker tiles, blocks).
// cost of starting / finishing an entry in K stages (fitted from the block trace: duration = 1.98 us stages + 7.3 us entries
// + 9.8), and the share of a piece moved from the blocks dispatched second to those dispatched first (B = 256: 145.1 us at 0,
// 142.6 at 0.12 - 0.16, 144.0 at 0.2, 149.6 at 0.4)
constexpr double QUAD_ENTRY_STAGES = 4.0, QUAD_SKEW = 0.12;
// The equal-cost pieces of the tape are dealt to the XCDs by their place in K (the heavy and the light ones separately, so that
// the skew keeps its meaning): same pieces, same slots, the same sums bit for bit - 12 % fewer L2 misses of the walker operand
// (493 -> 435 MB per launch at B = 256), the same launch time with one batch in flight and ~1 % less with two (round 4).
constexpr bool QUAD_K_BANDS = true;

static vmx_engine::QuadList* quad_build_tape(vmx_engine* e, int B)
{
    const int tn = (B + GEMM_BN - 1) / GEMM_BN;
    std::vector<vmx_plan::TapeProblem> probs;
    for (auto* it : e->items) probs.push_back({it->dev.nq, it->dev.nq_pad});
    // (k_bands = false: dealing the pieces to the XCDs by their place in K takes 12 % off the launch's L2 misses - 493 -> 435 MB at
    // B = 256 - and nothing off its time: 151.4 us either way, round 4; the sums are the same bit for bit)
    vmx_plan::Tape T = vmx_plan::plan_quad_tape(probs, tn, e->quad_blocks, QUAD_ENTRY_STAGES, QUAD_SKEW, GEMM_BM, GEMM_BK, QUAD_K_BANDS);
    auto* ql = new vmx_engine::QuadList();
    ql->n_blocks = T.n_blocks;
    ql->n_entries = (int)T.n_slots;
    if (T.work.empty()) T.work.push_back(GemmWork{-1, 0, 0, 0, 0, 0, 0, 0});
    if (ql->work.upload(T.work.data(), T.work.size()) || ql->queue.upload(T.queue.data(), T.queue.size()) ||
        ql->nt_off.upload(T.nt_off.data(), T.nt_off.size()) ||
        ql->part.alloc(std::max<size_t>((size_t)T.n_slots, 1) * 128, true)) { delete ql; return nullptr; }
    return ql;
}

// the launch itself: every block walks its queue of the tape and leaves contraction partials in the tape's slots
static void quad_launch_list(vmx_engine* e, vmx_engine::QuadList* ql, int B)
{
    GemmGroup G{};
    for (size_t q = 0; q < e->items.size(); ++q) {
        ItemHost* it = e->items[q];
        const ItemDev& d = it->dev;
        GemmArgs g{};
        g.A = it->q_mat.p; g.lda = d.nq_pad; g.X = it->q_x.p; g.ldx = d.nq_pad; g.D = it->q_z.p; g.ldd = d.nq_pad;
        g.M = d.nq; g.N = B; g.K = d.nq_pad; g.tri = 1; g.nsplit = 1; g.klen = d.nq_pad;
        g.d_slab = (int64_t)B * d.nq_pad;
        g.tm = (d.nq + GEMM_BM - 1) / GEMM_BM; g.tn = (B + GEMM_BN - 1) / GEMM_BN;
        g.part = ql->part.p; g.lin = it->q_lin.p; g.lin_row = e->call_mock ? e->call_mock : e->mock_index.p; g.lin_pool = d.mock_pool ? 1 : 0;
        g.row0 = vmx_plan::tape_row0(d.nq, GEMM_BM);       // (the tape's K ranges are those of this tiling)
        G.p[G.n++] = g;
    }
    G.work = ql->work.p;
    G.queue = ql->queue.p;
    if (getenv("VMX_QUAD_TRACE")) {          // block timeline of this launch (debugging aid, written by vmx_sync as VMX_GEMM_TRACE is)
        e->gemm_trace_blocks = (size_t)ql->n_blocks;
        if (e->gemm_trace.n < 4 * e->gemm_trace_blocks) (void)e->gemm_trace.alloc(4 * e->gemm_trace_blocks, true);
        else (void)hipMemsetAsync(e->gemm_trace.p, 0, 4 * e->gemm_trace_blocks * sizeof(unsigned long long), e->cur);
        G.trace = e->gemm_trace.p;
    }
    hipLaunchKernelGGL((k_gemm_nt44<KC_QUAD>), dim3(ql->n_blocks, 1), dim3(GEMM44_THREADS), 0, e->cur, G);
}

static vmx_engine::QuadList* quad_work_list(vmx_engine* e, int B)
{
    // a tape depends on the batch size through its number of walker tiles only
    const int tn = (B + GEMM_BN - 1) / GEMM_BN;
    auto found = e->quad_lists.find(tn);
    if (found != e->quad_lists.end()) return found->second;
    vmx_engine::QuadList* ql = quad_build_tape(e, B);
    if (ql) e->quad_lists[tn] = ql;
    return ql;
}

// chi2 of the full chain, r^T C^-1 r = 2 r^T (L r) with the half-form inverse covariance L, is the same contraction as the
// quadratic form's: the tape over the items' covariances (or over the global one), the residuals as walker vectors, no linear
// term.  (Round 5: the 16x16x4 product + slabs + k_chi2 took 98 us at B = 256 where this launch takes ~30.)
static bool cinv_tape_applies(const vmx_engine* e, int B)
{
    if (B <= 8 || e->no_cinv_tape) return false;
    if (e->gcinv.p) return true;
    if (e->items.size() > VMX_MAX_GROUP) return false;
    for (auto* it : e->items) if (!it->has_cinv) return false;
    return true;
}

static vmx_engine::QuadList* cinv_work_list(vmx_engine* e, int B)
{
    const int tn = (B + GEMM_BN - 1) / GEMM_BN;
    auto found = e->cinv_lists.find(tn);
    if (found != e->cinv_lists.end()) return found->second;
    std::vector<vmx_plan::TapeProblem> probs;
    int longest = 0;
    if (e->gcinv.p) { probs.push_back({e->g_n, e->g_ld}); longest = e->g_ld; }
    else for (auto* it : e->items) { probs.push_back({it->dev.n_masked, it->dev.n_masked_pad}); longest = std::max(longest, (int)it->dev.n_masked_pad); }
    if (e->zero_row.n < (size_t)longest && e->zero_row.alloc((size_t)longest, true)) return nullptr;
    vmx_plan::Tape T = vmx_plan::plan_quad_tape(probs, tn, e->quad_blocks, QUAD_ENTRY_STAGES, QUAD_SKEW, GEMM_BM, GEMM_BK, QUAD_K_BANDS);
    auto* ql = new vmx_engine::QuadList();
    ql->n_blocks = T.n_blocks;
    ql->n_entries = (int)T.n_slots;
    if (T.work.empty()) T.work.push_back(GemmWork{-1, 0, 0, 0, 0, 0, 0, 0});
    if (ql->work.upload(T.work.data(), T.work.size()) || ql->queue.upload(T.queue.data(), T.queue.size()) ||
        ql->nt_off.upload(T.nt_off.data(), T.nt_off.size()) ||
        ql->part.alloc(std::max<size_t>((size_t)T.n_slots, 1) * 128, true)) { delete ql; return nullptr; }
    e->cinv_lists[tn] = ql;
    return ql;
}

static void cinv_launch_list(vmx_engine* e, vmx_engine::QuadList* ql, int B)
{
    GemmGroup G{};
    auto add = [&](const double* A, const double* X, int n, int ld) {
        GemmArgs g{};
        g.A = A; g.lda = ld; g.X = X; g.ldx = ld; g.D = nullptr; g.ldd = ld;
        g.M = n; g.N = B; g.K = ld; g.tri = 1; g.nsplit = 1; g.klen = ld;
        g.d_slab = (int64_t)B * ld;
        g.tm = (n + GEMM_BM - 1) / GEMM_BM; g.tn = (B + GEMM_BN - 1) / GEMM_BN;
        g.part = ql->part.p; g.lin = e->zero_row.p; g.lin_row = nullptr; g.lin_pool = 0;
        g.row0 = vmx_plan::tape_row0(n, GEMM_BM);
        G.p[G.n++] = g;
    };
    if (e->gcinv.p) add(e->gcinv.p, e->gres.p, e->g_n, e->g_ld);
    else for (auto* it : e->items) add(it->cinv.p, it->res.p, it->dev.n_masked, it->dev.n_masked_pad);
    G.work = ql->work.p;
    G.queue = ql->queue.p;
    hipLaunchKernelGGL((k_gemm_nt44<KC_QUAD>), dim3(ql->n_blocks, 1), dim3(GEMM44_THREADS), 0, e->cur, G);
}

// static spline-coefficient basis of the polynomial pipelines: C[ell][basis][i] = OP_ell . V_i[ell] (k_poly_basis), with the
// product kernels of the chain; redone when the linear spectra change (vmx_set_linear_spectra)
static int poly_basis_build(vmx_engine* e)
{
    e->poly_dirty = false;
    const int ns = (int)e->pk_static.size();
    e->xi_static_taps.clear(); e->xi_static_bins.clear();
    e->sums_dirty = true;
    if (ns == 0) return 0;
    DevBuf<double> V;
    const int64_t rows = (int64_t)ns * 3;
    if (V.alloc((size_t)VMX_MAX_ELL * rows * e->nkp, true)) return -2;
    e->cur = e->stream;
    hipLaunchKernelGGL(k_poly_basis, dim3((e->nkp + 255) / 256, ns), dim3(256), 0, e->stream, e->dev, e->d_pk_static.p, V.p);
    launch_product(e, KC_OTHER, e->op.p, e->nkp, (int64_t)e->ncp * e->nkp, e->n_coef, e->nkp,
                   V.p, e->nkp, rows * e->nkp, (int)rows, e->poly_coef.p, e->ncp, rows * e->ncp, VMX_MAX_ELL, (int)rows);
    HIP_OK(hipGetLastError());
    if (e->poly_bins_total > 0) {
        std::vector<int32_t> ok(ns, 1);
        DevBuf<int32_t> d_ok;
        if (d_ok.upload(ok.data(), ok.size())) return -2;
        int max_n = 0;
        for (int p : e->pk_static) max_n = std::max(max_n, (int)e->pipes[p].n);
        hipLaunchKernelGGL(k_poly_bins, dim3((max_n + 255) / 256, ns, 3), dim3(256), 0, e->stream, e->dev, e->d_pk_static.p,
                           e->poly_bins.p, d_ok.p);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(e->stream));
        HIP_OK(hipMemcpy(ok.data(), d_ok.p, ok.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        bool changed = false;
        for (int sb = 0; sb < ns; ++sb)
            if (!ok[sb] && e->pipes[e->pk_static[sb]].poly_bins_off >= 0) { e->pipes[e->pk_static[sb]].poly_bins_off = -1; changed = true; }
        if (changed) HIP_OK(hipMemcpy(e->d_pipes.p, e->pipes.data(), e->pipes.size() * sizeof(PipeDev), hipMemcpyHostToDevice));
    }
    HIP_OK(hipStreamSynchronize(e->stream));
    for (int p : e->pk_static) (e->pipes[p].poly_bins_off >= 0 ? e->xi_static_bins : e->xi_static_taps).push_back(p);
    std::vector<int32_t> a = e->xi_static_taps, b = e->xi_static_bins;
    a.push_back(-1); b.push_back(-1);
    if (e->d_xi_static_taps.upload(a.data(), a.size()) || e->d_xi_static_bins.upload(b.data(), b.size())) return -2;
    return 0;
}

// enqueue the whole kernel chain for the B parameter points already in e->theta
// Kronecker-form metal matrices of an item: one launch, one block per (walker, metal) (k_metal_kron)
static void launch_metal_kron(vmx_engine* e, const EngineDev& D, ItemHost* it, int B)
{
    int item = 0;
    for (size_t q = 0; q < e->items.size(); ++q) if (e->items[q] == it) item = (int)q;
    size_t shmem = 0;
    for (auto* m : it->metals) {
        if (!m->dev.kron_a) continue;
        const int n = m->dev.kron_nrp * m->dev.kron_nrt;
        shmem = std::max(shmem, (size_t)(2 * n + std::max(m->dev.kron_nrp * m->dev.kron_nrp, m->dev.kron_nrt * m->dev.kron_nrt)) * sizeof(double));
    }
    if (shmem == 0) return;
    ScopedTimer t(e, KC_METAL);
    if (!e->kron_attr) { (void)hipFuncSetAttribute((const void*)k_metal_kron, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e->kron_attr = true; }
    hipLaunchKernelGGL(k_metal_kron, dim3(B, (unsigned)it->metals.size()), dim3(256), shmem, e->cur, D, item, B);
}

// k_xi_quad_plain serves items that are plain peak / smooth pairs without additive template or pre-distortion broadband
static bool xi_plain_args(vmx_engine* e, XiPlainArgs& A)
{
    if (e->items.size() > VMX_MAX_GROUP || e->extrapolate || e->direct) return false;
    for (size_t q = 0; q < e->items.size(); ++q) {
        const ItemHost* it = e->items[q];
        const ItemDev& d = it->dev;
        if (!it->lean_pair || d.add_vec || d.n_bb[VMX_BB_PRE_MUL] || d.n_bb[VMX_BB_PRE_ADD]) return false;
        const PipeDev& Ps = e->pipes[d.d.pipe_smooth];
        const PipeDev& Pp = e->pipes[d.d.pipe_peak];
        if (Ps.d.n_ell != Pp.d.n_ell || Ps.d.single_ell >= 0 || Pp.d.single_ell >= 0 || Ps.col < 0 || Pp.col < 0) return false;
        XiPlainItem& I = A.it[q];
        I.coord_off = Pp.coord_off; I.q_x0 = it->q_x0.p; I.q_x = it->q_x.p;
        I.n_model = d.d.n_model; I.nq = d.nq; I.nq_pad = d.nq_pad; I.pipe_s = d.d.pipe_smooth; I.pipe_p = d.d.pipe_peak;
        I.col_s = Ps.col; I.col_p = Pp.col; I.n_ell = Ps.d.n_ell; I.split_evol = Pp.split_evol; I.bao_slot = d.d.bao_amp_slot;
        I.item = (int32_t)q;
        I.radiation = Ps.d.radiation;
    }
    return true;
}

// Which bins arrive pre-summed (ItemSums): per item, the lean per-walker pipelines in groups of four and the static-coordinate
// pipelines in groups of sixteen, each member with the factor it enters the item's vector with.  A pipeline with a second
// reader of its own array - another metal pair (fast_metals sharing), a metal matrix product - keeps that array (a group of
// one, presum = 0).  A pure function of the engine's configuration; rebuilt when the static basis is.
static int xi_sum_plan(vmx_engine* e)
{
    e->sums_dirty = false;
    e->lean_groups.clear(); e->static_groups.clear(); e->static_single.clear();
    e->sums_any = false;
    const int np = (int)e->pipes.size();
    struct Role { int item = -1, fkind = 0, findex = 0, metal = -1, refs = 0; bool plain = true; };
    std::vector<Role> role(np);
    int metal_base = 0;
    for (int q = 0; q < (int)e->items.size(); ++q) {
        const ItemHost* it = e->items[q];
        const vmx_item_desc& d = it->dev.d;
        auto take = [&](int p, int fkind, int findex, int metal, bool plain) {
            if (p < 0 || p >= np) return;
            Role& r = role[p];
            r.refs += 1; r.item = q; r.fkind = fkind; r.findex = findex; r.metal = metal; r.plain = r.plain && plain;
        };
        take(d.pipe_smooth, 0, 0, -1, true);
        if (d.pipe_peak != d.pipe_smooth) take(d.pipe_peak, 1, d.bao_amp_slot, -1, true);
        else role[d.pipe_smooth].plain = false;
        for (int m = 0; m < (int)it->metals.size(); ++m) {
            const MetalDev& md = it->metals[m]->dev;
            if (md.d.pipeline >= 0) take(md.d.pipeline, 2, metal_base + m, m, !md.basis && !md.svec && md.mat_off < 0 && m < 64);
        }
        metal_base += (int)it->metals.size();
    }
    std::vector<ItemSums> sums(e->items.size());
    for (size_t q = 0; q < sums.size(); ++q) { sums[q] = ItemSums{}; sums[q].n_pad = e->items[q]->dev.n_model_pad; }
    auto member_of = [&](int p) {
        const PipeDev& P = e->pipes[p];
        XiMember M{};
        M.coord_off = P.coord_off; M.poly_off = P.poly_bins_off; M.pipe = p; M.col = P.col; M.n_ell = P.d.n_ell;
        M.split_evol = P.split_evol; M.radiation = P.d.is_peak ? 0 : P.d.radiation; M.same_tracer = P.d.same_tracer;
        M.fkind = role[p].fkind; M.findex = role[p].findex;
        return M;
    };
    auto groupable = [&](int p) {
        const Role& r = role[p];
        return e->xi_sums && r.refs == 1 && r.plain && r.item >= 0 && e->pipes[p].n_pad == e->items[r.item]->dev.n_model_pad &&
               sums[r.item].n_arrays < 8;
    };
    auto mark = [&](int p) {
        const Role& r = role[p];
        ItemSums& sm = sums[r.item];
        if (r.fkind == 0) sm.core_smooth = 1;
        else if (r.fkind == 1) sm.core_peak = 1;
        else sm.metal_mask |= 1ull << r.metal;
    };
    // the lean per-walker pipelines
    for (int q = 0; q < (int)e->items.size(); ++q) {
        XiLeanGroup G{};
        auto flush = [&]() {
            if (G.n_members == 0) return;
            if (G.n_members == 1) G.presum = 0;        // (a sum of one: the plain array, read with its factor as before)
            else {
                sums[q].off[sums[q].n_arrays++] = G.out_off;
                for (int m = 0; m < G.n_members; ++m) mark(G.m[m].pipe);
                e->sums_any = true;
            }
            e->lean_groups.push_back(G);
            G = XiLeanGroup{};
        };
        for (int p : e->xi_lean_pipes) {
            if (role[p].item != q || !groupable(p)) continue;
            const PipeDev& P = e->pipes[p];
            if (G.n_members == 0) { G.out_off = P.xi_off; G.n = P.n; G.n_pad = P.n_pad; G.presum = 1; }
            G.m[G.n_members++] = member_of(p);
            if (G.n_members == VMX_XI_GROUP_MEMBERS || sums[q].n_arrays >= 7) flush();
        }
        flush();
    }
    for (int p : e->xi_lean_pipes) {
        bool placed = false;
        for (auto& G : e->lean_groups) for (int m = 0; m < G.n_members; ++m) placed = placed || G.m[m].pipe == p;
        if (placed) continue;
        const PipeDev& P = e->pipes[p];
        XiLeanGroup G{};
        G.out_off = P.xi_off; G.n = P.n; G.n_pad = P.n_pad; G.presum = 0; G.n_members = 1; G.m[0] = member_of(p);
        e->lean_groups.push_back(G);
    }
    // the static-coordinate pipelines (standard evolution, nothing added: what k_xi_bins_static_nw serves)
    auto static_lean = [&](int p) {
        const PipeDev& P = e->pipes[p];
        const vmx_pipe_desc& d = P.d;
        return d.tracer[0].evol_kind == VMX_EVOL_STD && d.tracer[1].evol_kind == VMX_EVOL_STD && !d.uv_shotnoise && !P.odd_rel && !P.odd_asy &&
               (!d.radiation || d.is_peak);
    };
    for (int q = 0; q < (int)e->items.size(); ++q) {
        XiStaticGroup G{};
        auto flush = [&]() {
            if (G.n_members == 0) return;
            if (G.n_members == 1) { e->static_single.push_back(G.m[0].pipe); G = XiStaticGroup{}; return; }
            sums[q].off[sums[q].n_arrays++] = G.out_off;
            for (int m = 0; m < G.n_members; ++m) mark(G.m[m].pipe);
            e->sums_any = true;
            e->static_groups.push_back(G);
            G = XiStaticGroup{};
        };
        for (int p : e->xi_static_bins) {
            if (role[p].item != q || !groupable(p) || !static_lean(p)) continue;
            const PipeDev& P = e->pipes[p];
            if (G.n_members == 0) { G.out_off = P.xi_off; G.n = P.n; G.n_pad = P.n_pad; G.presum = 1; }
            G.m[G.n_members++] = member_of(p);
            if (G.n_members == VMX_XI_SGROUP_MEMBERS) flush();
        }
        flush();
    }
    for (int p : e->xi_static_bins) {
        bool placed = false;
        for (auto& G : e->static_groups) for (int m = 0; m < G.n_members; ++m) placed = placed || G.m[m].pipe == p;
        for (int s1 : e->static_single) placed = placed || s1 == p;
        if (!placed) e->static_single.push_back(p);
    }
    HIP_OK(hipSetDevice(e->device));
    if (!e->lean_groups.empty() && e->d_lean_groups.upload(e->lean_groups.data(), e->lean_groups.size())) return -2;
    if (!e->static_groups.empty() && e->d_static_groups.upload(e->static_groups.data(), e->static_groups.size())) return -2;
    if (e->d_item_sums.upload(sums.data(), sums.size())) return -2;
    std::vector<int32_t> single = e->static_single;
    single.push_back(-1);
    if (e->d_static_single.upload(single.data(), single.size())) return -2;
    return 0;
}

static int run_chain(vmx_engine* e, int B, int tab_mode, bool zero_copy = false, const double* d_theta = nullptr,
                     double* d_chi2 = nullptr, int32_t* d_status = nullptr, bool quad = false,
                     const double* theta_by_value = nullptr)
{
    EngineDev D = e->dev;
    if (e->call_mock) D.mock_index = e->call_mock;      // (this call's walkers bring their own mock rows)
    // tab_mode: table level of the P(k,mu) stage (EngineDev::xtab_level)
    D.xtab_level = tab_mode;
    e->last_tab_level = tab_mode;
    D.n_const_slots = tab_mode >= 2 ? (int)e->const_slots2.size() : tab_mode ? (int)e->const_slots.size() : 0;
    if (tab_mode >= 2) D.const_slots = e->d_const_slots2.p;
    if (zero_copy) {
        D.theta_host = e->dpin_theta; D.theta_copy = e->theta.p; D.src_host = 1;
        D.chi2_host = e->dpin_chi2; D.status_host = e->dpin_status;
    } else if (d_theta) {
        // device entry point, eager launches: the first kernel reads the caller's walkers in place and the last one
        // writes the caller's outputs - no staging copies
        D.theta_host = d_theta; D.theta_copy = e->theta.p; D.src_host = 0;
        D.chi2_host = d_chi2; D.status_host = d_status;
    }
    if (theta_by_value && zero_copy && B == 1 && e->dpin_done && !e->profiling) { D.done_host = e->dpin_done; D.done_seq = ++e->done_seq; }
    const int n_pipe = D.n_pipe;
    const bool skip_xtab = e->skip_xtab_once;
    e->skip_xtab_once = false;
    bool xtab_launched = !(tab_mode && !skip_xtab);
    D.xtab_block0 = 0;
    {
        ScopedTimer t(e, KC_PROLOGUE);
        // grid = (2 n_pipe + 1 slots: the P(k,mu) half and the xi half of every pipeline's scalars, the walker-level values) x (chunks of PRO_T walkers) [+ the blocks that check the P(k,mu) tables against walker 0:
        // k_xtab's grid, flattened]; LDS of a block: the constant-slot list with walker 0's values,
        // the mu rule's box, its walkers
        const int n_pro = (2 * n_pipe + 1) * ((B + PRO_T - 1) / PRO_T);
        const size_t lds = ((size_t)2 * D.n_const_slots + (size_t)3 * e->rule_slot.size() + (size_t)PRO_T * (e->n_params | 1)) * sizeof(double);
        int n_tab = 0;
        // (walkers in mapped host memory: a thousand table blocks would each cross PCIe for walker 0 - k_xtab follows instead)
        if (!xtab_launched && !(zero_copy && !theta_by_value)) {
            const int rows = XTAB_ROWS * (256 / PRO_T);
            n_tab = ((e->nkp + PRO_T - 1) / PRO_T) * ((e->n_rows + rows - 1) / rows) * e->n_xtab;
            D.xtab_block0 = n_pro;
            xtab_launched = true;
        }
#ifdef VMX_EXP_PRO_TRACE
        if (getenv("VMX_PK_TRACE")) {
            if (!e->pk_trace.p && e->pk_trace.alloc(65536, true)) return -2;
            e->pk_trace_blocks = 2 * (size_t)n_pro;
            D.pk_trace = e->pk_trace.p;
        }
#endif
        if (lds > 160 * 1024 - 256) return fail(-1, "too many parameters for k_prologue's walker rows in LDS (64 x n_params doubles)");
        if (lds > 64 * 1024) {
            // beyond ~125 parameters the rows need more than the default 64 KB of dynamic LDS: the function attribute is the
            // driver's, per device and monotone - the largest request so far is remembered per device
            static std::mutex mu;
            static size_t allowed[64] = {};
            std::lock_guard<std::mutex> lock(mu);
            size_t& a = allowed[e->device & 63];
            if (lds > a) {
                HIP_OK(hipFuncSetAttribute((const void*)k_prologue, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                HIP_OK(hipFuncSetAttribute((const void*)k_prologue_byval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                a = lds;
            }
        }
#ifdef VMX_EXP_SKIP_SMALL
        // experiment build (timing only, results are wrong): what the step costs without this launch - the bound of what folding
        // it into its neighbour can give
        static const int exp_skip = getenv("VMX_EXP_SKIP") ? atoi(getenv("VMX_EXP_SKIP")) : 0;
        static int exp_calls = 0;
        if ((exp_skip & 2) && ++exp_calls > 64) goto exp_no_prologue;
#endif
        if (zero_copy && theta_by_value) {
            // (eager launches only: a captured graph would replay the walker it was captured with)
            ThetaArg ta;
            std::memcpy(ta.v, theta_by_value, (size_t)B * e->n_params * sizeof(double));
            hipLaunchKernelGGL(k_prologue_byval, dim3(n_pro + n_tab), dim3(PRO_T), lds, e->stream, ta, D, B);
        } else
            hipLaunchKernelGGL(k_prologue, dim3(n_pro + n_tab), dim3(PRO_T), lds, e->stream, D, B);
#ifdef VMX_EXP_SKIP_SMALL
        exp_no_prologue:;
#endif
    }
    {
        ScopedTimer t(e, KC_PK);
        // reduction scratch + (unless every looping group runs against its D_NL table) the mu^bv tables
        bool need_mubv = false;
        for (auto& g : e->pk_groups)
            if (e->pipes[g.pipe].d.nl_model == VMX_NL_ARINYO && !(tab_mode && g.xtab >= 0)) need_mubv = true;
        size_t shmem = ((need_mubv ? (size_t)e->n_rows : 0) + 2048) * sizeof(double);
        // + the (mu^2, mu^4) table of the tabulated and shared-W mu loops (the large-batch shapes; 40 KB per block at most,
        // four blocks per CU)
        bool want_mu_tab = false;
        for (auto& g : e->pk_groups)
            if ((tab_mode && g.xtab >= 0) || g.variant == PKV_SHARED_W) want_mu_tab = true;
        int mu_tab_off = want_mu_tab ? (int)(shmem / sizeof(double)) : -1;
        if (want_mu_tab) shmem += (size_t)2 * e->n_mu * sizeof(double);
        const int n_groups = (int)e->pk_groups.size();
        if (!xtab_launched)
            hipLaunchKernelGGL(k_xtab, dim3((e->nkp + 255) / 256, (e->n_rows + XTAB_ROWS - 1) / XTAB_ROWS, e->n_xtab), dim3(256), 0, e->stream, D);
        if (!e->pk_poly.empty())
            hipLaunchKernelGGL(k_pk_poly, dim3(B, (int)e->pk_poly.size()), dim3(256), 0, e->stream, D, e->d_pk_poly.p, B);
        int tm = tab_mode;
        // level-2 groups run in their own kernel; the general one follows for whatever else the configuration has
        int n_other = n_groups;
        if (tab_mode >= 2 && e->n_xtab > 0) {
            n_other = n_groups - e->n_xtab;
            // (the node tables' image, + [2 terms][2 walkers][64 wavenumbers] of UV / HeII bias terms behind it)
            const size_t sh1 = std::max<size_t>(2048, (size_t)2 * e->n_mu + 4 * e->n_extra + 256 + 16) * sizeof(double);     // (+ the waves' votes)
            if (sh1 > 64 * 1024) {
                // (num_bins_muk near its limit of 4096: the node tables alone are 64 KB - the driver's per-device attribute, monotone)
                static std::mutex mu;
                static size_t allowed[64] = {};
                std::lock_guard<std::mutex> lock(mu);
                size_t& a = allowed[e->device & 63];
                if (sh1 > a) {
                    HIP_OK(hipFuncSetAttribute((const void*)k_pk_tab2<64, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));
                    HIP_OK(hipFuncSetAttribute((const void*)k_pk_tab2<64, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));
                    HIP_OK(hipFuncSetAttribute((const void*)k_pk_tab2<16, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1));
                    a = sh1;
                }
            }
#ifdef VMX_EXP_PRO_TRACE       // (experiment build: the trace buffer holds k_prologue's stamps, scripts/gpu_pro_trace.py)
            D.pk_trace = nullptr;
            if (false) {
#else
            if (getenv("VMX_PK_TRACE")) {
#endif
                e->pk_trace_blocks = (size_t)B * ((e->nk + 7) / 8) * e->tab2_groups.size();     // (an upper bound: unused entries stay zero)
                if (!e->pk_trace.p && e->pk_trace.alloc(4 * (size_t)e->max_batch * ((e->nk + 7) / 8) * e->tab2_groups.size(), true)) return -2;
                D.pk_trace = e->pk_trace.p;
            }
            for (size_t first = 0; first < e->tab2_groups.size(); first += VMX_TAB2_GROUPS) {
                Tab2Args A{};
                const int n = (int)std::min<size_t>(VMX_TAB2_GROUPS, e->tab2_groups.size() - first);
                for (int q = 0; q < n; ++q) A.g[q] = e->tab2_groups[first + q];
                if ((int64_t)B * e->n_xtab >= 24) {
                    if (B >= 64)
                        hipLaunchKernelGGL((k_pk_tab2<64, 4, 2>), dim3((B + 1) / 2, n, (e->nk + 63) / 64), dim3(256), std::max(sh1, (size_t)4096 * sizeof(double)), e->stream, D, A, B);
                    else
                        hipLaunchKernelGGL((k_pk_tab2<64, 4, 1>), dim3(B, n, (e->nk + 63) / 64), dim3(256), sh1, e->stream, D, A, B);
                } else      // (8 x 32 blocks for a single walker measured the same: 12 us)
                    hipLaunchKernelGGL((k_pk_tab2<16, 16, 1>), dim3(B, n, (e->nk + 15) / 16), dim3(256), sh1, e->stream, D, A, B);
            }
        }
        // the shared-W groups in their own kernel (large batches)
        const int n_w = ((int64_t)B * (int)e->w_groups.size() >= 24 && !e->no_pk_w) ? (int)e->w_groups.size() : 0;
        if (n_w > 0) {
            n_other -= n_w;
            tm |= 32;
            const size_t shw = std::max<size_t>((size_t)2 * 6 * 256, (size_t)2 * e->n_mu + 4 * e->n_extra + 16) * sizeof(double);    // (+ the waves' votes)
            if (B >= 64)        // (four walkers per thread: 161 registers, 465 against 426 us for the stage at B = 512)
                hipLaunchKernelGGL((k_pk_w<2>), dim3((B + 1) / 2, n_w, (e->nk + 63) / 64), dim3(256), shw, e->stream, D, e->d_pk_groups.p, e->d_pk_members.p, e->d_w_groups.p, B);
            else
                hipLaunchKernelGGL((k_pk_w<1>), dim3(B, n_w, (e->nk + 63) / 64), dim3(256), shw, e->stream, D, e->d_pk_groups.p, e->d_pk_members.p, e->d_w_groups.p, B);
        }
        if (n_other == 0) {}
        else {
            // the instantiation without the run-time-switched loops needs fewer registers (4 instead of 3 waves per
            // SIMD): use it whenever every group has a specialised loop
            bool generic = false;
            for (auto& g : e->pk_groups) if (g.variant == PKV_GENERIC) generic = true;
            const PkGroup* gp = e->d_pk_groups.p;
            const int32_t* mp = e->d_pk_members.p;
#define VMX_LAUNCH_PK(KT, MS)                                                                                        \
            do {                                                                                                     \
                const dim3 grid(B, n_groups, (e->nk + (KT) - 1) / (KT));                                             \
                if (generic) hipLaunchKernelGGL((k_pk_multipoles<KT, MS, 1, true>), grid, dim3(256), shmem, e->stream, D, gp, mp, tm, B, mu_tab_off);   \
                else hipLaunchKernelGGL((k_pk_multipoles<KT, MS, 1, false>), grid, dim3(256), shmem, e->stream, D, gp, mp, tm, B, mu_tab_off);         \
            } while (0)
            if ((int64_t)B * n_groups >= 24) VMX_LAUNCH_PK(64, 4);
            else if ((int64_t)B * n_groups >= 4) VMX_LAUNCH_PK(16, 16);
            else {
                // + the block's slice of the G table, staged in LDS ([n_mu][8])
                shmem = ((size_t)e->n_rows + 2048 + (size_t)8 * e->n_rows) * sizeof(double);
                mu_tab_off = -1;
                if (!e->pk_small_attr) {
                    HIP_OK(hipFuncSetAttribute((const void*)k_pk_multipoles<8, 32, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                    HIP_OK(hipFuncSetAttribute((const void*)k_pk_multipoles<8, 32, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                    e->pk_small_attr = true;
                }
                VMX_LAUNCH_PK(8, 32);
            }
#undef VMX_LAUNCH_PK
        }
    }
    {
        const int64_t ncols = (int64_t)B * e->n_active;
        const bool pads = e->pad_l + e->pad_r > 0;
        if (ncols > 0 && pads) {
            // fht_extrap: the power-law pads of every (multipole, pipeline, walker) row, behind its samples
            ScopedTimer t(e, KC_FFTLOG);
            hipLaunchKernelGGL(k_pk_extrap, dim3((unsigned)(VMX_MAX_ELL * ncols)), dim3(256), 0, e->stream, D, B);
        }
        if (ncols > 0)      // (with pads the product runs over the whole row: the live-wavenumber limit is about the samples)
            launch_product(e, KC_FFTLOG, e->op.p, e->nkp, (int64_t)e->ncp * e->nkp, e->n_coef, e->nkp,
                           e->pl.p, e->nkp, ncols * e->nkp, (int)ncols, e->coef.p, e->ncp, ncols * e->ncp,
                           VMX_MAX_ELL, 0, pads ? nullptr : e->k_live.p, -1, false, e->coef_win.p);
    }
    // chi2-only small batches of items without metal terms: bins + quadratic-form entries in one kernel
    bool xi_fused = quad && (size_t)n_pipe == 2 * e->items.size();
    for (auto* it : e->items) if (!it->metals.empty() || it->dev.d.pipe_peak == it->dev.d.pipe_smooth) xi_fused = false;
    e->last_taps = !(xi_fused && B > 8);
    if (xi_fused) {
        ScopedTimer t(e, KC_XI);
        int max_nq = 0;
        for (auto* it : e->items) max_nq = std::max(max_nq, (int)it->dev.nq_pad);
        XiPlainArgs XA{};
        // (two walkers per thread: 41 us at B = 256 against 53 with one; four measured the same as two)
        if (B > 8 && xi_plain_args(e, XA))
            hipLaunchKernelGGL((D.same_grid ? k_xi_quad_plain<2, true> : k_xi_quad_plain<2, false>), dim3((max_nq + 255) / 256, (B + 1) / 2, (unsigned)e->items.size()), dim3(256), 0, e->stream, D, XA, 0, B);
        else
        hipLaunchKernelGGL(k_xi_assemble_quad, dim3((max_nq + 255) / 256, B, (unsigned)e->items.size()), dim3(256), 0, e->stream, D, 0, B <= 8 ? 1 : 0);
    } else {
        ScopedTimer t(e, KC_XI);
        int max_n = 0;
        for (auto& p : e->pipes) max_n = p.n > max_n ? p.n : max_n;
        // bins that arrive pre-summed (xi_sum_plan): large chi2-only or model batches of engines with lean pipelines
        const bool use_sums = B > 8 && !e->direct && e->xi_lean && e->xi_sums && !e->extrapolate;
        const bool sums_on = use_sums && e->sums_any && !e->sums_dirty;      // (the plan is made with the static basis, outside any capture)
        if (sums_on) { D.sums = e->d_item_sums.p; D.sums_on = 1; e->last_taps = false; }
        if (sums_on && !e->lean_groups.empty()) {
            // (walkers per thread of the lean kernels: 1 / 2 / 4 measured the same - the stage is bound by its dependent lookups)
            const dim3 grid((max_n + 255) / 256, (unsigned)e->lean_groups.size(), (B + 1) / 2);
            hipLaunchKernelGGL((D.same_grid ? k_xi_bins_group<2, true> : k_xi_bins_group<2, false>), grid, dim3(256), 0, e->stream, D, e->d_lean_groups.p, B);
            if (!e->xi_rest_pipes.empty())
                hipLaunchKernelGGL(k_xi_bins<false>, dim3((max_n + 255) / 256, (unsigned)e->xi_rest_pipes.size(), B), dim3(256), 0, e->stream, D, e->d_xi_rest_pipes.p);
        } else if (B > 8 && !e->direct && !e->xi_lean_pipes.empty()) {
            XiLeanArgs LA{};
            for (size_t q = 0; q < e->xi_lean_pipes.size(); ++q) {
                const PipeDev& P = e->pipes[e->xi_lean_pipes[q]];
                XiLeanPipe& L = LA.p[q];
                L.coord_off = P.coord_off; L.xi_off = P.xi_off; L.n = P.n; L.n_pad = P.n_pad; L.pipe = e->xi_lean_pipes[q]; L.col = P.col;
                L.n_ell = P.d.n_ell; L.split_evol = P.split_evol; L.radiation = P.d.is_peak ? 0 : P.d.radiation;
            }
            const dim3 grid((max_n + 255) / 256, (unsigned)e->xi_lean_pipes.size(), (B + 1) / 2);
            hipLaunchKernelGGL((D.same_grid ? k_xi_bins_lean<2, true> : k_xi_bins_lean<2, false>), grid, dim3(256), 0, e->stream, D, LA, B);
            if (!e->xi_rest_pipes.empty())
                hipLaunchKernelGGL(k_xi_bins<false>, dim3((max_n + 255) / 256, (unsigned)e->xi_rest_pipes.size(), B), dim3(256), 0, e->stream, D, e->d_xi_rest_pipes.p);
        } else if (e->n_active > 0)
            hipLaunchKernelGGL(k_xi_bins<false>, dim3((max_n + 255) / 256, e->n_active, B), dim3(256), 0, e->stream, D, e->d_pipe_active.p);
        if (!e->pk_static.empty())
        {
            // static-basis pipelines: tap form, and the ones already evaluated on their (static) bins
            if (!e->xi_static_taps.empty())
                hipLaunchKernelGGL(k_xi_bins<true>, dim3((max_n + 255) / 256, (unsigned)e->xi_static_taps.size(), B), dim3(256), 0, e->stream, D, e->d_xi_static_taps.p);
            if (sums_on && !e->xi_static_bins.empty()) {
                if (!e->static_groups.empty()) {
                    const dim3 grid((max_n + 255) / 256, (unsigned)e->static_groups.size(), (B + 1) / 2);
                    hipLaunchKernelGGL(k_xi_bins_static_group<2>, grid, dim3(256), 0, e->stream, D, e->d_static_groups.p, B);
                }
                if (!e->static_single.empty())
                    hipLaunchKernelGGL(k_xi_bins_static, dim3((max_n + 255) / 256, (unsigned)e->static_single.size(), B), dim3(256), 0, e->stream, D, e->d_static_single.p);
            } else if (!e->xi_static_bins.empty()) {
                bool lean = e->xi_lean && B > 8 && (int)e->xi_static_bins.size() <= VMX_XI_LEAN_MAX;
                for (int p : e->xi_static_bins) {
                    const PipeDev& P = e->pipes[p];
                    const vmx_pipe_desc& d = P.d;
                    lean = lean && d.tracer[0].evol_kind == VMX_EVOL_STD && d.tracer[1].evol_kind == VMX_EVOL_STD && !d.uv_shotnoise &&
                           !P.odd_rel && !P.odd_asy && (!d.radiation || d.is_peak);
                }
                if (lean) {
                    XiLeanArgs LA{};
                    for (size_t q = 0; q < e->xi_static_bins.size(); ++q) {
                        const PipeDev& P = e->pipes[e->xi_static_bins[q]];
                        XiLeanPipe& L = LA.p[q];
                        L.coord_off = P.coord_off; L.xi_off = P.xi_off; L.poly_off = P.poly_bins_off; L.n = P.n; L.n_pad = P.n_pad;
                        L.pipe = e->xi_static_bins[q]; L.split_evol = P.split_evol; L.same_tracer = P.d.same_tracer;
                    }
                    // (four walkers per thread load the three basis values and the two per-bin factors once: 96 -> 51 us at B = 512)
                    const dim3 grid((max_n + 255) / 256, (unsigned)e->xi_static_bins.size(), (B + 3) / 4);
                    hipLaunchKernelGGL(k_xi_bins_static_nw<4>, grid, dim3(256), 0, e->stream, D, LA, B);
                } else
                    hipLaunchKernelGGL(k_xi_bins_static, dim3((max_n + 255) / 256, (unsigned)e->xi_static_bins.size(), B), dim3(256), 0, e->stream, D, e->d_xi_static_bins.p);
            }
        }
    }
    // dense metal-matrix products of an item (no split-K: the consumer reads one slab)
    auto metal_products = [&](ItemHost* it) {
        launch_metal_kron(e, D, it, B);
        for (auto* m : it->metals) {
            if (m->dev.mat_off < 0 || m->dev.kron_a) continue;
            const PipeDev& P = e->pipes[m->dev.d.pipeline];
            launch_product(e, KC_METAL, m->mat.p, m->dev.mat_ld, 0, m->rows, m->dev.mat_ld,
                           e->xi.p + P.xi_off, P.n_pad, 0, B, e->xim.p + m->dev.xim_off, it->dev.n_model_pad, 0, 1, 0);
        }
    };
    if (quad) {
        // chi2 only: x' - x0' per item, ONE half-triangle product with the static Q' per item, reduction (vmx_device.h)
        e->cur = e->stream;
        int max_nq = 0;
        for (auto* it : e->items) { max_nq = std::max(max_nq, (int)it->dev.nq_pad); metal_products(it); }
        if (!xi_fused) {
            ScopedTimer t(e, KC_ASSEMBLE);
            hipLaunchKernelGGL(k_assemble_quad, dim3((max_nq + 255) / 256, B, (unsigned)e->items.size()), dim3(256), 0, e->cur, D);
        }
        SlabInfo qs{};
        for (size_t q = 0; q < e->items.size(); ++q) qs.z[q] = 1;
        if (e->quad_factored) {
            // chi2 = sum over items of || u0 - F dx ||^2: one rectangular product per item (all of them in one launch for B > 8,
            // split-K slabs summed in fixed order by the reduction), then the sum of squares
            if (B > 8 && e->items.size() <= VMX_MAX_GROUP) {
                GemmGroup G{};
                int tiles_total = 0, per_xcd_total = 0;
                for (auto* it : e->items) tiles_total += gemm_tiles(it->dev.n_masked, B, false);
                std::vector<size_t> by_size(e->items.size());
                for (size_t q = 0; q < by_size.size(); ++q) by_size[q] = q;
                std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) {
                    const ItemDev& da = e->items[a]->dev; const ItemDev& db = e->items[b]->dev;
                    return (int64_t)da.n_masked * da.nq > (int64_t)db.n_masked * db.nq; });
                std::vector<int>& splits = e->group_splits[3 * 100000 + B];
                if (splits.empty()) {
                    std::vector<SplitProblem> sp;
                    for (size_t q : by_size) {
                        const ItemDev& d = e->items[q]->dev;
                        int max_split = 1;
                        while (max_split < 8 && (int64_t)max_split * 2 * B <= e->fact_slab_rows) max_split *= 2;
                        sp.push_back({gemm_tiles(d.n_masked, B, false), (d.nq_pad + GEMM_BK - 1) / GEMM_BK, (int64_t)B * d.n_masked_pad * 8, max_split,
                                      (d.n_masked + GEMM_BM - 1) / GEMM_BM, (B + GEMM_BN - 1) / GEMM_BN});
                    }
                    splits = choose_group_splits(sp, true);
                    if (splits.empty()) splits.push_back(0);
                }
                size_t gi = 0;
                for (size_t q : by_size) {
                    ItemHost* it = e->items[q];
                    const ItemDev& d = it->dev;
                    GemmArgs g{};
                    g.A = it->q_f.p; g.lda = d.nq_pad; g.X = it->q_x.p; g.ldx = d.nq_pad; g.D = it->q_y.p; g.ldd = d.n_masked_pad;
                    g.M = d.n_masked; g.N = B; g.K = d.nq_pad;
                    int per_xcd = 0;
                    const int forced = gi < splits.size() ? splits[gi] : 0;
                    ++gi;
                    qs.z[q] = plan_gemm(e, g, 1, e->fact_slab_rows, nullptr, false, tiles_total - gemm_tiles(d.n_masked, B, false), &per_xcd, forced);
                    per_xcd_total += per_xcd;
                    G.p[G.n] = g; G.seq_end[G.n] = per_xcd_total; ++G.n;
                }
                ScopedTimer t(e, KC_QUAD);
                launch_gemm_group(e, KC_QUAD, G, per_xcd_total, 1);
            } else {
                for (size_t q = 0; q < e->items.size(); ++q) {
                    ItemHost* it = e->items[q];
                    const ItemDev& d = it->dev;
                    qs.z[q] = launch_product(e, KC_QUAD, it->q_f.p, d.nq_pad, 0, d.n_masked, d.nq_pad, it->q_x.p, d.nq_pad, 0, B,
                                             it->q_y.p, d.n_masked_pad, 0, 1, e->fact_slab_rows);
                }
            }
            {
                ScopedTimer t(e, KC_CHI2);
                hipLaunchKernelGGL(k_chi2_quad<true>, dim3(B), dim3(CHI2_THREADS), 0, e->stream, D, B, qs);
            }
            HIP_OK(hipGetLastError());
            e->last_B = B;
            e->last_full = false;
            e->last_form = 2;
            return 0;
        }
        e->last_form = 1;
        if (B > 8 && e->items.size() <= VMX_MAX_GROUP) {
            // the persistent tape: every block the same work to a stage, contraction partials in the tape's slots
            vmx_engine::QuadList* ql = quad_work_list(e, B);
            if (!ql) return -2;
            {
                ScopedTimer t(e, KC_QUAD);
                quad_launch_list(e, ql, B);
            }
            // the launch left contraction partials instead of the product: a small kernel adds them up
            ScopedTimer t(e, KC_CHI2);
#ifdef VMX_EXP_SKIP_SMALL
            static const int exp_skip1 = getenv("VMX_EXP_SKIP") ? atoi(getenv("VMX_EXP_SKIP")) : 0;
            static int exp_calls1 = 0;
            if (!((exp_skip1 & 1) && ++exp_calls1 > 64))
#endif
            hipLaunchKernelGGL(k_chi2_parts, dim3((B + 3) / 4), dim3(256), 0, e->stream, D, B, (const double*)ql->part.p,
                               (const int32_t*)ql->nt_off.p, 1);
            HIP_OK(hipGetLastError());
            e->last_B = B;
            e->last_full = false;
            return 0;
        } else if (B == 1 && D.done_host && e->dpin_part && !e->no_host_reduce && e->items.size() <= VMX_MAX_GROUP) {     // (Q' form)
            // single walker through the host entry: every block of the streaming products leaves its share of the contraction
            // in mapped host memory, the host adds them up (vmx_eval) - no chi2 kernel
            bool ok = true;
            for (auto* it : e->items) ok = ok && gemv1_applies(1, it->dev.nq_pad) && gemv1_blocks(it->dev.nq) <= 1024 && it->dev.nq_pad <= 2560 * 2;
            if (ok) {
                e->host_reduce_items = (int)e->items.size();
                for (size_t q = 0; q < e->items.size(); ++q) {
                    ItemHost* it = e->items[q];
                    const ItemDev& d = it->dev;
                    const int mock = (d.mock_pool && e->h_mock_index[0] >= 0) ? e->h_mock_index[0] : -1;
                    GemmArgs g{};
                    g.A = it->q_mat.p; g.lda = d.nq_pad; g.X = it->q_x.p; g.ldx = d.nq_pad; g.D = it->q_z.p; g.ldd = d.nq_pad;
                    g.M = d.nq; g.N = 1; g.K = d.nq_pad; g.tri = 1; g.nsplit = 1; g.klen = d.nq_pad;
                    g.part = e->dpin_part + q * 1024; g.lin = it->q_lin.p + (size_t)(mock >= 0 ? 1 + mock : 0) * d.nq_pad;
                    const int blocks = gemv1_blocks(d.nq);
                    e->host_reduce_blocks[q] = blocks;
                    ScopedTimer t(e, KC_QUAD);
                    const int last = q + 1 == e->items.size() ? 1 : 0;
                    if (d.nq_pad <= 2560) hipLaunchKernelGGL((k_gemv1<5, 2>), dim3(blocks), dim3(256), (size_t)d.nq_pad * sizeof(double), e->stream, g, D, last);
                    else hipLaunchKernelGGL((k_gemv1<10, 2>), dim3(blocks), dim3(256), (size_t)d.nq_pad * sizeof(double), e->stream, g, D, last);
                }
                HIP_OK(hipGetLastError());
                e->last_B = B;
                e->last_full = false;
                return 0;
            }
            for (size_t q = 0; q < e->items.size(); ++q) {
                ItemHost* it = e->items[q];
                const ItemDev& d = it->dev;
                qs.z[q] = launch_product(e, KC_QUAD, it->q_mat.p, d.nq_pad, 0, d.nq, d.nq_pad, it->q_x.p, d.nq_pad, 0, B,
                                         it->q_z.p, d.nq_pad, 0, 1, e->slab_rows, nullptr, -1, true);
            }
        } else {
            for (size_t q = 0; q < e->items.size(); ++q) {
                ItemHost* it = e->items[q];
                const ItemDev& d = it->dev;
                qs.z[q] = launch_product(e, KC_QUAD, it->q_mat.p, d.nq_pad, 0, d.nq, d.nq_pad, it->q_x.p, d.nq_pad, 0, B,
                                         it->q_z.p, d.nq_pad, 0, 1, e->slab_rows, nullptr, -1, true);
            }
        }
        {
            ScopedTimer t(e, KC_CHI2);
            hipLaunchKernelGGL(k_chi2_quad<false>, dim3(B), dim3(CHI2_THREADS), 0, e->stream, D, B, qs);
        }
        HIP_OK(hipGetLastError());
        e->last_B = B;
        e->last_full = false;
        return 0;
    }
    e->last_form = 0;
    SlabInfo slabs{};
    for (size_t q = 0; q < e->items.size(); ++q) slabs.z[q] = 1;
    const bool grouped = B > 8 && e->items.size() <= VMX_MAX_GROUP;
    if (grouped) {
        // Large batches: the items advance together on one stream and their products share launches - every tile of
        // every item's distortion product (then of every C^-1 product) is in flight at once.
        e->cur = e->stream;
        int max_model = 0, max_dist = 0;
        for (auto* it : e->items) {
            max_model = std::max(max_model, (int)it->dev.d.n_model); max_dist = std::max(max_dist, (int)it->dev.d.n_dist);
            metal_products(it);
        }
        {
            ScopedTimer t(e, KC_ASSEMBLE);
            hipLaunchKernelGGL(k_assemble_all, dim3((max_model + 255) / 256, B, (unsigned)e->items.size()), dim3(256), 0, e->cur, D);
        }
        SlabInfo dslabs{};
        for (size_t q = 0; q < e->items.size(); ++q) dslabs.z[q] = 1;
        for (int stage = 0; stage < 2; ++stage) {       // 0: distortion products, 1: C^-1 products
            if (stage == 1 && (e->gcinv.p || cinv_tape_applies(e, B))) break;
            GemmGroup G{};
            int tiles_total = 0, per_xcd_total = 0;
            for (auto* it : e->items) {
                const ItemDev& d = it->dev;
                if (stage == 0 ? it->has_dm : it->has_cinv)
                    tiles_total += stage == 0 ? gemm_tiles(d.d.n_dist, B, false) : gemm_tiles(d.n_masked, B, true);
            }
            // largest problem first: its blocks are the long ones, and the short blocks of the others then fill the tail
            std::vector<size_t> by_size(e->items.size());
            for (size_t q = 0; q < by_size.size(); ++q) by_size[q] = q;
            std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) {
                const ItemDev& da = e->items[a]->dev; const ItemDev& db = e->items[b]->dev;
                return stage == 0 ? (int64_t)da.d.n_dist * da.d.n_model > (int64_t)db.d.n_dist * db.d.n_model : da.n_masked > db.n_masked; });
            // K splits of the group: simulated once per (stage, batch size)
            std::vector<int>& splits = e->group_splits[stage * 100000 + B];
            if (splits.empty() && stage == 0) {        // (the model describes the self-pipelined 4x4x4 kernel)
                std::vector<SplitProblem> sp;
                for (size_t q : by_size) {
                    ItemHost* it = e->items[q];
                    const ItemDev& d = it->dev;
                    if (!(stage == 0 ? it->has_dm : it->has_cinv)) continue;
                    const int M = stage == 0 ? d.d.n_dist : d.n_masked, K = stage == 0 ? d.n_model_pad : d.n_masked_pad;
                    const int ldd = stage == 0 ? d.n_dist_pad : d.n_masked_pad;
                    int max_split = 1;
                    while (max_split < 8 && (int64_t)max_split * 2 * B <= e->slab_rows) max_split *= 2;
                    // (a triangular problem pairs its row tiles: every block covers about one full K range)
                    sp.push_back({gemm_tiles(M, B, stage == 1), (K + GEMM_BK - 1) / GEMM_BK, (int64_t)B * ldd * 8, max_split,
                                  (M + GEMM_BM - 1) / GEMM_BM, (B + GEMM_BN - 1) / GEMM_BN});
                }
                // (the one-block-per-CU model: 288 against 317 us for the distortion launch at B = 256 with the per-XCD model's
                // choice, 101 against 94 at B = 64 - fitted on these shapes in round 2 and kept for them)
                splits = choose_group_splits(sp);
            }
            if (splits.empty()) splits.push_back(0);
            size_t gi = 0;
            for (size_t q : by_size) {
                ItemHost* it = e->items[q];
                const ItemDev& d = it->dev;
                if (!(stage == 0 ? it->has_dm : it->has_cinv)) continue;
                const int forced = gi < splits.size() ? splits[gi] : 0;
                ++gi;
                GemmArgs g{};
                if (stage == 0) {
                    g.A = it->dm.p; g.lda = d.n_model_pad; g.X = it->vec.p; g.ldx = d.n_model_pad; g.D = it->dist.p; g.ldd = d.n_dist_pad;
                    g.M = d.d.n_dist; g.N = B; g.K = d.n_model_pad;
                } else {
                    g.A = it->cinv.p; g.lda = d.n_masked_pad; g.X = it->res.p; g.ldx = d.n_masked_pad; g.D = it->z.p; g.ldd = d.n_masked_pad;
                    g.M = d.n_masked; g.N = B; g.K = d.n_masked_pad;
                }
                int per_xcd = 0;
                const int own = stage == 0 ? gemm_tiles(g.M, B, false) : gemm_tiles(g.M, B, true);
                const int ns = plan_gemm(e, g, 1, e->slab_rows, nullptr, stage == 1, tiles_total - own, &per_xcd, forced);
                (stage == 0 ? dslabs : slabs).z[q] = ns;
                per_xcd_total += per_xcd;
                G.p[G.n] = g; G.seq_end[G.n] = per_xcd_total; ++G.n;
            }
            if (G.n > 0) {
                ScopedTimer t(e, stage == 0 ? KC_DISTORTION : KC_INVCOV);
                launch_gemm_group(e, stage == 0 ? KC_DISTORTION : KC_INVCOV, G, per_xcd_total, 1);
            }
            if (stage == 0)
                for (auto* it : e->items) if (it->has_csr) launch_csr(e, it, B);       // CSR matrices: 8 walkers per pass
            if (stage == 0) {
                ScopedTimer t(e, KC_POST);
                hipLaunchKernelGGL(k_post_all, dim3((max_dist + 255) / 256, B, (unsigned)e->items.size()), dim3(256), 0, e->cur, D, B, dslabs);
            }
        }
    } else {
    if (e->items.size() > 1) HIP_OK(hipEventRecord(e->ev_fork, e->stream));
    // small batches are latency-bound: the items run on forked streams; the item with the largest distortion product is
    // the critical path and is enqueued first, on the main stream
    std::vector<size_t> order(e->items.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return (int64_t)e->items[a]->dev.d.n_dist * e->items[a]->dev.d.n_model > (int64_t)e->items[b]->dev.d.n_dist * e->items[b]->dev.d.n_model; });
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const size_t q = order[oi];
        ItemHost* it = e->items[q];
        const ItemDev& d = it->dev;
        e->cur = oi == 0 ? e->stream : e->aux[oi - 1];
        if (oi > 0) HIP_OK(hipStreamWaitEvent(e->cur, e->ev_fork, 0));
        metal_products(it);
        // a single walker is latency-bound: its distortion product assembles x while staging it and finishes
        // each row with the post step, instead of three launches
        const bool fuse = B == 1 && it->has_dm && gemv1_applies(1, d.n_model_pad) && !e->no_fuse;
        if (!fuse) {
            ScopedTimer t(e, KC_ASSEMBLE);
            hipLaunchKernelGGL(k_assemble, dim3((d.d.n_model + 255) / 256, B), dim3(256), 0, e->cur, D, (int)q);
        }
        int dist_slabs = 1;
        if (it->has_dm)
            dist_slabs = launch_product(e, KC_DISTORTION, it->dm.p, d.n_model_pad, 0, d.d.n_dist, d.n_model_pad,
                                        it->vec.p, d.n_model_pad, 0, B, it->dist.p, d.n_dist_pad, 0, 1, e->slab_rows,
                                        nullptr, fuse ? (int)q : -1);
        else if (it->has_csr) launch_csr(e, it, B);
        if (!fuse) {
            ScopedTimer t(e, KC_POST);
            hipLaunchKernelGGL(k_post, dim3((d.d.n_dist + 255) / 256, B), dim3(256), 0, e->cur, D, (int)q, B, dist_slabs);
        }
        if (it->has_cinv && !e->gcinv.p)
            slabs.z[q] = launch_product(e, KC_INVCOV, it->cinv.p, d.n_masked_pad, 0, d.n_masked, d.n_masked_pad,
                                        it->res.p, d.n_masked_pad, 0, B, it->z.p, d.n_masked_pad, 0, 1, e->slab_rows,
                                        nullptr, -1, true);
        if (oi > 0) HIP_OK(hipEventRecord(e->ev_join[oi - 1], e->cur));
    }
    e->cur = e->stream;
    for (size_t q = 1; q < e->items.size(); ++q) HIP_OK(hipStreamWaitEvent(e->stream, e->ev_join[q - 1], 0));
    }
    e->cur = e->stream;
    if (grouped && cinv_tape_applies(e, B)) {
        vmx_engine::QuadList* ql = cinv_work_list(e, B);
        if (!ql) return -2;
        {
            ScopedTimer t(e, KC_INVCOV);
            cinv_launch_list(e, ql, B);
        }
        ScopedTimer t(e, KC_CHI2);
        hipLaunchKernelGGL(k_chi2_parts, dim3((B + 3) / 4), dim3(256), 0, e->stream, D, B, (const double*)ql->part.p,
                           (const int32_t*)ql->nt_off.p, 0);
        HIP_OK(hipGetLastError());
        e->last_B = B;
        e->last_full = true;
        return 0;
    }
    slabs.g = 1;
    if (e->gcinv.p)
        slabs.g = launch_product(e, KC_INVCOV, e->gcinv.p, e->g_ld, 0, e->g_n, e->g_ld, e->gres.p, e->g_ld, 0, B,
                                 e->gz.p, e->g_ld, 0, 1, e->slab_rows, nullptr, -1, true);
    {
        ScopedTimer t(e, KC_CHI2);
        hipLaunchKernelGGL(k_chi2, dim3(B), dim3(CHI2_THREADS), 0, e->stream, D, B, slabs);
    }
    HIP_OK(hipGetLastError());
    e->last_B = B;
    e->last_full = true;
    return 0;
}

// run the chain for B walkers: replay a captured graph when one exists (or can be captured), else launch eagerly
static int run_chain_cached(vmx_engine* e, int B, int tab_mode, bool zero_copy = false, bool quad = false)
{
    if (e->n_xtab == 0) tab_mode = 0;
    // (the covariance tape of the full chain is built - allocations, uploads - outside any stream capture)
    if (!quad && B > 8 && e->items.size() <= VMX_MAX_GROUP && cinv_tape_applies(e, B) && !cinv_work_list(e, B)) return -2;
    if (!e->use_graphs || e->profiling) return run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad);
    const int key = (((B * 4 + tab_mode) * 2 + (zero_copy ? 1 : 0)) * 2 + (e->direct ? 1 : 0)) * 2 + (quad ? 1 : 0);
    auto it = e->graphs.find(key);
    if (it == e->graphs.end()) {
        if (e->graphs.size() >= 64) return run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad);
        hipGraph_t graph = nullptr;
        HIP_OK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad);
        hipError_t err = hipStreamEndCapture(e->stream, &graph);
        if (rc || err != hipSuccess || graph == nullptr) {
            if (graph) (void)hipGraphDestroy(graph);
            std::fprintf(stderr, "[vegamx] stream capture failed (%s): falling back to eager launches\n",
                         err != hipSuccess ? hipGetErrorString(err) : "launch error");
            (void)hipGetLastError();
            e->use_graphs = false;      // capture is not available: stay on eager launches
            return run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad);
        }
        hipGraphExec_t exec = nullptr;
        err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (err != hipSuccess) {
            std::fprintf(stderr, "[vegamx] graph instantiation failed (%s): falling back to eager launches\n", hipGetErrorString(err));
            e->use_graphs = false;
            return run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad);
        }
        it = e->graphs.emplace(key, exec).first;
    }
    HIP_OK(hipGraphLaunch(it->second, e->stream));
    e->last_B = B;
    e->last_full = !quad;
    return 0;
}

// Build (or refresh) the static tensors of the quadratic form of chi2: the reference point is evaluated with the full
// chain (its pre-distortion vectors x0 and its model m0 come from the engine's own kernels), then per item
//   X = (S DM')^T,  W = X C^-1,  Q' = W X^T (stored in half form),  g_k = W (d_k - S m0),  c0_k = (d_k - S m0)^T C^-1 (d_k - S m0)
// for the data vector (k = 0) and every mock of the pool, all with the engine's product kernels.  Returns 0, or a
// negative code; a reference point the model cannot be evaluated at switches the form off (the full chain then runs).
static int quad_build(vmx_engine* e)
{
    drop_lane(e);           // (the tensors of the form are re-made: a second lane would keep views of the old ones)
    HIP_OK(hipStreamSynchronize(e->stream));
    const int P = e->n_params;
    std::vector<double> tref(e->theta_ref);
    if (!e->blind_scale.empty())
        for (int i = 0; i < P; ++i) tref[i] = e->blind_scale[i] == 1.0 ? tref[i] + e->blind_shift[i] : e->blind_scale[i] * tref[i] + e->blind_shift[i];
    HIP_OK(hipMemcpy(e->theta.p, tref.data(), (size_t)P * sizeof(double), hipMemcpyHostToDevice));
    e->no_fuse = true;
    const int rc = run_chain(e, 1, false);
    e->no_fuse = false;
    if (rc) return rc;
    HIP_OK(hipStreamSynchronize(e->stream));
    int32_t st = 0;
    HIP_OK(hipMemcpy(&st, e->status.p, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (st) {
        std::fprintf(stderr, "[vegamx] the reference point of the quadratic chi2 form cannot be evaluated (status %d): full chain only\n", st);
        e->quad_eligible = false;
        return 0;
    }
    e->cur = e->stream;
    // Which form: Q' costs nq^2 flops per walker and item (half form), the factored form 2 n_masked nq.  The tape of the Q'
    // product runs at 0.7 of the MFMA peak and needs no reduction of slabs, the factored product is a plain grouped launch:
    // it has to win by a third to be taken (nq > 2.67 n_masked: a model grid finer than the data grid, COEFMOD >= 2).
    {
        double cost_q = 0.0, cost_f = 0.0;
        for (auto* it : e->items) {
            int na = 0;
            for (int q = 0; q < it->dev.n_bb[VMX_BB_POST_ADD]; ++q) na += it->dev.bb[VMX_BB_POST_ADD][q].n_coef;
            const double nq = it->dev.d.n_model + na;
            cost_q += nq * nq; cost_f += 2.0 * it->dev.n_masked * nq;
        }
        const bool fact = e->quad_kind == 2 || (e->quad_kind == 0 && cost_f < 0.75 * cost_q);
        if (fact != e->quad_factored) { e->quad_factored = fact; e->quad_mat_dirty = true; }
    }
    for (auto* it : e->items) {
        ItemDev& d = it->dev;
        const int nm = d.n_masked, nmp = d.n_masked_pad;
        int na = 0;
        std::vector<int64_t> boff;
        for (int q = 0; q < d.n_bb[VMX_BB_POST_ADD]; ++q) {
            const BBTermDev& term = d.bb[VMX_BB_POST_ADD][q];
            for (int c = 0; c < term.n_coef; ++c) { d.q_slot[na++] = term.slot[c]; boff.push_back(term.basis_off + (int64_t)c * d.d.n_dist); }
        }
        d.q_na = na; d.nq = d.d.n_model + na; d.nq_pad = vmx_pad(d.nq);
        {
            // k_xi_assemble_quad's lean path: both components are plain spline sums with the standard bias evolution
            bool plain = d.d.pipe_peak != d.d.pipe_smooth;
            bool lean = plain;
            for (int pq : {d.d.pipe_peak, d.d.pipe_smooth}) {
                const PipeDev& P = e->pipes[pq];
                if (P.d.tracer[0].evol_kind != VMX_EVOL_STD || P.d.tracer[1].evol_kind != VMX_EVOL_STD ||
                    P.d.uv_shotnoise || P.odd_rel || P.odd_asy || P.poly_basis >= 0) { plain = false; lean = false; }
                if (P.d.radiation) plain = false;
            }
            // ... on the same bins (each pipeline carries its own copy of the coordinates)
            const PipeDev& Pp = e->pipes[d.d.pipe_peak];
            const PipeDev& Ps = e->pipes[d.d.pipe_smooth];
            if (Pp.split_evol != Ps.split_evol || Pp.n != Ps.n) { plain = false; lean = false; }
            for (const std::vector<double>* h : {&e->h_r, &e->h_mu_c, &e->h_lnrelz, &e->h_lnrelz2, &e->h_growth})
                if ((plain || lean) && std::memcmp(h->data() + Pp.coord_off, h->data() + Ps.coord_off, (size_t)Pp.n * sizeof(double)) != 0) { plain = false; lean = false; }
            d.plain_pair = plain ? 1 : 0;
            it->lean_pair = lean;
        }
        const int nq = d.nq, nqp = d.nq_pad;
        // reference vector x0' = [vec(theta_ref) ; (1 + bao) c_j(theta_ref)]
        std::vector<double> x0((size_t)nqp, 0.0);
        HIP_OK(hipMemcpy(x0.data(), it->vec.p, (size_t)d.d.n_model * sizeof(double), hipMemcpyDeviceToHost));
        for (int j = 0; j < na; ++j) x0[d.d.n_model + j] = (e->direct ? 1.0 : 1.0 + tref[d.d.bao_amp_slot]) * tref[d.q_slot[j]];
        if (it->q_x0.upload(x0.data(), x0.size())) return -2;

        DevBuf<int32_t> midx;
        if (midx.upload(it->mask_idx.data(), it->mask_idx.size())) return -2;
        DevBuf<double>& cfull = it->q_cfull;       // (kept: the linear terms of mocks that arrive later need it, vmx_fit_migrad)
        if (it->has_cinv) {
            if (cfull.n < (size_t)nm * nmp && cfull.alloc((size_t)nm * nmp, true)) return -2;
            hipLaunchKernelGGL(k_sym_from_half, dim3((nm + 255) / 256, nm), dim3(256), 0, e->stream, cfull.p, it->cinv.p, nm, nmp);
        }
        if (it->q_m0.n < (size_t)nmp && it->q_m0.alloc((size_t)nmp, true)) return -2;
        hipLaunchKernelGGL(k_mask_gather, dim3((nm + 255) / 256), dim3(256), 0, e->stream, it->q_m0.p, (const double*)(e->model.p + d.model_off), midx.p, nm);
        if (e->quad_factored && (e->quad_mat_dirty || it->q_f.p == nullptr)) {
            // F = U X^T with C^-1 = U^T U: the Cholesky factor of the inverse covariance on the host (vmx_plan.h; n^3 / 3 flops,
            // a second for n = 3180 - set-up, redone only when the covariance changes)
            std::vector<double> hu((size_t)nm * nmp, 0.0);
            if (it->has_cinv) {
                std::vector<double> hc((size_t)nm * nmp);
                HIP_OK(hipStreamSynchronize(e->stream));
                HIP_OK(hipMemcpy(hc.data(), cfull.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost));
                if (!vmx_plan::cholesky_lower(hc.data(), nm, nmp)) {
                    std::fprintf(stderr, "[vegamx] the inverse covariance of an item is not positive definite: chi2 keeps the Q' form\n");
                    e->quad_kind = 1;
                    return quad_build(e);
                }
                for (int i = 0; i < nm; ++i)
                    for (int m = 0; m <= i; ++m) hu[(size_t)m * nmp + i] = hc[(size_t)i * nmp + m];       // U = L^T
            } else
                for (int i = 0; i < nm; ++i) hu[(size_t)i * nmp + i] = 1.0;
            if (it->q_u.p == nullptr && it->q_u.alloc((size_t)nm * nmp, true)) return -2;
            HIP_OK(hipMemcpy(it->q_u.p, hu.data(), hu.size() * sizeof(double), hipMemcpyHostToDevice));
            DevBuf<double> X;
            DevBuf<int64_t>& bo = it->q_basis_off;
            boff.push_back(0);
            if (bo.upload(boff.data(), boff.size())) return -2;
            if (X.alloc((size_t)nq * nmp, true)) return -2;
            hipLaunchKernelGGL(k_quad_gather, dim3((nm + 255) / 256, nq), dim3(256), 0, e->stream, X.p, nmp,
                               it->has_dm ? it->dm.p : (const double*)nullptr, d.n_model_pad, midx.p, nm, (int)d.d.n_model, nq,
                               e->bb_basis.p, bo.p, (int)d.d.n_dist, it->has_csr ? 1 : 0);
            if (it->has_csr)
                hipLaunchKernelGGL(k_quad_gather_csr, dim3(nm), dim3(256), 0, e->stream, X.p, nmp, it->csr_ptr.p, it->csr_idx.p,
                                   it->csr_val.p, midx.p, nm);
            // F[m][j] = sum_i X[j][i] U[m][i]  (allocated once: captured graphs hold the pointer)
            if (it->q_f.p == nullptr && it->q_f.alloc((size_t)nm * nqp, true)) return -2;
            launch_product(e, KC_OTHER, X.p, nmp, 0, nq, nmp, it->q_u.p, nmp, 0, nm, it->q_f.p, nqp, 0, 1, nm);
            HIP_OK(hipStreamSynchronize(e->stream));        // X is released here
        }
        if (!e->quad_factored && (e->quad_mat_dirty || it->q_mat.p == nullptr)) {
            DevBuf<double> X, qfull;
            DevBuf<int64_t>& bo = it->q_basis_off;
            boff.push_back(0);
            if (bo.upload(boff.data(), boff.size())) return -2;
            if (X.alloc((size_t)nq * nmp, true)) return -2;
            hipLaunchKernelGGL(k_quad_gather, dim3((nm + 255) / 256, nq), dim3(256), 0, e->stream, X.p, nmp,
                               it->has_dm ? it->dm.p : (const double*)nullptr, d.n_model_pad, midx.p, nm, (int)d.d.n_model, nq,
                               e->bb_basis.p, bo.p, (int)d.d.n_dist, it->has_csr ? 1 : 0);
            if (it->has_csr)
                hipLaunchKernelGGL(k_quad_gather_csr, dim3(nm), dim3(256), 0, e->stream, X.p, nmp, it->csr_ptr.p, it->csr_idx.p,
                                   it->csr_val.p, midx.p, nm);
            // (allocated once: captured graphs hold these pointers, and the sizes never change)
            if (it->q_w.p == nullptr && it->q_w.alloc((size_t)nq * nmp, true)) return -2;
            if (it->has_cinv)
                launch_product(e, KC_OTHER, cfull.p, nmp, 0, nm, nmp, X.p, nmp, 0, nq, it->q_w.p, nmp, 0, 1, nq);
            else
                HIP_OK(hipMemcpyAsync(it->q_w.p, X.p, (size_t)nq * nmp * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
            if (qfull.alloc((size_t)nq * nqp, true)) return -2;
            launch_product(e, KC_OTHER, X.p, nmp, 0, nq, nmp, it->q_w.p, nmp, 0, nq, qfull.p, nqp, 0, 1, nq);
            if (it->q_mat.p == nullptr && it->q_mat.alloc((size_t)nq * nqp, true)) return -2;
            hipLaunchKernelGGL(k_half_from_full, dim3((nqp + 255) / 256, nq), dim3(256), 0, e->stream, it->q_mat.p, qfull.p, nq, nqp);
            HIP_OK(hipStreamSynchronize(e->stream));        // X / qfull are released here
        }
        // linear terms for the data vector and every mock of the pool
        const int rows = 1 + it->n_mocks;
        DevBuf<double> R0, T;
        if (R0.alloc((size_t)rows * nmp, true)) return -2;
        hipLaunchKernelGGL(k_quad_rows, dim3((nm + 255) / 256, rows), dim3(256), 0, e->stream, R0.p, nmp,
                           (const double*)(e->model.p + d.model_off), midx.p, it->data.p,
                           it->n_mocks ? it->mock_pool.p : (const double*)nullptr, nm, rows);
        if (it->q_lin.alloc((size_t)rows * nqp, true) || it->q_c0.alloc(rows, true)) return -2;
        if (e->quad_factored) {
            // u0 = U r0 per data vector / mock
            if (it->q_u0.alloc((size_t)rows * nmp, true)) return -2;
            launch_product(e, KC_OTHER, it->q_u.p, nmp, 0, nm, nmp, R0.p, nmp, 0, rows, it->q_u0.p, nmp, 0, 1, rows);
            if (it->q_y.n < (size_t)e->fact_slab_rows * nmp && it->q_y.alloc((size_t)e->fact_slab_rows * nmp, true)) return -2;
            d.q_u0 = it->q_u0.p; d.q_y = it->q_y.p;
        } else
        launch_product(e, KC_OTHER, it->q_w.p, nmp, 0, nq, nmp, R0.p, nmp, 0, rows, it->q_lin.p, nqp, 0, 1, rows);
        if (e->quad_factored) {}
        else if (it->has_cinv) {
            if (T.alloc((size_t)rows * nmp, true)) return -2;
            launch_product(e, KC_OTHER, cfull.p, nmp, 0, nm, nmp, R0.p, nmp, 0, rows, T.p, nmp, 0, 1, rows);
            hipLaunchKernelGGL(k_rowdot, dim3(rows), dim3(256), 0, e->stream, it->q_c0.p, R0.p, T.p, nmp, nm);
        } else {
            hipLaunchKernelGGL(k_rowdot, dim3(rows), dim3(256), 0, e->stream, it->q_c0.p, R0.p, R0.p, nmp, nm);
        }
        it->q_rows = rows;
        if (it->q_x.n < (size_t)e->max_batch * nqp && it->q_x.alloc((size_t)e->max_batch * nqp, true)) return -2;
        if (!e->quad_factored && it->q_z.n < (size_t)e->slab_rows * nqp && it->q_z.alloc((size_t)e->slab_rows * nqp, true)) return -2;
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(e->stream));
        it->h_q_c0.resize(rows);
        HIP_OK(hipMemcpy(it->h_q_c0.data(), it->q_c0.p, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost));
        d.q_x0 = it->q_x0.p; d.q_lin = it->q_lin.p; d.q_c0 = it->q_c0.p; d.q_x = it->q_x.p; d.q_z = it->q_z.p;
    }
    std::vector<ItemDev> items;
    for (auto* it : e->items) items.push_back(it->dev);
    HIP_OK(hipMemcpy(e->d_items.p, items.data(), items.size() * sizeof(ItemDev), hipMemcpyHostToDevice));
    e->quad_mat_dirty = false;
    e->quad_lin_dirty = false;
    e->last_B = 0;                  // the reference evaluation is not a caller's evaluation
    // graphs captured before hold the previous per-mock buffers (q_lin / q_c0 grow with the pool): capture again
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second);
    e->graphs.clear();
    return 0;
}

// chi2-only evaluations take the quadratic form when it applies; refreshes its tensors when data or covariances changed
static int quad_ready(vmx_engine* e, bool* use, int B)
{
    *use = false;
    if (!e->quad_eligible || e->theta_ref.empty() || e->direct) return 0;
    if (e->quad_mat_dirty || e->quad_lin_dirty) {
        if (quad_build(e)) return -2;
        if (!e->quad_eligible) return 0;
    }
    // (device allocations must not happen inside a stream capture: the work list of this batch size is built here)
    if (B > 8 && e->items.size() <= VMX_MAX_GROUP && !quad_work_list(e, B)) return -2;
    *use = true;
    return 0;
}

// The second lane of an engine: every static tensor borrowed (DevBuf copies are views), the per-batch workspace, the
// tables of the batch's shared parameters, the work lists' partial-sum buffers, stream and events its own.
static vmx_engine* clone_lane(vmx_engine* e)
{
    auto* L = new vmx_engine(*e);
    // what the copy must not share (or free)
    L->lanes.clear(); L->n_lanes = 1; L->lane_calls = 0;
    L->fitws = nullptr; L->call_mock = nullptr;
    L->stream = nullptr; L->cur = nullptr; L->aux.clear(); L->ev_join.clear(); L->ev_fork = nullptr;
    L->graphs.clear(); L->quad_lists.clear(); L->cinv_lists.clear(); L->host_allocs.clear(); L->spans.clear(); L->span_used = 0; L->profiling = false;
    L->pin_theta = nullptr; L->pin_chi2 = nullptr; L->pin_status = nullptr; L->pin_done = nullptr; L->pin_part = nullptr;
    L->dpin_theta = nullptr; L->dpin_chi2 = nullptr; L->dpin_status = nullptr; L->dpin_done = nullptr; L->dpin_part = nullptr;
    L->host_key_valid = false; L->pending_key.clear();
    std::map<MetalHost*, MetalHost*> metal_map;
    L->metals.clear();
    for (auto* m : e->metals) { auto* c = new MetalHost(*m); metal_map[m] = c; L->metals.push_back(c); }
    L->items.clear();
    for (auto* it : e->items) {
        auto* c = new ItemHost(*it);
        for (auto*& m : c->metals) m = metal_map[m];
        L->items.push_back(c);
    }
    auto fail_out = [&]() -> vmx_engine* { delete L; return nullptr; };
    if (hipStreamCreate(&L->stream) != hipSuccess) return fail_out();
    L->cur = L->stream;
    for (size_t q = 1; q < L->items.size(); ++q) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        if (hipStreamCreate(&st) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail_out();
        L->aux.push_back(st); L->ev_join.push_back(ev);
    }
    if (hipEventCreateWithFlags(&L->ev_fork, hipEventDisableTiming) != hipSuccess) return fail_out();
    // per-batch workspace (sizes as in vmx_finalize)
    const int Bm = e->max_batch, n_pipe = (int)e->pipes.size();
    const size_t ncols = (size_t)Bm * n_pipe, acols = (size_t)Bm * std::max(e->n_active, 1);
    for (auto* b : {&L->theta, &L->scal, &L->metal_bias, &L->pl, &L->coef, &L->xi, &L->xim, &L->model, &L->chi2, &L->xtab, &L->xtab_k,
                    &L->xtab_key, &L->gres, &L->gz, &L->mv_part, &L->pk_direct}) b->forget();
    for (auto* b : {&L->status, &L->k_live, &L->coef_win}) b->forget();
    L->gemm_trace.forget(); L->pk_trace.forget(); L->d_items.forget();
    if (L->theta.alloc((size_t)Bm * e->n_params) || L->scal.alloc(ncols * VMX_NS) ||
        L->metal_bias.alloc((size_t)Bm * 3 * (e->metals.size() + 1)) ||
        L->pl.alloc((size_t)VMX_MAX_ELL * acols * e->nkp) || L->coef.alloc((size_t)VMX_MAX_ELL * acols * e->ncp) ||
        L->xi.alloc((size_t)e->xi_total) || L->xim.alloc((size_t)e->xim_total) ||
        L->model.alloc((size_t)Bm * e->model_size) || L->chi2.alloc(Bm) || L->status.alloc(Bm) || L->k_live.alloc(8)) return fail_out();
    {
        const int32_t empty_window[2] = {0x7fffffff, -1};
        if (L->coef_win.upload(empty_window, 2)) return fail_out();
    }
    if (e->n_xtab > 0) {
        std::vector<double> key((size_t)2 * e->n_xtab * VMX_XTAB_KEY + 1, std::nan(""));
        if (L->xtab.alloc(((size_t)e->n_xtab * 2 * e->n_rows + 128) * e->nkp, true) || L->xtab_k.alloc((size_t)e->n_xtab * 4 * e->nkp, true) ||
            L->xtab_key.upload(key.data(), key.size())) return fail_out();
    }
    if (e->gcinv.p && (L->gres.alloc((size_t)Bm * e->g_ld, true) || L->gz.alloc((size_t)e->slab_rows * e->g_ld, true))) return fail_out();
    std::vector<ItemDev> items;
    for (auto* it : L->items) {
        ItemDev& d = it->dev;
        for (auto* b : {&it->vec, &it->dist, &it->res, &it->z, &it->marg_out, &it->q_x, &it->q_z, &it->q_y}) {
            const size_t n = b->n;
            b->forget();
            if (n > 0 && b != &it->marg_out && b->alloc(n, true)) return fail_out();
        }
        d.vec = it->vec.p; d.dist = it->dist.p; d.res = it->res.p; d.z = it->z.p; d.q_x = it->q_x.p; d.q_z = it->q_z.p; d.q_y = it->q_y.p;
        items.push_back(d);
    }
    if (L->d_items.upload(items.data(), items.size())) return fail_out();
    EngineDev& D = L->dev;
    D.xtab = L->xtab.p; D.xtab_key = L->xtab_key.p; D.xtab_k = L->xtab_k.p;
    D.items = L->d_items.p;
    D.theta = L->theta.p; D.scal = L->scal.p; D.metal_bias = L->metal_bias.p; D.pl = L->pl.p; D.coef = L->coef.p;
    D.xi = L->xi.p; D.xim = L->xim.p; D.model = L->model.p; D.chi2 = L->chi2.p; D.status = L->status.p; D.k_live = L->k_live.p;
    D.coef_win = L->coef_win.p; D.pk_trace = nullptr; D.gres = L->gres.p; D.gz = L->gz.p; D.pk_direct = nullptr;
    L->direct = false;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return fail_out();
    return L;
}

// vmx_eval_device and its variants.  d_mock: this call's walkers are compared with those rows of the mock pools (device
// pointer, [B]; nullptr: the rows of vmx_set_mock_index); eager: no captured graph for small batches (a caller whose batch size
// changes from call to call - the fit driver - would capture one per size).
static int eval_device_impl(vmx_engine* e, const double* d_theta, int32_t B, double* d_chi2, double* d_model, int32_t* d_status,
                            const int32_t* d_mock, bool eager)
{
    e->host_key_valid = false;          // (device-resident walkers may rebuild the tables: the host no longer knows their key)
    bool quad = false;
    if (!d_model && quad_ready(e, &quad, B)) return -2;
    e->last_stream = e->stream;
    // two lanes: independent chi2-only batches alternate between this engine's workspace and its clone's, each on its own
    // stream - the kernels of one batch fill the first and last block rounds of the other's (every kernel of a B = 256
    // chain spends 20 - 30 % of its launch there).  Anything else waits for the second lane first.
    const bool two_lanes = e->n_lanes > 1 && quad && !d_model && e->blind_scale.empty() && B >= 64 && !e->direct &&
                           !(e->profiling && e->prof_mask == 0xffffffffu);
    struct MockScope {          // the per-call rows apply to the engine (or lane) that runs this call, and to this call only
        vmx_engine* x; MockScope(vmx_engine* x_, const int32_t* m) : x(x_) { x->call_mock = m; } ~MockScope() { x->call_mock = nullptr; }
    };
    if (!two_lanes) wait_lane(e);
    else if (const int which = (int)(e->lane_calls++ % e->n_lanes)) {
        while ((int)e->lanes.size() < which) {
            vmx_engine* made = clone_lane(e);
            if (!made) return fail(-2, "could not create another lane");
            e->lanes.push_back(made);
        }
        vmx_engine* L = e->lanes[which - 1];
        L->const_hint = e->const_hint;
        bool lq = false;
        if (quad_ready(L, &lq, B)) return -2;       // (its work lists: the tensors are the borrowed ones)
        if (!lq) return fail(-2, "the second lane cannot take the quadratic form");
        const int tab = (B >= 16 && L->n_xtab > 0) ? L->const_hint : 0;
        MockScope scope(L, d_mock);
        if (run_chain(L, B, tab, false, d_theta, d_chi2, d_status, true)) return -2;
        e->last_stream = L->stream;
        return 0;
    }
    MockScope scope(e, d_mock);
    if (!e->blind_scale.empty()) {
        // parameter-level blinding: the walkers are transformed in the engine's own copy
        HIP_OK(hipMemcpyAsync(e->theta.p, d_theta, (size_t)B * e->n_params * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        const int n = B * e->n_params;
        hipLaunchKernelGGL(k_theta_affine, dim3((n + 255) / 256), dim3(256), 0, e->stream, e->theta.p, e->d_blind.p, e->n_params, n);
        const int tab = (B >= 16 && e->n_xtab > 0) ? e->const_hint : 0;
        if (eager ? run_chain(e, B, tab, false, nullptr, nullptr, nullptr, quad) : run_chain_cached(e, B, B >= 16 ? e->const_hint : 0, false, quad)) return -2;
        if (d_chi2) HIP_OK(hipMemcpyAsync(d_chi2, e->chi2.p, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        if (d_status) HIP_OK(hipMemcpyAsync(d_status, e->status.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, e->stream));
    } else if (B >= 64 || !e->use_graphs || e->profiling || eager) {
        // large batches: eager launches cost nothing next to the kernels, and they let the chain read / write the
        // caller's buffers directly (a captured graph would pin their addresses)
        const int tab = (B >= 16 && e->n_xtab > 0) ? e->const_hint : 0;
        if (run_chain(e, B, tab, false, d_theta, d_chi2, d_status, quad)) return -2;
    } else {
        HIP_OK(hipMemcpyAsync(e->theta.p, d_theta, (size_t)B * e->n_params * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        if (run_chain_cached(e, B, B >= 16 ? e->const_hint : 0, false, quad)) return -2;
        if (d_chi2) HIP_OK(hipMemcpyAsync(d_chi2, e->chi2.p, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        if (d_status) HIP_OK(hipMemcpyAsync(d_status, e->status.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, e->stream));
    }
    if (d_model) HIP_OK(hipMemcpyAsync(d_model, e->model.p, (size_t)B * e->model_size * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    return 0;
}

int vmx_eval_device(vmx_engine* e, const double* d_theta, int32_t B, double* d_chi2, double* d_model,
                    int32_t* d_status)
{
    REQUIRE(e && e->finalized && d_theta, "vmx_eval_device");
    REQUIRE(B > 0 && B <= e->max_batch, "batch exceeds max_batch");
    HIP_OK(hipSetDevice(e->device));
    return eval_device_impl(e, d_theta, B, d_chi2, d_model, d_status, nullptr, false);
}

int vmx_eval_device_mocks(vmx_engine* e, const double* d_theta, int32_t B, double* d_chi2, int32_t* d_status,
                          const int32_t* d_mock_index)
{
    REQUIRE(e && e->finalized && d_theta && d_mock_index, "vmx_eval_device_mocks");
    REQUIRE(B > 0 && B <= e->max_batch, "batch exceeds max_batch");
    HIP_OK(hipSetDevice(e->device));
    return eval_device_impl(e, d_theta, B, d_chi2, nullptr, d_status, d_mock_index, true);
}

// ---- fits where the walkers live (vmx_fit.h)
static_assert(sizeof(vmx_fit_stage) == sizeof(vmx_migrad::StageSpec) && sizeof(vmx_fit_spec) == sizeof(vmx_migrad::Spec),
              "include/vegamx.h and vmx_migrad.h describe the same fit");
static_assert(VMX_FIT_MAXN == vmx_migrad::MAXN && VMX_FIT_MAX_STAGES == vmx_migrad::MAX_STAGES, "fit limits");

int vmx_fit_migrad(vmx_engine* e, const vmx_fit_spec* spec, int32_t n_fits, const double* theta0, const int32_t* mock_row,
                   const vmx_fit_options* opt, vmx_fit_result* results, vmx_fit_stats* stats)
{
    REQUIRE(e && e->finalized && spec && theta0 && results && n_fits > 0, "vmx_fit_migrad");
    REQUIRE(spec->n_stages >= 1 && spec->n_stages <= VMX_FIT_MAX_STAGES && spec->n_params == e->n_params, "vmx_fit_migrad: stages / parameter columns");
    REQUIRE(spec->iterate >= 1 && spec->maxfcn > 0 && spec->up > 0.0 && spec->tol > 0.0, "vmx_fit_migrad: iterate, maxfcn, up, tol");
    int max_req = 1, n_max = 1;
    for (int s = 0; s < spec->n_stages; ++s) {
        const vmx_fit_stage& st = spec->stage[s];
        const vmx_fit_result& r = results[s];
        REQUIRE(r.x && r.ext && r.V && r.fval && r.edm && r.flags && r.nfcn && r.n_iter, "vmx_fit_migrad: result arrays of every stage");
        n_max = std::max(n_max, (int)st.n);
        REQUIRE(st.n >= 1 && st.n <= VMX_FIT_MAXN, "vmx_fit_migrad: 1 .. 32 free parameters per stage");
        for (int i = 0; i < st.n; ++i) {
            REQUIRE(st.col[i] >= 0 && st.col[i] < e->n_params, "vmx_fit_migrad: parameter column");
            for (int j = 0; j < i; ++j) REQUIRE(st.col[j] != st.col[i], "vmx_fit_migrad: a column is listed twice");
            REQUIRE(!(st.has_lo[i] && st.has_hi[i]) || st.hi[i] > st.lo[i], "vmx_fit_migrad: limits");
            REQUIRE(st.err[i] > 0.0, "vmx_fit_migrad: step sizes must be positive");
        }
        max_req = std::max(max_req, vmx_migrad::max_request(st.n));
    }
    const vmx_mock_stream* ms = opt ? opt->mocks : nullptr;
    int wave = 0, n_waves = 0;
    std::vector<int64_t> draw_off;          // column of every item's draws in a mock's row
    if (ms) {
        // mocks made while the fits run: fit f is fitted to pool row f, admitted with its wave
        REQUIRE(ms->n_mocks == n_fits && ms->draws && ms->n_drawn && !e->gcinv.p, "vmx_fit_migrad: a mock stream makes one mock per fit (no global covariance)");
        int64_t off = 0;
        for (auto* it : e->items) {
            REQUIRE(it->has_factor && it->has_mask, "vmx_fit_migrad: vmx_item_set_mock_factor for every item first");
            draw_off.push_back(off);
            off += it->dev.n_masked;
        }
        REQUIRE(ms->stride >= off, "vmx_fit_migrad: the draws of a mock are its items' side by side");
        wave = ms->wave > 0 ? ms->wave : 64;
        n_waves = (n_fits + wave - 1) / wave;
        if (mock_row) for (int f = 0; f < n_fits; ++f) REQUIRE(mock_row[f] == f, "vmx_fit_migrad: with a mock stream fit f takes pool row f");
    }
    else if (mock_row)
        for (int f = 0; f < n_fits; ++f)
            for (auto* it : e->items) REQUIRE(mock_row[f] < 0 || mock_row[f] < it->n_mocks, "vmx_fit_migrad: mock row exceeds the pool");
    const int chunk = std::max(1, std::min(opt && opt->chunk > 0 ? opt->chunk : 512, e->max_batch));
    const int want_lanes = opt && opt->lanes > 0 ? std::min(opt->lanes, VMX_MAX_LANES) : 2;
    int hint = opt ? opt->const_hint : -1;
    REQUIRE(hint >= -1 && hint <= 2, "vmx_fit_migrad: const_hint -1 (derive it), 0, 1 or 2");
    if (hint < 0) {
        // the table level the rows of a round allow (vmx_set_constant_nl_hint), as vmx_eval derives it from host walkers: a column
        // varies when a stage frees it or the fits' rows differ in it
        std::vector<char> varies(e->n_params, 0);
        for (int s = 0; s < spec->n_stages; ++s)
            for (int i = 0; i < spec->stage[s].n; ++i) varies[spec->stage[s].col[i]] = 1;
        for (int f = 1; f < n_fits; ++f)
            for (int c = 0; c < e->n_params; ++c)
                if (theta0[(size_t)f * e->n_params + c] != theta0[c]) varies[c] = 1;
        hint = e->n_xtab > 0 ? (e->no_tab2 ? 1 : 2) : 0;
        for (int slot : e->const_slots) if (varies[slot]) hint = 0;
        for (int slot : e->const_slots2) if (hint == 2 && varies[slot]) hint = 1;
    }
    HIP_OK(hipSetDevice(e->device));
    const auto t_begin = std::chrono::steady_clock::now();
    wait_lane(e);
    HIP_OK(hipStreamSynchronize(e->stream));

    const int F = n_fits, P = e->n_params;
    const size_t cap = (size_t)F * max_req;
    std::vector<int32_t> own_rows;
    if (ms) {
        // pools of n_fits rows, to be filled wave by wave (rows that have not arrived are zeros: nothing reads them before their
        // fits are admitted); the quadratic form's per-mock terms are sized for them by the build the first evaluation triggers
        // (a run of the same size as the last one finds its pools, the form's per-mock rows and the second lane as they are: the
        // waves overwrite the rows their fits read, nothing reads a row before its wave has come)
        bool resized = false;
        for (auto* it : e->items)
            if (it->mock_pool.n < (size_t)n_fits * it->dev.n_masked || it->n_mocks != n_fits || !it->dev.mock_pool) resized = true;
        if (resized) drop_lane(e);
        for (size_t q = 0; q < e->items.size(); ++q) {
            ItemHost* it = e->items[q];
            const size_t need = (size_t)n_fits * it->dev.n_masked;
            if (resized) {
                if (it->mock_pool.n < need && it->mock_pool.alloc(need, true)) return -2;
                it->n_mocks = n_fits;
                it->dev.mock_pool = it->mock_pool.p;
                HIP_OK(hipMemcpy(e->d_items.p + q, &it->dev, sizeof(ItemDev), hipMemcpyHostToDevice));
                e->quad_lin_dirty = true;
            }
            const size_t rows = (size_t)wave * it->dev.n_masked_pad;
            if (ensure(it->mc_z, rows) || ensure(it->mc_noise, rows) || ensure(it->mc_r0, rows) || ensure(it->mc_t, rows)) return -2;
            HIP_OK(hipMemset(it->mc_z.p, 0, rows * sizeof(double)));       // (the pad columns of the draws stay zero)
        }
        own_rows.resize(n_fits);
        for (int f = 0; f < n_fits; ++f) own_rows[f] = f;
        mock_row = own_rows.data();
    }
    if (!e->fitws) e->fitws = new FitWorkspace();
    FitWorkspace& W = *e->fitws;
    const int fit_cap = n_max <= 4 ? 4 : n_max <= 8 ? 8 : n_max <= 16 ? 16 : 32;
    const size_t state_bytes = fit_state_bytes(fit_cap);
    if (ensure(W.state, (size_t)F * state_bytes / sizeof(double)) || ensure(W.done, F) || ensure(W.spec, 1) || ensure(W.base, (size_t)F * P) || ensure(W.theta, cap * P) || ensure(W.chi2, cap) ||
        ensure(W.mock_row, F) || ensure(W.count, F) || ensure(W.offset, (size_t)F + 1) || ensure(W.mock, cap) || ensure(W.status, cap)) return -2;
    for (int s = 0; s < spec->n_stages; ++s) {
        const size_t n = spec->stage[s].n;
        if (ensure(W.ox[s], F * n) || ensure(W.oext[s], F * n) || ensure(W.oV[s], F * n * n) || ensure(W.ofval[s], F) || ensure(W.oedm[s], F) ||
            ensure(W.oflags[s], F) || ensure(W.oiter[s], F) || ensure(W.onfcn[s], F)) return -2;
    }
    if (!W.pin_word) {
        HIP_OK(hipHostMalloc((void**)&W.pin_word, 16 * sizeof(int32_t), hipHostMallocMapped));
        HIP_OK(hipHostGetDevicePointer((void**)&W.dpin_word, W.pin_word, 0));
        HIP_OK(hipEventCreateWithFlags(&W.ev_lane, hipEventDisableTiming));
    }
    hipStream_t st = e->stream;
    HIP_OK(hipMemsetAsync(W.state.p, 0, (size_t)F * state_bytes, st));       // (all zeros = a fit at its start)
    HIP_OK(hipMemsetAsync(W.done.p, 0, (size_t)F * sizeof(int32_t), st));
    HIP_OK(hipMemsetAsync(W.offset.p, 0, ((size_t)F + 1) * sizeof(int32_t), st));
    HIP_OK(hipMemcpyAsync(W.spec.p, spec, sizeof(vmx_fit_spec), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(W.base.p, theta0, (size_t)F * P * sizeof(double), hipMemcpyHostToDevice, st));
    if (mock_row) HIP_OK(hipMemcpyAsync(W.mock_row.p, mock_row, (size_t)F * sizeof(int32_t), hipMemcpyHostToDevice, st));

    FitDev D{};
    D.state = W.state.p; D.spec = W.spec.p; D.base = W.base.p; D.mock_row = mock_row ? W.mock_row.p : nullptr;
    for (int s = 0; s < spec->n_stages; ++s)
        D.out[s] = vmx_migrad::StageOut{W.ox[s].p, W.oext[s].p, W.oV[s].p, W.ofval[s].p, W.oedm[s].p, W.oflags[s].p, W.onfcn[s].p, W.oiter[s].p};
    D.count = W.count.p; D.offset = W.offset.p; D.done = W.done.p; D.theta = W.theta.p; D.mock = W.mock.p; D.chi2 = W.chi2.p;
    D.host_word = W.dpin_word; D.F = F; D.P = P; D.admitted = ms ? 0 : F;

    // the engine as the fits' objective: chi2-only device evaluations of the round's rows, eager launches, two lanes when the
    // quadratic form serves them; the table level the caller vouches for
    const int saved_hint = e->const_hint, saved_lanes = e->n_lanes;
    const bool saved_ring = e->ring_allowed;
    e->const_hint = hint;
    if (want_lanes > e->n_lanes) { e->n_lanes = want_lanes; e->ring_allowed = false; }
    struct Restore {
        vmx_engine* e; int hint, lanes; bool ring;
        ~Restore() { wait_lane(e); e->const_hint = hint; e->n_lanes = lanes; e->ring_allowed = ring; e->lane_calls = 0; e->last_stream = e->stream; }
    } restore{e, saved_hint, saved_lanes, saved_ring};

    vmx_fit_stats S{};
    const size_t emit_lds = (size_t)((P + 1) & ~1) * sizeof(int32_t) + vmx_migrad::MAXN * sizeof(double);
    if (fit_round(fit_cap, D, st, emit_lds, true)) return -2;
    const auto t_loop = std::chrono::steady_clock::now();
    double wait_s = 0.0;
    size_t gap_used = 0;
    int next_wave = 0;
    double producer_wait_s = 0.0, wave_host_s = 0.0;
    bool quad_form = false;
    if (ms && quad_ready(e, &quad_form, chunk)) return -2;       // (the form's tensors, with room for every mock's terms)
    for (;;) {
        if (ms && next_wave < n_waves) {
            // Wave `next_wave` joins at this round - a fixed schedule (one wave per round from the start), so that the rounds'
            // batches, hence every bit of every chi2, do not depend on how fast the draws arrive; the host waits for the producer
            // when it is behind.  The wave's mocks = fiducial + L . draws with the chain's product kernels (reference
            // vega/data.py:751-753), then their rows of the quadratic form's linear terms and constants, all on the stream
            // ahead of the round's bookkeeping.
            const int a = next_wave * wave, b = std::min(n_fits, a + wave), cnt = b - a;
            const auto t_p0 = std::chrono::steady_clock::now();
            const double limit = ms->timeout_seconds > 0.0 ? ms->timeout_seconds : 600.0;
            while (*ms->n_drawn < b) {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p0).count() > limit)
                    return fail(-2, "vmx_fit_migrad: the producer of the mock draws stalled");
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            producer_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p0).count();
            std::atomic_thread_fence(std::memory_order_acquire);
            e->cur = st;
            for (size_t q = 0; q < e->items.size(); ++q) {
                ItemHost* it = e->items[q];
                const ItemDev& d = it->dev;
                const int nm = d.n_masked, nmp = d.n_masked_pad;
                HIP_OK(hipMemcpy2DAsync(it->mc_z.p, (size_t)nmp * sizeof(double), ms->draws + (size_t)a * ms->stride + draw_off[q],
                                        (size_t)ms->stride * sizeof(double), (size_t)nm * sizeof(double), cnt, hipMemcpyHostToDevice, st));
                launch_product(e, KC_OTHER, it->mc_chol.p, nmp, 0, nm, nmp, it->mc_z.p, nmp, 0, cnt, it->mc_noise.p, nmp, 0, 1, cnt);
                hipLaunchKernelGGL(k_mock_rows, dim3((nm + 255) / 256, cnt), dim3(256), 0, st, it->mock_pool.p + (size_t)a * nm,
                                   quad_form ? it->mc_r0.p : (double*)nullptr, nmp, (const double*)it->mc_noise.p, (const double*)it->mc_fid.p,
                                   (const double*)it->q_m0.p, nm, cnt);
                if (!quad_form) continue;
                if (e->quad_factored)
                    launch_product(e, KC_OTHER, it->q_u.p, nmp, 0, nm, nmp, it->mc_r0.p, nmp, 0, cnt, it->q_u0.p + (size_t)(1 + a) * nmp, nmp, 0, 1, cnt);
                else {
                    launch_product(e, KC_OTHER, it->q_w.p, nmp, 0, d.nq, nmp, it->mc_r0.p, nmp, 0, cnt, it->q_lin.p + (size_t)(1 + a) * d.nq_pad, d.nq_pad, 0, 1, cnt);
                    if (it->has_cinv) {
                        launch_product(e, KC_OTHER, it->q_cfull.p, nmp, 0, nm, nmp, it->mc_r0.p, nmp, 0, cnt, it->mc_t.p, nmp, 0, 1, cnt);
                        hipLaunchKernelGGL(k_rowdot, dim3(cnt), dim3(256), 0, st, it->q_c0.p + 1 + a, (const double*)it->mc_r0.p, (const double*)it->mc_t.p, nmp, nm);
                    } else
                        hipLaunchKernelGGL(k_rowdot, dim3(cnt), dim3(256), 0, st, it->q_c0.p + 1 + a, (const double*)it->mc_r0.p, (const double*)it->mc_r0.p, nmp, nm);
                }
            }
            HIP_OK(hipGetLastError());
            next_wave += 1;
            D.admitted = b;
            wave_host_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p0).count();
        }
        {
            const auto t_r0 = std::chrono::steady_clock::now();
            if (fit_round(fit_cap, D, st, emit_lds, false)) return -2;
            S.seconds_enqueuing_rounds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_r0).count();
        }
        if (gap_used + 2 > W.ev_gap.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
            W.ev_gap.push_back(a); W.ev_gap.push_back(b);
        }
        HIP_OK(hipEventRecord(W.ev_gap[gap_used], st));
        const auto t_w0 = std::chrono::steady_clock::now();
        HIP_OK(hipStreamSynchronize(st));
        wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count();
        const int total = W.pin_word[0];
        if (total <= 0 && (!ms || next_wave >= n_waves)) break;
        if (total <= 0) { HIP_OK(hipEventRecord(W.ev_gap[gap_used + 1], st)); gap_used += 2; continue; }      // (nothing asked for yet: the next wave joins)
        if ((size_t)total > cap) return fail(-2, "vmx_fit_migrad: a round asked for more rows than its buffers hold");
        HIP_OK(hipEventRecord(W.ev_gap[gap_used + 1], st));
        gap_used += 2;
        S.rounds += 1;
        S.evaluations += total;
        bool lane_used = false;
        hipStream_t lane_stream = nullptr;
        const auto t_c0 = std::chrono::steady_clock::now();
        for (int off = 0; off < total; off += chunk) {
            const int B = std::min(chunk, total - off);
            if (eval_device_impl(e, W.theta.p + (size_t)off * P, B, W.chi2.p + off, nullptr, W.status.p + off, mock_row ? W.mock.p + off : nullptr, true)) return -2;
            if (e->last_stream != st) { lane_used = true; lane_stream = e->last_stream; }
            S.engine_calls += 1;
            int bin = 0;
            while (bin < 7 && B > (1 << (2 * bin))) ++bin;          // 1, 2..4, 5..16, 17..64, 65..256, 257..1024, 1025..4096, more
            S.calls_by_batch[bin] += 1;
            S.evaluations_by_batch[bin] += B;
        }
        S.seconds_enqueuing_calls += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c0).count();
        if (lane_used) {        // the next round's bookkeeping reads every chunk's chi2
            HIP_OK(hipEventRecord(W.ev_lane, lane_stream));
            HIP_OK(hipStreamWaitEvent(st, W.ev_lane, 0));
        }
    }
    S.fits_unfinished = W.pin_word[1];
    S.seconds_waiting_for_draws = producer_wait_s;
    S.seconds_enqueuing_waves = wave_host_s - producer_wait_s;
    if (ms && quad_form)        // (the host copy of the constants serves the single-walker chain of vmx_eval)
        for (auto* it : e->items)
            if (!e->quad_factored && it->h_q_c0.size() == (size_t)it->q_rows)
                HIP_OK(hipMemcpy(it->h_q_c0.data(), it->q_c0.p, it->h_q_c0.size() * sizeof(double), hipMemcpyDeviceToHost));
    const auto t_done = std::chrono::steady_clock::now();
    for (int s = 0; s < spec->n_stages; ++s) {
        const size_t n = spec->stage[s].n;
        const vmx_fit_result& r = results[s];
        HIP_OK(hipMemcpy(r.x, W.ox[s].p, F * n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.ext, W.oext[s].p, F * n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.V, W.oV[s].p, F * n * n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.fval, W.ofval[s].p, F * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.edm, W.oedm[s].p, F * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.flags, W.oflags[s].p, F * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.nfcn, W.onfcn[s].p, F * sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(r.n_iter, W.oiter[s].p, F * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    double idle_ms = 0.0;
    for (size_t i = 0; i + 1 < gap_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, W.ev_gap[i], W.ev_gap[i + 1]) == hipSuccess) idle_ms += t;
    }
    const auto t_end = std::chrono::steady_clock::now();
    S.seconds = std::chrono::duration<double>(t_end - t_begin).count();
    S.seconds_setup = std::chrono::duration<double>(t_loop - t_begin).count();
    S.seconds_rounds = std::chrono::duration<double>(t_done - t_loop).count();
    S.seconds_host_waiting = wait_s;
    S.gpu_idle_seconds_between_rounds = idle_ms * 1e-3;
    if (stats) *stats = S;
    return 0;
}

void* vmx_stream(vmx_engine* e) { return e ? (void*)e->stream : nullptr; }
void* vmx_last_stream(vmx_engine* e) { return e ? (void*)(e->last_stream ? e->last_stream : e->stream) : nullptr; }

int vmx_set_lanes(vmx_engine* e, int32_t lanes)
{
    REQUIRE(e && e->finalized && lanes >= 1 && lanes <= VMX_MAX_LANES, "vmx_set_lanes: 1 or 2 (after vmx_finalize)");
    HIP_OK(hipSetDevice(e->device));
    if (lanes < e->n_lanes) drop_lane(e);
    e->n_lanes = lanes;
    // (the four-stage ring of the FFTLog product wins 5 us when its launch has the chip to itself; with two batches in flight
    // its 128 KB blocks keep the other lane's kernels off their CUs: 831k against 849k evaluations / s at B = 256)
    e->ring_allowed = lanes == 1;
    e->lane_calls = 0;
    return 0;
}

int vmx_set_direct_pk(vmx_engine* e, const double* pk, int32_t B, int32_t nk)
{
    REQUIRE(e && e->finalized, "vmx_set_direct_pk");
    drop_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (!pk) { e->direct = false; e->dev.pk_direct = nullptr; return 0; }
    REQUIRE(B > 0 && B <= e->max_batch && nk == e->nk, "direct_pk shape: [B <= max_batch][nk]");
    for (auto* m : e->metals)
        REQUIRE(!(m->dev.d.in_direct && m->dev.d.pipeline >= 0 && e->pipes[m->dev.d.pipeline].poly_basis >= 0),
                "direct_pk: a metal pair that enters the direct model sits on a static basis of the template's spectra - build the "
                "engine with vmx_set_static_poly(e, 0)");
    // (every check before any state changes: a refused call leaves the engine as it was)
    for (auto& p : e->pipes)
        REQUIRE(!(p.odd_rel || p.odd_asy) || p.odd_dyn_off >= 0,
                "direct_pk: a pipeline with odd-multipole terms needs their operator form (vmx_pipeline_set_odd_operator)");
    if (e->pk_direct.n < (size_t)e->max_batch * e->nkp && e->pk_direct.alloc((size_t)e->max_batch * e->nkp, true)) return -2;
    HIP_OK(hipMemcpy2D(e->pk_direct.p, (size_t)e->nkp * sizeof(double), pk, (size_t)nk * sizeof(double),
                       (size_t)nk * sizeof(double), B, hipMemcpyHostToDevice));
    // the odd-multipole terms read the caller's spectrum too: the walkers' spline coefficients = operator . spectrum
    for (auto& kv : e->odd_op_off) {
        const PipeDev& p = e->pipes[kv.first];
        if (vmx_matvec_device(e, e->odd_op.p + kv.second, 4 * p.odd_ncoef, e->nkp, e->pk_direct.p, B, e->odd_dyn.p + p.odd_dyn_off)) {
            e->direct = false; e->dev.pk_direct = nullptr;      // (a failed product: back to the template)
            return -2;
        }
    }
    e->direct = true;
    e->dev.pk_direct = e->pk_direct.p;       // every kernel of the chain takes its EngineDev from e->dev
    return 0;
}

int vmx_set_linear_spectra(vmx_engine* e, const double* pk_peak, const double* pk_smooth, const double* pk_full, int32_t nk)
{
    REQUIRE(e && e->finalized && pk_peak && pk_smooth && pk_full, "vmx_set_linear_spectra (after vmx_finalize)");
    drop_lane(e);
    REQUIRE(nk == e->nk, "linear spectra must live on the template's k grid");
    for (auto& p : e->pipes) REQUIRE(!p.odd_rel && !p.odd_asy, "the odd-multipole terms hold static splines of the template's spectra");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    const double* src[3] = {pk_peak, pk_smooth, pk_full};
    for (int i = 0; i < 3; ++i)
        HIP_OK(hipMemcpy(e->pklin.p + (size_t)i * e->nkp, src[i], (size_t)nk * sizeof(double), hipMemcpyHostToDevice));
    if (poly_basis_build(e)) return -2;
    return xi_sum_plan(e);
}

int vmx_item_set_marg_matrix(vmx_engine* e, int32_t item, const double* m, int32_t n_templates, int32_t n_masked)
{
    REQUIRE(e && !e->finalized && m, "vmx_item_set_marg_matrix (before vmx_finalize)");
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(it->has_mask && n_masked == it->dev.n_masked && n_templates > 0, "marginalisation matrix shape: [n_templates][n_masked]");
    HIP_OK(hipSetDevice(e->device));
    if (upload_padded(it->marg, m, n_templates, n_masked, it->dev.n_masked_pad)) return -2;
    it->n_templates = n_templates;
    return 0;
}

int vmx_marg_coeff(vmx_engine* e, int32_t item, double* out, int32_t B)
{
    REQUIRE(e && e->finalized && out, "vmx_marg_coeff");
    wait_lane(e);
    REQUIRE(item >= 0 && item < (int)e->items.size(), "item id");
    ItemHost* it = e->items[item];
    REQUIRE(it->n_templates > 0, "no marginalisation matrix was set for this item");
    REQUIRE(B > 0 && B == e->last_B, "vmx_marg_coeff reads the residuals of the last evaluation: B must be its batch size");
    REQUIRE(e->last_full, "the last evaluation took the quadratic chi2 form, which forms no residuals: evaluate with a model "
                          "output (or switch the form off with vmx_set_quadratic_form(e, NULL))");
    HIP_OK(hipSetDevice(e->device));
    const int ldo = vmx_pad(it->n_templates);
    if (it->marg_out.n < (size_t)e->max_batch * ldo && it->marg_out.alloc((size_t)e->max_batch * ldo, true)) return -2;
    if (vmx_matvec_device(e, it->marg.p, it->n_templates, it->dev.n_masked_pad, it->res.p, B, it->marg_out.p)) return -2;
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy2D(out, (size_t)it->n_templates * sizeof(double), it->marg_out.p, (size_t)ldo * sizeof(double),
                       (size_t)it->n_templates * sizeof(double), B, hipMemcpyDeviceToHost));
    return 0;
}

int vmx_get_mu_nodes(vmx_engine* e, double* mu, double* w, int32_t capacity)
{
    REQUIRE(e && e->nk > 0, "vmx_get_mu_nodes (after vmx_set_template)");
    if (!mu || !w || capacity < e->n_extra) return e->n_extra;
    HIP_OK(hipSetDevice(e->device));
    if (e->n_extra > 0) {
        HIP_OK(hipMemcpy(mu, e->mu.p + e->n_mu, (size_t)e->n_extra * sizeof(double), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(w, e->node_w.p, (size_t)e->n_extra * sizeof(double), hipMemcpyDeviceToHost));
    }
    return e->n_extra;
}

int vmx_set_mu_rule_box(vmx_engine* e, int32_t n, const int32_t* slots, const double* lo, const double* hi)
{
    REQUIRE(e && !e->finalized && n >= 0 && (n == 0 || (slots && lo && hi)), "vmx_set_mu_rule_box (before vmx_finalize)");
    e->rule_slot.clear(); e->rule_lo.clear(); e->rule_hi.clear();
    for (int i = 0; i < n; ++i) {
        REQUIRE(slots[i] >= 0 && lo[i] <= hi[i], "mu-rule box entry");
        e->rule_slot.push_back(slots[i]); e->rule_lo.push_back(lo[i]); e->rule_hi.push_back(hi[i]);
    }
    return 0;
}

int vmx_set_mu_quadrature(vmx_engine* e, int32_t node_rule)
{
    REQUIRE(e && e->finalized, "vmx_set_mu_quadrature (after vmx_finalize)");
    drop_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    e->mu_nodes_on = node_rule != 0 && e->n_extra > 0;
    e->dev.k_node_max = e->mu_nodes_on ? e->k_node_max : 0.0;
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second);      // (captured graphs hold the previous setting)
    e->graphs.clear();
    e->quad_lin_dirty = true;       // the reference point (x0', m0) is re-evaluated with the new rule; Q' and W do not depend on it
    return e->mu_nodes_on ? 1 : 0;
}

int vmx_set_static_poly(vmx_engine* e, int32_t enabled)
{
    REQUIRE(e && !e->finalized, "vmx_set_static_poly (before vmx_finalize)");
    e->static_poly = enabled != 0;
    return 0;
}

int vmx_set_quadratic_form_kind(vmx_engine* e, int32_t kind)
{
    REQUIRE(e && e->finalized && kind >= 0 && kind <= 2, "vmx_set_quadratic_form_kind: 0 (cheaper), 1 (Q'), 2 (factored), after vmx_finalize");
    if (kind != e->quad_kind) { e->quad_kind = kind; e->quad_mat_dirty = true; }
    return 0;
}

int vmx_set_quadratic_form(vmx_engine* e, const double* theta_ref)
{
    REQUIRE(e && e->finalized, "vmx_set_quadratic_form (after vmx_finalize)");
    drop_lane(e);
    e->theta_ref.clear();
    if (theta_ref) e->theta_ref.assign(theta_ref, theta_ref + e->n_params);
    e->quad_mat_dirty = true;       // (a new reference point: everything is rebuilt at the next chi2-only evaluation)
    return e->quad_eligible ? 1 : 0;
}

int vmx_set_constant_nl_hint(vmx_engine* e, int32_t enabled)
{
    REQUIRE(e, "vmx_set_constant_nl_hint");
    e->const_hint = enabled <= 0 ? 0 : enabled >= 2 ? 2 : 1;
    if (e->no_tab2 && e->const_hint > 1) e->const_hint = 1;
    return 0;
}

int vmx_sync(vmx_engine* e)
{
    REQUIRE(e, "vmx_sync");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    for (auto* l : e->lanes) HIP_OK(hipStreamSynchronize(l->stream));
    if (e->profiling) collect_spans(e);
    if (e->gemm_trace.p && e->gemm_trace_blocks && (getenv("VMX_GEMM_TRACE") || getenv("VMX_QUAD_TRACE"))) {
        std::vector<unsigned long long> h(4 * e->gemm_trace_blocks);
        HIP_OK(hipMemcpy(h.data(), e->gemm_trace.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const char* path = getenv("VMX_QUAD_TRACE") ? getenv("VMX_QUAD_TRACE") : getenv("VMX_GEMM_TRACE");
        if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    if (e->pk_trace.p && e->pk_trace_blocks) {
        std::vector<unsigned long long> h(4 * e->pk_trace_blocks);
        HIP_OK(hipMemcpy(h.data(), e->pk_trace.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("VMX_PK_TRACE"), "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    return 0;
}

int vmx_eval(vmx_engine* e, const double* theta, int32_t B, double* chi2, double* model, int32_t* status)
{
    REQUIRE(e && e->finalized && theta, "vmx_eval");
    wait_lane(e);
    REQUIRE(B > 0 && B <= e->max_batch, "batch exceeds max_batch");
    const auto t_begin = std::chrono::steady_clock::now();
    HIP_OK(hipSetDevice(e->device));
    // small batches are latency-bound: the first kernel reads the walkers from the mapped pinned buffer and the last
    // one stores chi2 / status there, which removes three staging copies (~25 us of a ~100 us evaluation)
    const bool zero_copy = B <= 8 && B * ((int)e->pipes.size() + 1) <= 1024 && (size_t)B * e->n_params * sizeof(double) <= 48 * 1024 &&
                           e->dpin_theta && e->dpin_chi2 && e->dpin_status;
    bool quad = false;
    if (!model && quad_ready(e, &quad, B)) return -2;
    if (e->blind_scale.empty()) std::memcpy(e->pin_theta, theta, (size_t)B * e->n_params * sizeof(double));
    else        // parameter-level blinding, applied while staging (same expression as k_theta_affine)
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < e->n_params; ++i) {
                const size_t o = (size_t)b * e->n_params + i;
                e->pin_theta[o] = e->blind_scale[i] == 1.0 ? theta[o] + e->blind_shift[i] : e->blind_scale[i] * theta[o] + e->blind_shift[i];
            }
    if (!zero_copy)
        HIP_OK(hipMemcpyAsync(e->theta.p, e->pin_theta, (size_t)B * e->n_params * sizeof(double), hipMemcpyHostToDevice, e->stream));
    // the D_NL * G table pays off once a batch shares its Arinyo parameters (checked here, on the host copy)
    int tab_mode = (e->n_xtab > 0) ? (e->no_tab2 ? 1 : 2) : 0;
    for (int b = 1; b < B && tab_mode; ++b) {
        for (int slot : e->const_slots)
            if (theta[(size_t)b * e->n_params + slot] != theta[slot]) { tab_mode = 0; break; }
        for (int slot : e->const_slots2)
            if (tab_mode == 2 && theta[(size_t)b * e->n_params + slot] != theta[slot]) { tab_mode = 1; break; }
    }
    // The tables outlive an evaluation (keyed on the shared parameters).  A small batch - a single walker of a minimiser or
    // sampler above all - runs against level-2 tables once those parameters have come twice in a row; a caller that varies
    // them from call to call keeps the per-walker loops (building tables costs more than one small evaluation).
    bool tables_current = false;
    {
        std::vector<double> cur;
        for (int slot : e->const_slots2) cur.push_back(e->pin_theta[slot]);      // (the transformed values: what the device sees)
        tables_current = tab_mode == 2 && e->host_key_valid && cur == e->host_key;
        if (B < 16 && tab_mode) {
            if (tab_mode == 2 && (tables_current || cur == e->pending_key)) tab_mode = 2;
            else { if (tab_mode == 2) e->pending_key = cur; tab_mode = 0; }
        }
        if (tab_mode == 2) { e->host_key = cur; e->host_key_valid = true; }
        else if (tab_mode == 1) e->host_key_valid = false;       // (the device tables now hold level 1)
    }
    const auto t_staged = std::chrono::steady_clock::now();
    // a single walker is latency-bound end to end: eager launches start the first kernel while the later ones are
    // still being enqueued, which a graph launch cannot (measured: 70 against 75 us per evaluation)
    e->host_reduce_items = 0;
    if (B == 1) {
        const bool by_value = zero_copy && e->n_params <= VMX_THETA_ARG_MAX;
        e->skip_xtab_once = tables_current;         // (the host knows the tables hold these parameters: no check launch)
        if (quad && by_value && e->pin_part && e->dpin_part) {
            // the slots the last products may fill (k_gemv1 MODE 2) start from a pattern no computation produces
            for (size_t q = 0; q < e->items.size() && q < VMX_MAX_GROUP; ++q) {
                const int blocks = gemv1_blocks(e->items[q]->dev.nq);
                for (int i = 0; i < blocks && i < 1024; ++i) std::memcpy(&e->pin_part[q * 1024 + i], &VMX_PART_SENTINEL, sizeof(double));
            }
            e->pin_status[0] = -1;
        }
        if (run_chain(e, B, tab_mode, zero_copy, nullptr, nullptr, nullptr, quad, by_value ? e->pin_theta : nullptr)) return -2;
    }
    else if (run_chain_cached(e, B, tab_mode, zero_copy, quad)) return -2;
    if (chi2 && !zero_copy) HIP_OK(hipMemcpyAsync(e->pin_chi2, e->chi2.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (status && !zero_copy) HIP_OK(hipMemcpyAsync(e->pin_status, e->status.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (model) HIP_OK(hipMemcpyAsync(model, e->model.p, (size_t)B * e->model_size * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    const auto t_enqueued = std::chrono::steady_clock::now();
    bool waited = false;
    if (e->host_reduce_items > 0) {
        // the chain ended with the streaming products of the quadratic form: add up what their blocks left here
        // (fixed order), the constants and the priors - vega_interface.py:316, :423-446
        // The wait is bounded by TIME (~200 us of spinning on the mapped words); when that runs out - a GPU shared with another
        // process, a profiler serialising the launches, a slow first launch - the stream is drained and the words are read
        // again: once the stream is empty they are guaranteed to be written, so only a sentinel that survives the drain is an error.
        double c = 0.0;
        bool complete = false;
        volatile int32_t* st_word = e->pin_status;
        const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        for (int attempt = 0; attempt < 2 && !complete; ++attempt) {
            if (attempt == 1) HIP_OK(hipStreamSynchronize(e->stream));
            complete = true;
            c = 0.0;
            for (int q = 0; q < e->host_reduce_items && complete; ++q) {
                const volatile uint64_t* slot = (const volatile uint64_t*)(e->pin_part + (size_t)q * 1024);
                for (int i = 0; i < e->host_reduce_blocks[q] && complete; ++i) {
                    int spin = 0;
                    while (slot[i] == VMX_PART_SENTINEL && attempt == 0) {
                        if ((++spin & 255) == 0 && std::chrono::steady_clock::now() > spin_until) break;
                    }
                    uint64_t bits = slot[i];
                    if (bits == VMX_PART_SENTINEL) { complete = false; break; }
                    double v; std::memcpy(&v, &bits, sizeof(double));
                    c += v;
                }
            }
            for (int spin = 0; complete && *st_word == -1 && attempt == 0;) {
                if ((++spin & 255) == 0 && std::chrono::steady_clock::now() > spin_until) break;
            }
            if (*st_word == -1) complete = false;
        }
        if (!complete) { e->host_reduce_items = 0; return fail(-2, "single-walker chain: the device left a result slot unwritten after the stream drained"); }
        for (int q = 0; q < e->host_reduce_items; ++q) {
            const ItemHost* it = e->items[q];
            const int mock = (it->dev.mock_pool && e->h_mock_index[0] >= 0) ? e->h_mock_index[0] : -1;
            c += it->h_q_c0[mock >= 0 ? 1 + mock : 0];
        }
        for (size_t q = 0; q < e->prior_slot.size(); ++q) {
            const double dlt = e->pin_theta[e->prior_slot[q]] - e->prior_mean[q];
            c += dlt * dlt / (e->prior_sigma[q] * e->prior_sigma[q]);
        }
        int32_t st = *st_word;
        if (!(c == c) || c > 1e300 || c < -1e300) st |= VMX_STATUS_NONFINITE;
        e->pin_status[0] = st;
        e->pin_chi2[0] = st ? 1e100 : c;
        e->host_reduce_items = 0;
        waited = true;
    }
    if (!waited && B == 1 && zero_copy && !model && e->dpin_done && !e->profiling && e->n_params <= VMX_THETA_ARG_MAX) {
        // the last kernel of a single-walker chain publishes a sequence number after chi2 / status (system-scope fence):
        // the host waits on that word in mapped memory - a few microseconds sooner than the stream's completion signal
        const int64_t want = e->done_seq;
        volatile int64_t* word = e->pin_done;
        for (int64_t spin = 0; spin < (int64_t)1 << 26 && !waited; ++spin) waited = *word == want;
    }
    if (!waited && vmx_sync(e)) return -2;
    if (chi2) std::memcpy(chi2, e->pin_chi2, (size_t)B * sizeof(double));
    if (status) std::memcpy(status, e->pin_status, (size_t)B * sizeof(int32_t));
    if (e->trace_host) {
        const auto t_end = std::chrono::steady_clock::now();
        e->host_ns[0] += std::chrono::duration<double, std::nano>(t_staged - t_begin).count();
        e->host_ns[1] += std::chrono::duration<double, std::nano>(t_enqueued - t_staged).count();
        e->host_ns[2] += std::chrono::duration<double, std::nano>(t_end - t_enqueued).count();
        ++e->host_calls;
    }
    return 0;
}

int64_t vmx_debug_read(vmx_engine* e, int32_t what, int32_t index, double* out, int64_t capacity)
{
    if (!e || !e->finalized || !out || e->last_B <= 0) { fail(-1, "invalid argument: vmx_debug_read"); return -1; }
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) { fail(-2, "hip sync"); return -2; }
    wait_lane(e);
    const int B = e->last_B;
    const double* src = nullptr; int64_t count = 0;
    if (what == 0) { src = e->pl.p; count = (int64_t)VMX_MAX_ELL * B * e->n_active * e->nkp; }
    else if (what == 1) {
        if (index < 0 || index >= (int)e->pipes.size()) { fail(-1, "invalid argument: pipeline index"); return -1; }
        if (!e->last_taps) { fail(-1, "the last evaluation (chi2 only, more than 8 walkers, no metal terms) did not store the per-pipeline bins: ask for a model"); return -1; }
        src = e->xi.p + e->pipes[index].xi_off; count = (int64_t)B * e->pipes[index].n_pad;
    } else if (what == 2) { src = e->coef.p; count = (int64_t)VMX_MAX_ELL * B * e->n_active * e->ncp; }
    else if (what == 3) {
        // correlation of metal `index` (global order of vmx_item_add_metal) after its metal matrix
        if (index < 0 || index >= (int)e->metals.size() || e->metals[index]->dev.mat_off < 0) { fail(-1, "invalid argument: metal without matrix"); return -1; }
        int n_model_pad = 0;
        for (auto* it : e->items) for (auto* m : it->metals) if (m == e->metals[index]) n_model_pad = it->dev.n_model_pad;
        src = e->xim.p + e->metals[index]->dev.xim_off; count = (int64_t)B * n_model_pad;
    }
    else if (what == 4) {
        // [0] the number of leading wavenumbers with a live P(k,mu) block in the last evaluation (the rest are exact zeros),
        // [1] the wavenumber up to which the mu sums take the node rule (0: plain loop), [2] nodes per wavenumber of that rule
        if (capacity < 3) { fail(-1, "invalid argument: capacity too small"); return -1; }
        int32_t live[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpy(live, e->k_live.p, 8 * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) { fail(-2, "hipMemcpy"); return -2; }
        out[0] = live[0]; out[1] = e->dev.k_node_max; out[2] = e->mu_lo + e->mu_hi + e->n_extra;
        if (capacity >= 4) out[3] = live[1];
        if (capacity >= 5) out[4] = e->last_tab_level;
        if (capacity >= 7) { out[5] = live[2]; out[6] = live[3]; }
        if (capacity >= 9) out[8] = e->last_form;               // 0: full chain, 1: the quadratic form Q', 2: its factored form
        if (capacity >= 8) { out[7] = live[4]; return capacity >= 9 ? 9 : 8; }       // walkers that left the mu rule's box since vmx_finalize
        return capacity >= 7 ? 7 : capacity >= 5 ? 5 : capacity >= 4 ? 4 : 3;
    }
    else { fail(-1, "invalid argument: what"); return -1; }
    if (count > capacity) { fail(-1, "invalid argument: capacity too small"); return -1; }
    if (hipMemcpy(out, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { fail(-2, "hipMemcpy"); return -2; }
    return count;
}

int vmx_matvec_device(vmx_engine* e, const double* d_A, int32_t rows, int32_t cols, const double* d_x, int32_t B,
                      double* d_y)
{
    REQUIRE(e && d_A && d_x && d_y, "vmx_matvec_device");
    wait_lane(e);
    REQUIRE(rows > 0 && cols > 0 && cols % VMX_PAD == 0, "cols (leading dimension) must be a multiple of 32, zero padded");
    REQUIRE(B > 0, "vmx_matvec_device: B > 0");
    HIP_OK(hipSetDevice(e->device));
    const int ldy = vmx_pad(rows);
    if (B <= 8) {
        launch_product(e, KC_MATVEC, d_A, cols, 0, rows, cols, d_x, cols, 0, B, d_y, ldy, 0, 1, 0);
    } else {
        const size_t need = (size_t)8 * B * ldy;
        if (e->mv_part.n < need) {
            HIP_OK(hipStreamSynchronize(e->stream));
            if (e->mv_part.alloc(need, false)) return -2;
        }
        const int ns = launch_product(e, KC_MATVEC, d_A, cols, 0, rows, cols, d_x, cols, 0, B, e->mv_part.p, ldy, 0, 1, 8 * B);
        const int64_t count = (int64_t)B * ldy;
        hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->cur, e->mv_part.p, d_y, count, ns);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int vmx_matmul_host(vmx_engine* e, const double* A, int32_t rows, int32_t cols, const double* X, int32_t B, double* Y)
{
    REQUIRE(e && A && X && Y && rows > 0 && cols > 0 && B > 0, "vmx_matmul_host");
    wait_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    const int ld = vmx_pad(cols), ldy = vmx_pad(rows);
    DevBuf<double> dA, dX, dY;
    if (upload_padded(dA, A, rows, cols, ld) || upload_padded(dX, X, B, cols, ld) || dY.alloc((size_t)B * ldy, true)) return -2;
    if (vmx_matvec_device(e, dA.p, rows, ld, dX.p, B, dY.p)) return -2;
    HIP_OK(hipStreamSynchronize(e->stream));
    HIP_OK(hipMemcpy2D(Y, (size_t)rows * sizeof(double), dY.p, (size_t)ldy * sizeof(double), (size_t)rows * sizeof(double), B,
                       hipMemcpyDeviceToHost));
    return 0;
}

int vmx_set_profiling(vmx_engine* e, int32_t enabled)
{
    REQUIRE(e, "vmx_set_profiling");
    wait_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->profiling) collect_spans(e);
    e->profiling = enabled != 0;
    e->prof_mask = 0xffffffffu;
    e->prof_stride = 1;
    // create the event pool up front so that no event is created between two launches of a timed run
    while (e->profiling && e->spans.size() < 96) {
        vmx_engine::Span s{};
        HIP_OK(hipEventCreate(&s.a));
        HIP_OK(hipEventCreate(&s.b));
        e->spans.push_back(s);
    }
    return 0;
}

int vmx_set_profiling_mask(vmx_engine* e, uint32_t kernel_class_mask)
{
    REQUIRE(e, "vmx_set_profiling_mask");
    wait_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    if (e->profiling) collect_spans(e);
    // bits 28 - 31: sampling stride minus one (0: every launch of the selected classes is timed; n: every (n + 1)-th - an event
    // pair costs the queue a few microseconds, ~4 % of a B = 256 step when it sits around one kernel of every step)
    e->prof_stride = kernel_class_mask == 0xffffffffu ? 1 : 1 + (int)(kernel_class_mask >> 28);
    e->prof_mask = kernel_class_mask == 0xffffffffu ? kernel_class_mask : (kernel_class_mask & 0x0fffffffu);
    for (auto& c : e->prof_count) c = 0;
    return 0;
}

int vmx_get_timings(vmx_engine* e, double* ms, int64_t* launches, int32_t reset)
{
    REQUIRE(e && ms && launches, "vmx_get_timings");
    wait_lane(e);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->stream));
    collect_spans(e);
    for (int i = 0; i < VMX_N_KERNELS; ++i) { ms[i] = e->ms[i]; launches[i] = e->launches[i]; }
    if (reset) for (int i = 0; i < VMX_N_KERNELS; ++i) { e->ms[i] = 0; e->launches[i] = 0; }
    return 0;
}

const char* vmx_kernel_name(int32_t kc) { return (kc >= 0 && kc < VMX_N_KERNELS) ? kKernelNames[kc] : ""; }

}  // extern "C"
