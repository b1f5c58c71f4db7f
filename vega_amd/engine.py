"""ctypes binding of libvegamx.so and the lowering of a :class:`vega_amd.setup.Problem` to it.

This is the reference-side binding INTEGRATION.md describes: plain ``ctypes`` against the C ABI of
``include/vegamx.h``.  There is no CPU fallback: if the library (or a GPU) is missing the engine
fails loudly.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import fftlog_op, static_terms

_LIB_PATH = Path(__file__).resolve().parent / 'libvegamx.so'
_lib = None

VMX_MAX_ELL = 4
VMX_MAX_SMOOTH = 3
VMX_N_KERNELS = 13
HCD = {'none': 0, 'Rogers': 1, 'sinc': 2, 'fvoigt': 3}
NL = {'none': 0, 'arinyo': 1, 'mcdonald': 2}
VD = {None: 0, 'gauss': 1, 'lorentz': 2}
SCALE_UNIT, SCALE_AP_AT, SCALE_AISO_EPS, SCALE_PHI_ALPHA = 0, 1, 2, 3
PKLIN = {'peak': 0, 'smooth': 1, 'full': 2}
MAT_DISTORTION, MAT_INVCOV, MAT_METAL = 0, 1, 2
BB_POS = {('pre', 'mul'): 0, ('pre', 'add'): 1, ('post', 'mul'): 2, ('post', 'add'): 3}
BB_POLY, BB_SKY = 0, 1
DEFAULT_GROWTH_RATE = 0.970386   # reference vega/utils.py:60

STATUS_BOUNDS, STATUS_ARINYO, STATUS_NONFINITE, STATUS_NOT_CONSTANT = 1, 2, 4, 8


class EngineError(RuntimeError):
    pass


class Tracer(C.Structure):
    _fields_ = [('bias_slot', C.c_int32), ('bias_eta_slot', C.c_int32), ('beta_slot', C.c_int32),
                ('is_lya', C.c_int32), ('discrete', C.c_int32), ('vd_sigma_slot', C.c_int32),
                ('evol_kind', C.c_int32), ('alpha_slot', C.c_int32)]


class PipeDesc(C.Structure):
    _fields_ = [
        ('tracer', Tracer * 2), ('same_tracer', C.c_int32), ('growth_rate_slot', C.c_int32),
        ('growth_rate_default', C.c_double), ('fast_metals', C.c_int32), ('pk_lin_kind', C.c_int32),
        ('is_peak', C.c_int32),
        ('uvb', C.c_int32), ('heii', C.c_int32),
        ('bias_gamma_slot', C.c_int32), ('bias_prim_slot', C.c_int32), ('lambda_uv_slot', C.c_int32),
        ('bias_gamma_e_slot', C.c_int32), ('lambda_heii_slot', C.c_int32),
        ('hcd_model', C.c_int32), ('bias_hcd_slot', C.c_int32), ('beta_hcd_slot', C.c_int32),
        ('l0_hcd_slot', C.c_int32), ('l0_default', C.c_double),
        ('nl_model', C.c_int32), ('arinyo_slot', C.c_int32 * 6), ('arinyo_power', C.c_double),
        ('gk_table', C.c_int32), ('mock_los_slot', C.c_int32), ('mock_los_size', C.c_double),
        ('peak_nl', C.c_int32), ('sigma_nl_par_slot', C.c_int32), ('sigma_nl_per_slot', C.c_int32),
        ('n_smooth', C.c_int32), ('smooth_par_slot', C.c_int32 * VMX_MAX_SMOOTH),
        ('smooth_per_slot', C.c_int32 * VMX_MAX_SMOOTH), ('smooth_weight', C.c_double * VMX_MAX_SMOOTH),
        ('exp_par_slot', C.c_int32), ('exp_per_slot', C.c_int32),
        ('vd_kind', C.c_int32), ('damping_scale', C.c_double), ('damping_power', C.c_int32),
        ('n_ell', C.c_int32), ('scale_mode', C.c_int32), ('scale_slot', C.c_int32 * 2),
        ('drp_slot', C.c_int32), ('croom_slot', C.c_int32 * 2), ('radiation', C.c_int32),
        ('rad_slot', C.c_int32 * 4), ('uv_shotnoise', C.c_int32), ('uvsn_slot', C.c_int32 * 3),
        ('single_ell', C.c_int32), ('z_eff', C.c_double)]


class MetalDesc(C.Structure):
    _fields_ = [('pipeline', C.c_int32), ('tracer', Tracer * 2), ('same_tracer', C.c_int32),
                ('growth_rate_slot', C.c_int32), ('growth_rate_default', C.c_double),
                ('extra_bias_slot', C.c_int32), ('apply_bias', C.c_int32), ('multiplicity', C.c_double),
                ('amplitude_slot', C.c_int32), ('in_direct', C.c_int32)]


class ItemDesc(C.Structure):
    _fields_ = [('n_model', C.c_int32), ('n_dist', C.c_int32), ('pipe_peak', C.c_int32),
                ('pipe_smooth', C.c_int32), ('bao_amp_slot', C.c_int32)]


VMX_FIT_MAXN, VMX_FIT_MAX_STAGES = 32, 2


class FitStage(C.Structure):
    _fields_ = [('n', C.c_int32), ('col', C.c_int32 * VMX_FIT_MAXN), ('has_lo', C.c_int32 * VMX_FIT_MAXN),
                ('has_hi', C.c_int32 * VMX_FIT_MAXN), ('lo', C.c_double * VMX_FIT_MAXN), ('hi', C.c_double * VMX_FIT_MAXN),
                ('err', C.c_double * VMX_FIT_MAXN)]


class FitSpec(C.Structure):
    _fields_ = [('n_stages', C.c_int32), ('n_params', C.c_int32), ('iterate', C.c_int32), ('maxfcn', C.c_int32),
                ('up', C.c_double), ('tol', C.c_double), ('stage', FitStage * VMX_FIT_MAX_STAGES)]


class MockStream(C.Structure):
    _fields_ = [('n_mocks', C.c_int32), ('wave', C.c_int32), ('draws', C.POINTER(C.c_double)), ('stride', C.c_int64),
                ('n_drawn', C.POINTER(C.c_int32)), ('timeout_seconds', C.c_double)]


class FitOptions(C.Structure):
    _fields_ = [('const_hint', C.c_int32), ('chunk', C.c_int32), ('lanes', C.c_int32), ('reserved', C.c_int32),
                ('mocks', C.POINTER(MockStream))]


class FitResultArrays(C.Structure):
    _fields_ = [('x', C.POINTER(C.c_double)), ('ext', C.POINTER(C.c_double)), ('V', C.POINTER(C.c_double)),
                ('fval', C.POINTER(C.c_double)), ('edm', C.POINTER(C.c_double)), ('flags', C.POINTER(C.c_int32)),
                ('nfcn', C.POINTER(C.c_int64)), ('n_iter', C.POINTER(C.c_int32))]


class FitStats(C.Structure):
    _fields_ = [('rounds', C.c_int64), ('evaluations', C.c_int64), ('engine_calls', C.c_int64), ('fits_unfinished', C.c_int64),
                ('calls_by_batch', C.c_int64 * 8), ('evaluations_by_batch', C.c_int64 * 8),
                ('seconds', C.c_double), ('seconds_setup', C.c_double), ('seconds_rounds', C.c_double),
                ('seconds_host_waiting', C.c_double), ('gpu_idle_seconds_between_rounds', C.c_double),
                ('seconds_waiting_for_draws', C.c_double), ('seconds_enqueuing_waves', C.c_double),
                ('seconds_enqueuing_rounds', C.c_double), ('seconds_enqueuing_calls', C.c_double)]


FIT_BATCH_BINS = ('1', '2..4', '5..16', '17..64', '65..256', '257..1024', '1025..4096', '4097..')


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def load_library():
    """Load libvegamx.so (built in-tree by ``__graft_entry__.build()``); no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get('VEGAMX_LIBRARY', _LIB_PATH))
    if not path.is_file():
        raise EngineError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; '
                          'g.build()"`. The vegamx engine has no CPU fallback.')
    lib = C.CDLL(str(path))
    lib.vmx_last_error.restype = C.c_char_p
    lib.vmx_kernel_name.restype = C.c_char_p
    lib.vmx_kernel_name.argtypes = [C.c_int32]
    lib.vmx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.vmx_destroy.argtypes = [C.c_void_p]
    lib.vmx_destroy.restype = None
    dptr, iptr = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.vmx_set_template.argtypes = [C.c_void_p, C.c_int32, dptr, dptr, dptr, dptr, dptr, C.c_int32]
    lib.vmx_set_fftlog.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_double, C.c_double, C.c_int32]
    lib.vmx_add_gk_table.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.vmx_add_gk_table_mock.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.vmx_set_spline_extrapolation.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_set_fftlog_padding.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.vmx_set_fvoigt_table.argtypes = [C.c_void_p, dptr, dptr, C.c_int32]
    lib.vmx_add_pipeline.argtypes = [C.c_void_p, C.POINTER(PipeDesc), C.c_int32, dptr, dptr, dptr, dptr, dptr]
    lib.vmx_add_item.argtypes = [C.c_void_p, C.POINTER(ItemDesc)]
    lib.vmx_set_shotnoise_table.argtypes = [C.c_void_p, dptr, C.c_int32, C.c_double, C.c_double]
    lib.vmx_item_set_additive_template.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_int32, C.c_double]
    lib.vmx_pipeline_set_odd_terms.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_double, C.c_double,
                                               C.c_int32, C.c_int32, iptr]
    lib.vmx_pipeline_set_odd_operator.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_int32]
    lib.vmx_item_add_metal.argtypes = [C.c_void_p, C.c_int32, C.POINTER(MetalDesc)]
    lib.vmx_item_add_broadband.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, iptr, dptr,
                                           C.c_int32]
    lib.vmx_item_set_matrix.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dptr]
    lib.vmx_item_set_matrix_csr.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), iptr, dptr]
    lib.vmx_item_set_mask.argtypes = [C.c_void_p, C.c_int32, iptr, C.c_int32]
    lib.vmx_item_set_data.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32]
    lib.vmx_set_global_invcov.argtypes = [C.c_void_p, dptr, C.c_int32]
    lib.vmx_item_set_mock_pool.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_int32]
    lib.vmx_set_mock_index.argtypes = [C.c_void_p, iptr, C.c_int32]
    lib.vmx_item_set_mock_factor.argtypes = [C.c_void_p, C.c_int32, dptr, dptr, C.c_int32]
    lib.vmx_item_get_mock_pool.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_int32]
    lib.vmx_host_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64]
    lib.vmx_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.vmx_add_prior.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double]
    lib.vmx_finalize.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.vmx_model_size.argtypes = [C.c_void_p]
    lib.vmx_pipeline_column.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_eval.argtypes = [C.c_void_p, dptr, C.c_int32, dptr, dptr, iptr]
    lib.vmx_eval_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vmx_sync.argtypes = [C.c_void_p]
    lib.vmx_eval_device_mocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vmx_fit_migrad.argtypes = [C.c_void_p, C.POINTER(FitSpec), C.c_int32, dptr, iptr, C.POINTER(FitOptions),
                                   C.POINTER(FitResultArrays), C.POINTER(FitStats)]
    lib.vmx_set_constant_nl_hint.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_set_direct_pk.argtypes = [C.c_void_p, dptr, C.c_int32, C.c_int32]
    lib.vmx_set_linear_spectra.argtypes = [C.c_void_p, dptr, dptr, dptr, C.c_int32]
    lib.vmx_item_set_marg_matrix.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32, C.c_int32]
    lib.vmx_marg_coeff.argtypes = [C.c_void_p, C.c_int32, dptr, C.c_int32]
    lib.vmx_set_quadratic_form.argtypes = [C.c_void_p, dptr]
    lib.vmx_set_quadratic_form_kind.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_set_static_poly.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_set_mu_quadrature.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_get_mu_nodes.argtypes = [C.c_void_p, dptr, dptr, C.c_int32]
    lib.vmx_set_mu_rule_box.argtypes = [C.c_void_p, C.c_int32, iptr, dptr, dptr]
    lib.vmx_stream.argtypes = [C.c_void_p]
    lib.vmx_stream.restype = C.c_void_p
    lib.vmx_last_stream.argtypes = [C.c_void_p]
    lib.vmx_last_stream.restype = C.c_void_p
    lib.vmx_set_lanes.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dptr, C.c_int64]
    lib.vmx_debug_read.restype = C.c_int64
    lib.vmx_matvec_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p]
    lib.vmx_item_set_metal_static.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dptr, C.c_int32]
    lib.vmx_item_set_metal_basis.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dptr, C.c_int32]
    lib.vmx_item_set_metal_kron.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dptr, C.c_int32, dptr, C.c_int32]
    lib.vmx_set_metal_beta_override.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    lib.vmx_set_parameter_transform.argtypes = [C.c_void_p, dptr, dptr]
    lib.vmx_matmul_host.argtypes = [C.c_void_p, dptr, C.c_int32, C.c_int32, dptr, C.c_int32, dptr]
    lib.vmx_pipeline_set_tracer_evolution.argtypes = [C.c_void_p, C.c_int32, dptr, dptr, C.c_int32]
    lib.vmx_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    lib.vmx_set_profiling_mask.argtypes = [C.c_void_p, C.c_uint32]
    lib.vmx_get_timings.argtypes = [C.c_void_p, dptr, C.POINTER(C.c_int64), C.c_int32]
    lib.vmx_struct_size.argtypes = [C.c_int32]
    for which, struct in enumerate((Tracer, PipeDesc, MetalDesc, ItemDesc, FitSpec, FitOptions, FitResultArrays, FitStats)):
        if lib.vmx_struct_size(which) != C.sizeof(struct):
            raise EngineError(f'ABI mismatch: {struct.__name__} is {C.sizeof(struct)} bytes here, '
                              f'{lib.vmx_struct_size(which)} in libvegamx.so')
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    'vmx_last_error', 'vmx_struct_size', 'vmx_create', 'vmx_destroy', 'vmx_set_template', 'vmx_set_fftlog', 'vmx_set_fftlog_padding', 'vmx_set_spline_extrapolation', 'vmx_set_fvoigt_table', 'vmx_add_gk_table', 'vmx_add_gk_table_mock',
    'vmx_add_pipeline', 'vmx_pipeline_set_tracer_evolution', 'vmx_pipeline_set_odd_terms', 'vmx_pipeline_set_odd_operator', 'vmx_set_shotnoise_table',
    'vmx_item_set_additive_template', 'vmx_add_item', 'vmx_item_add_metal', 'vmx_item_set_metal_static', 'vmx_item_set_metal_basis', 'vmx_item_set_metal_kron', 'vmx_set_metal_beta_override', 'vmx_item_add_broadband', 'vmx_item_set_matrix', 'vmx_item_set_matrix_csr',
    'vmx_item_set_mask', 'vmx_item_set_data', 'vmx_item_set_mock_pool', 'vmx_set_mock_index', 'vmx_item_set_mock_factor', 'vmx_item_get_mock_pool', 'vmx_host_alloc', 'vmx_host_free', 'vmx_set_global_invcov', 'vmx_add_prior', 'vmx_finalize',
    'vmx_model_size', 'vmx_pipeline_column', 'vmx_eval', 'vmx_eval_device', 'vmx_eval_device_mocks', 'vmx_fit_migrad', 'vmx_sync', 'vmx_set_constant_nl_hint', 'vmx_set_direct_pk', 'vmx_set_linear_spectra', 'vmx_item_set_marg_matrix', 'vmx_marg_coeff', 'vmx_set_quadratic_form', 'vmx_set_quadratic_form_kind', 'vmx_set_static_poly', 'vmx_set_mu_quadrature', 'vmx_set_mu_rule_box', 'vmx_get_mu_nodes', 'vmx_set_parameter_transform', 'vmx_stream', 'vmx_last_stream', 'vmx_set_lanes', 'vmx_debug_read', 'vmx_matvec_device', 'vmx_matmul_host',
    'vmx_set_profiling', 'vmx_set_profiling_mask', 'vmx_get_timings', 'vmx_kernel_name']


# --------------------------------------------------------------------------------------
# lowering: Problem -> descriptors
# --------------------------------------------------------------------------------------
class Lowering:
    """Resolves every name-keyed lookup of the reference's per-call code into theta columns."""

    def __init__(self, problem, extra_names=()):
        self.prob = problem
        self.names = sorted(set(problem.params) | set(extra_names))
        self.slot = {n: i for i, n in enumerate(self.names)}
        self.theta0 = np.array([problem.params.get(n, 0.0) for n in self.names], dtype=np.float64)

    def s(self, name):
        return self.slot.get(name, -1) if name is not None else -1

    def need(self, name):
        if name not in self.slot:
            raise KeyError(f'parameter {name!r} is required by the configured model but missing from '
                           '[parameters]')
        return self.slot[name]

    # -- tracers (reference vega/utils.py:45-82; metals.py:289-293 for the beta override)
    def tracer(self, tr, pk_opts=None, xi_opts=None, beta_name=None):
        t = Tracer()
        t.bias_slot = self.s('bias_' + tr.name)
        t.bias_eta_slot = self.s('bias_eta_' + tr.name)
        t.beta_slot = self.s(beta_name or ('beta_' + tr.name))
        if (t.bias_slot >= 0) + (t.bias_eta_slot >= 0) + (t.beta_slot >= 0) < 2:
            raise KeyError('For each tracer, you need to specify two of these three: (bias, bias_eta, beta). '
                           f'Offending tracer: {tr.name}')
        if t.bias_slot >= 0 and t.beta_slot >= 0:
            t.bias_eta_slot = -1    # "If all three are given, we use bias and beta"
        t.is_lya = int(tr.name == 'LYA')
        t.discrete = int(tr.type == 'discrete')
        t.vd_sigma_slot = -1
        if pk_opts is not None and pk_opts.velocity_dispersion is not None and t.discrete:
            t.vd_sigma_slot = self.need(f'sigma_velo_disp_{pk_opts.velocity_dispersion}_{tr.name}')
        t.evol_kind = 0
        t.alpha_slot = -1
        if xi_opts is not None:
            if xi_opts.evol_model.get(tr.name, 'standard') == 'croom':
                if tr.name != 'QSO':
                    raise ValueError('the Croom bias evolution only applies to QSO')
                t.evol_kind = 1
            else:
                t.alpha_slot = self.need('alpha_' + tr.name)
        return t

    def scale(self, pipe, is_peak):
        """Static resolution of ScaleParameters.get_ap_at (reference scale_parameters.py:38-160)."""
        sc = self.prob.scale
        if pipe.metal_corr and not sc.metal_scaling:
            return SCALE_UNIT, (-1, -1)

        def bao():
            if sc.parametrisation == 'ap_at':
                return SCALE_AP_AT, (self.need('ap'), self.need('at'))
            if sc.parametrisation == 'aiso_epsilon':
                return SCALE_AISO_EPS, (self.need('aiso'), self.need('epsilon'))
            return SCALE_PHI_ALPHA, (self.need('phi'), self.need('alpha'))

        def fullshape():
            if sc.parametrisation != 'phi_alpha' and not sc.full_shape_alpha:
                raise ValueError('Only the "phi_alpha" parametrisation works with split full-shape. '
                                 'Set full-shape-alpha to True for other parametrisations.')
            if sc.parametrisation == 'ap_at':
                return SCALE_AP_AT, (self.need('ap_full'), self.need('at_full'))
            if sc.parametrisation == 'aiso_epsilon':
                return SCALE_AISO_EPS, (self.need('aiso_full'), self.need('epsilon_full'))
            phi = 'phi_full' if sc.full_shape else 'phi_smooth'
            if sc.full_shape_alpha:
                alpha = 'alpha_full'
            elif is_peak:
                alpha = 'alpha'
            elif sc.two_alpha_smooth:
                alpha = f'alpha_smooth_{pipe.corr_name}'
            else:
                alpha = 'alpha_smooth'
            return SCALE_PHI_ALPHA, (self.need(phi), self.need(alpha))

        if sc.full_shape:
            return fullshape()
        if is_peak:
            return bao()
        if sc.smooth_scaling:
            return fullshape()
        return SCALE_UNIT, (-1, -1)

    def pipeline(self, engine, pipe, component, fast_metals=False, beta_names=(None, None),
                 growth_rate_override=None):
        pk, xi = pipe.pk, pipe.xi
        if xi.single_multipole >= 0 and (xi.single_multipole % 2 or xi.single_multipole > xi.ell_max):
            raise ValueError(f'single_multipole = {xi.single_multipole} is not one of the even multipoles <= ell_max')
        if (xi.relativistic or xi.asymmetry) and self.prob.scale.two_alpha_smooth:
            raise NotImplementedError('odd multipoles with two-alpha-smooth: the reference looks up alpha_smooth_None for these terms '
                                      '(vega/correlation_func.py:514, vega/scale_parameters.py:155-156) - there is nothing to reproduce')
        if xi.ell_max not in (0, 2, 4, 6):
            raise NotImplementedError(f'ell_max = {xi.ell_max} is not supported (even, <= 6)')
        is_peak = component == 'peak'
        n1, n2 = pipe.tracer1.name, pipe.tracer2.name
        params = self.prob.params

        d = PipeDesc()
        d.tracer[0] = self.tracer(pipe.tracer1, pk, xi, beta_names[0])
        d.tracer[1] = self.tracer(pipe.tracer2, pk, xi, beta_names[1])
        d.same_tracer = int(n1 == n2)
        if growth_rate_override is not None:
            d.growth_rate_slot, d.growth_rate_default = -1, growth_rate_override
        else:
            d.growth_rate_slot, d.growth_rate_default = self.s('growth_rate'), DEFAULT_GROWTH_RATE
        d.fast_metals = int(fast_metals)
        d.pk_lin_kind = PKLIN[component]
        d.is_peak = int(is_peak)

        d.uvb, d.heii = int(pk.uvb), int(pk.heii)
        for f in ('bias_gamma_slot', 'bias_prim_slot', 'lambda_uv_slot', 'bias_gamma_e_slot', 'lambda_heii_slot'):
            setattr(d, f, -1)
        lya = 'LYA' in (n1, n2)
        if pk.uvb and lya:
            d.bias_gamma_slot, d.bias_prim_slot = self.need('bias_gamma'), self.need('bias_prim')
            d.lambda_uv_slot = self.need('lambda_uv')
        if pk.heii and lya:
            d.bias_gamma_e_slot, d.bias_prim_slot = self.need('bias_gamma_e'), self.need('bias_prim')
            d.lambda_heii_slot = self.need('lambda_HeII')
        if not lya:
            d.uvb = d.heii = 0

        d.hcd_model = HCD[pk.hcd_model or 'none'] if lya else 0
        d.bias_hcd_slot = d.beta_hcd_slot = d.l0_hcd_slot = -1
        d.l0_default = 1.0
        if d.hcd_model:
            # per-correlation names win (reference power_spectrum.py:281-288)
            name = f'bias_hcd_{pipe.corr_name}'
            d.bias_hcd_slot = self.slot[name] if name in self.slot else self.need('bias_hcd')
            name = f'beta_hcd_{pipe.corr_name}'
            d.beta_hcd_slot = self.slot[name] if name in self.slot else self.need('beta_hcd')
            d.l0_hcd_slot = (self.need('L0_hcd') if pk.hcd_model == 'Rogers' else
                             self.s('L0_sinc') if pk.hcd_model == 'sinc' else self.s('L0_fvoigt'))
            if pk.hcd_model == 'fvoigt':
                engine._set_fvoigt(pk.fvoigt_table)

        skip_nl = pk.skip_nl_in_peak and is_peak
        nl = 'none' if (pk.small_scale_nl is None or skip_nl) else pk.small_scale_nl
        d.arinyo_power = 0.0
        for i in range(6):
            d.arinyo_slot[i] = -1
        if nl == 'arinyo':
            two = 'LY' in n1 and 'LY' in n2
            one = 'LY' in n1 or 'LY' in n2
            d.arinyo_power = 1.0 if two else (0.5 if one else 0.0)
            if not one:
                nl = 'none'
            else:
                for i, key in enumerate(('q1', 'q2', 'kv', 'av', 'bv', 'kp')):
                    d.arinyo_slot[i] = self.s('dnl_arinyo_' + key) if key == 'q2' else self.need('dnl_arinyo_' + key)
        if nl == 'mcdonald' and not (n1 == 'LYA' and n2 == 'LYA'):
            raise ValueError('dnl_mcdonald only applies to LYA x LYA')
        d.nl_model = NL[nl]

        d.gk_table = -1
        d.mock_los_slot, d.mock_los_size = -1, 0.0
        bs_rp = bs_rt = mock_rp = mock_rt = 0.0
        if pk.use_gk:
            # frozen at the parameters of the first call (reference power_spectrum.py:139-141, :494-495)
            bs_rp = params.get(f'par binsize {pipe.dataset}', pk.bin_size_rp)
            bs_rt = params.get(f'per binsize {pipe.dataset}', pk.bin_size_rt)
        if pk.mock_bin_size is not None:
            # a second binning factor (reference power_spectrum.py:143-160); static unless it follows a parameter
            mock_rp = mock_rt = pk.mock_bin_size
            if pk.mock_los_smoothing in ('growth', 'amplitude'):
                # the binning kernel follows a parameter: static as long as that parameter is not sampled
                name = 'growth_rate' if pk.mock_los_smoothing == 'growth' else 'los_smooth_amp'
                sampled = set((self.prob.sample_params or {}).get('limits', {}))
                if name in sampled:
                    # ... and a factor of its own in the mu loop, per walker, when it is (the plain 1000-point loop: no
                    # tables, no node rule); the table keeps the transverse factor
                    d.mock_los_slot, d.mock_los_size = self.need(name), float(pk.mock_bin_size)
                    mock_rp = 0.0
                else:
                    mock_rp *= 1 + params[name]
            elif pk.mock_los_smoothing == 'only-los':
                mock_rt = 0.0
        if pk.use_gk or pk.mock_bin_size is not None:
            d.gk_table = engine._gk_table(float(bs_rp), float(bs_rt), float(mock_rp), float(mock_rt))

        d.peak_nl = int(is_peak)
        d.sigma_nl_par_slot, d.sigma_nl_per_slot = self.s('sigmaNL_par'), self.s('sigmaNL_per')
        if is_peak and d.sigma_nl_par_slot < 0 and d.sigma_nl_per_slot < 0:
            raise ValueError('No parameters for peak NL found. Add sigmaNL_par and/or sigmaNL_per.')
        if is_peak and (d.sigma_nl_par_slot < 0 or d.sigma_nl_per_slot < 0) and 'growth_rate' not in self.slot:
            raise ValueError('growth_rate is needed to derive the missing sigmaNL parameter')

        terms = []
        d.exp_par_slot = d.exp_per_slot = -1
        if pk.fullshape_smoothing is not None and not skip_nl:
            if pk.fullshape_smoothing == 'gauss':
                main1, main2 = n1 in ('LYA', 'QSO'), n2 in ('LYA', 'QSO')
                if 'par_sigma_smooth' in self.slot or 'per_sigma_smooth' in self.slot:
                    par, per = self.s('par_sigma_smooth'), self.s('per_sigma_smooth')
                    par, per = (per if par < 0 else par), (par if per < 0 else per)
                    terms.append((par, per, 1.0))
                elif ('par_sigma_smooth_metals' in self.slot and 'per_sigma_smooth_metals' in self.slot
                      and not (main1 and main2)):
                    terms.append((self.slot['par_sigma_smooth_metals'], self.slot['per_sigma_smooth_metals'], 1.0))
                else:
                    for n in (n1, n2):
                        terms.append((self.need(f'par_sigma_smooth_{n}'), self.need(f'per_sigma_smooth_{n}'), 0.5))
            else:
                terms.append((self.need('par_sigma_smooth'), self.need('per_sigma_smooth'), 0.5))
                d.exp_par_slot, d.exp_per_slot = self.need('par_exp_smooth'), self.need('per_exp_smooth')
        d.n_smooth = len(terms)
        for i in range(VMX_MAX_SMOOTH):
            par, per, w = terms[i] if i < len(terms) else (-1, -1, 0.0)
            d.smooth_par_slot[i], d.smooth_per_slot[i], d.smooth_weight[i] = par, per, w

        d.vd_kind = VD[pk.velocity_dispersion]
        if d.vd_kind and 'discrete' not in (pipe.tracer1.type, pipe.tracer2.type):
            raise ValueError('velocity dispersion needs a discrete tracer')
        d.damping_scale = pk.damping_scale if pk.damping_scale is not None else 0.0
        d.damping_power = pk.damping_power

        d.n_ell = xi.ell_max // 2 + 1
        d.single_ell = xi.single_multipole // 2 if xi.single_multipole >= 0 else -1
        mode, slots = self.scale(pipe, is_peak)
        d.scale_mode = mode
        d.scale_slot[0], d.scale_slot[1] = slots
        d.drp_slot = self.s(pipe.delta_rp_name)
        d.croom_slot[0], d.croom_slot[1] = self.s('croom_par0'), self.s('croom_par1')
        # (2: `rescale-coords-systematics` - the term is evaluated on the rescaled coordinates, reference
        # correlation_func.py:470-475, :681-684)
        d.radiation = int(xi.radiation) * (2 if xi.rescale_coords_systematics else 1)
        for i, key in enumerate(('strength', 'asymmetry', 'lifetime', 'decrease')):
            d.rad_slot[i] = self.need('qso_rad_' + key) if xi.radiation else -1
        d.uv_shotnoise = int(xi.uv_shotnoise) * (2 if xi.rescale_coords_systematics else 1)
        for i in range(3):
            d.uvsn_slot[i] = -1
        if xi.uv_shotnoise:
            gamma = 'bias_gamma' if 'bias_gamma' in self.slot else 'bias_gamma_e'
            d.uvsn_slot[0], d.uvsn_slot[1] = self.need('uv_shotnoise_amp'), self.need('lambda_uv')
            d.uvsn_slot[2] = self.need(gamma)
            engine._set_shotnoise_table()
        d.z_eff = self.prob.z_eff
        return d


class Engine:
    """One vegamx engine handle on one GPU, built from a Problem."""

    def __init__(self, problem, max_batch=256, device=0, extra_names=(), metal_plan=None, kron_metals=True,
                 csr_threshold=None, static_poly=True, global_chi2_matrix=None):
        self.lib = load_library()
        self.prob = problem
        # engine_group: this engine's diagonal block of the global inverse covariance of a problem split over several engines
        self.global_chi2_matrix = None if global_chi2_matrix is None else _f64(global_chi2_matrix)
        # False: polynomial pipelines keep their per-walker P(k) -> xi path instead of the static spline-coefficient basis of
        # the template's spectra (include/vegamx.h: vmx_set_static_poly) - what direct_pk needs when metal terms are part of it
        self.static_poly = bool(static_poly)
        # fast_metals (see fast_metal_plan): per item, per metal pair ('pipeline', None) | ('share', leader index)
        # | ('static', xi vector) | ('basis', [3, n_model] Kaiser basis)
        self.metal_plan = metal_plan or {}
        self.kron_metals = bool(kron_metals)   # False: Kronecker-form metal matrices are uploaded dense (diagnosis)
        # a scipy.sparse distortion matrix stays in CSR form on the device when its density is below this fraction
        # (12 bytes per non-zero against 8 per entry, and the dense MFMA product wins for large batches well before that)
        self.csr_threshold = float(os.environ.get('VEGAMX_CSR_THRESHOLD', 0.3)) if csr_threshold is None else float(csr_threshold)
        self.csr_items = []
        self.metal_source = {}          # (item name, pair index) -> (global metal index, pipeline id, has matrix)
        self.low = Lowering(problem, extra_names)
        self.names = self.low.names
        self.n_params = len(self.names)
        self.max_batch = int(max_batch)
        self._small_io = None
        self.lanes = 1
        self._h = C.c_void_p()
        self._gk = {}
        self.device = int(device)
        self._check(self.lib.vmx_create(C.byref(self._h), int(device)))
        try:
            self._build()
        except Exception:
            self.close()
            raise

    # ---- helpers
    def _check(self, rc):
        if rc < 0:
            raise EngineError(self.lib.vmx_last_error().decode())
        return rc

    def _gk_table(self, bs_rp, bs_rt, mock_rp=0.0, mock_rt=0.0):
        key = (bs_rp, bs_rt, mock_rp, mock_rt)
        if key not in self._gk:
            self._gk[key] = self._check(self.lib.vmx_add_gk_table_mock(self._h, bs_rp, bs_rt, mock_rp, mock_rt))
        return self._gk[key]

    def _set_shotnoise_table(self):
        if getattr(self, '_shotnoise_set', False):
            return
        tau, a = static_terms.shotnoise_a()
        a = _f64(a)
        self._check(self.lib.vmx_set_shotnoise_table(self._h, _dp(a), a.size, float(tau[0]), float(tau[1] - tau[0])))
        self._shotnoise_set = True

    def _set_fvoigt(self, table):
        key = hash(np.ascontiguousarray(table).tobytes())
        if getattr(self, '_fvoigt_key', None) == key:
            return
        if getattr(self, '_fvoigt_key', None) is not None:
            raise NotImplementedError('all correlations must share one fvoigt table')
        x, f = _f64(table[:, 0]), _f64(table[:, 1])
        self._check(self.lib.vmx_set_fvoigt_table(self._h, _dp(x), _dp(f), x.size))
        self._fvoigt_key = key

    def close(self):
        if self._h:
            self.lib.vmx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- construction
    def _add_pipeline(self, desc, pipe, pk_lin=None):
        n = pipe.r.size
        pid = self._check(self.lib.vmx_add_pipeline(
            self._h, C.byref(desc), n, _dp(_f64(pipe.r)), _dp(_f64(pipe.mu)), _dp(_f64(pipe.z)),
            _dp(_f64(pipe.rel_z_evol)), _dp(_f64(pipe.xi_growth))))
        self.n_pipelines = pid + 1
        if getattr(pipe, 'rel_z_evol_1', None) is not None:
            # new-bias-evolution: each tracer of a cross-correlation evolves with its own redshift
            self._check(self.lib.vmx_pipeline_set_tracer_evolution(
                self._h, pid, _dp(_f64(pipe.rel_z_evol_1)), _dp(_f64(pipe.rel_z_evol_2)), n))
        if pipe.xi.relativistic or pipe.xi.asymmetry:
            # static splines of the odd-multipole terms of this component's linear spectrum
            low, k = self.low, self.prob.k
            rows = [fftlog_op.hamilton_spline(k, pk_lin, ell, kind)
                    for kind, ell in (('rel', 1), ('rel', 3), ('asy', 0), ('asy', 2))]
            coef = _f64(np.stack([r[0] for r in rows]))
            slots = np.array([low.need(n_) if use else -1 for n_, use in (
                ('Arel1', pipe.xi.relativistic), ('Arel3', pipe.xi.relativistic), ('Aasy0', pipe.xi.asymmetry),
                ('Aasy2', pipe.xi.asymmetry), ('Aasy3', pipe.xi.asymmetry))], dtype=np.int32)
            self._check(self.lib.vmx_pipeline_set_odd_terms(self._h, pid, _dp(coef), coef.shape[1], rows[0][1],
                                                            rows[0][2], int(pipe.xi.relativistic),
                                                            int(pipe.xi.asymmetry), _ip(slots)))
            # ... and as operators of the spectrum, for `direct_pk` (the caller's spectrum is the terms' pk_lin then)
            ops = _f64(np.stack([fftlog_op.hamilton_spline_operator(k, ell, kind)
                                 for kind, ell in (('rel', 1), ('rel', 3), ('asy', 0), ('asy', 2))]))
            self._check(self.lib.vmx_pipeline_set_odd_operator(self._h, pid, _dp(ops), ops.shape[1], ops.shape[2]))
        return pid

    def _build(self):
        prob, low, lib = self.prob, self.low, self.lib
        k = _f64(prob.k)
        pk_peak = _f64(prob.pk_full - prob.pk_smooth)       # reference model.py:177
        delta2 = _f64(k**3 * prob.pk_fid / (2 * np.pi**2))  # reference power_spectrum.py:462
        n_mu = {it.core.pk.n_mu for it in prob.items.values()}
        for it in prob.items.values():
            n_mu |= {m.pipeline.pk.n_mu for m in it.metals}
        if len(n_mu) != 1:
            raise NotImplementedError('one engine holds one mu grid: correlations with different num_bins_muk go through engine_group.make_engine')
        pipes = [p for it in prob.items.values() for p in [it.core] + [m.pipeline for m in it.metals]]
        old = {p.xi.old_fftlog for p in pipes}
        if len(old) != 1:
            raise NotImplementedError('one engine holds one transform: correlations with different old_fftlog go through engine_group.make_engine')
        self.old_fftlog = old.pop()
        lowring = {p.xi.fht_lowring for p in pipes}
        extrap = {bool(getattr(p.xi, 'fht_extrap', False)) and not self.old_fftlog for p in pipes}
        if len(lowring) != 1 or len(extrap) != 1:
            raise NotImplementedError('one engine holds one operator set: correlations with different fht_lowring / fht_extrap go through engine_group.make_engine')
        lowring, self.fht_extrap = lowring.pop(), extrap.pop()
        self.fftlog_pads = (0, 0)
        if self.fht_extrap:
            # `fht_extrap = True` (reference pktoxi.py:41,141): the FFTLog's power-law pads ride behind the samples of every
            # P_ell row and the operators carry their columns (fftlog_op.xi_operator(extrap=True))
            n_pad = 2 ** int(np.ceil(np.log2(2 * k.size))) - k.size
            self.fftlog_pads = (n_pad // 2, n_pad - n_pad // 2)
            self._check(lib.vmx_set_fftlog_padding(self._h, *self.fftlog_pads))
        self._check(lib.vmx_set_template(self._h, k.size, _dp(k), _dp(pk_peak), _dp(_f64(prob.pk_smooth)),
                                         _dp(_f64(prob.pk_full)), _dp(delta2), n_mu.pop()))
        if self.old_fftlog:
            self._check(lib.vmx_set_spline_extrapolation(self._h, 1))
        for i, ell in enumerate((0, 2, 4, 6)):
            op, x0, h, n_knots = fftlog_op.hamilton_xi_operator(k, ell) if self.old_fftlog else \
                fftlog_op.xi_operator(k, ell, lowring=lowring, extrap=self.fht_extrap)
            op = _f64(op)
            self._check(lib.vmx_set_fftlog(self._h, i, _dp(op), op.shape[0], x0, h, n_knots))

        self.item_names = list(prob.items)
        self.model_slices = {}
        n_metals_total = 0
        off = 0
        self.pipe_index = {}
        for qi, (name, item) in enumerate(prob.items.items()):
            peak = self._add_pipeline(low.pipeline(self, item.core, 'peak'), item.core, pk_peak)
            smooth = self._add_pipeline(low.pipeline(self, item.core, 'smooth'), item.core, prob.pk_smooth)
            self.pipe_index[(name, 'peak')] = peak
            self.pipe_index[(name, 'smooth')] = smooth
            idesc = ItemDesc(item.model_grid.size, item.dist_grid.size, peak, smooth, low.need('bao_amp'))
            iid = self._check(lib.vmx_add_item(self._h, C.byref(idesc)))
            assert iid == qi

            if item.metals:
                opts = item.metal_opts
                main = (item.tracer1.name, item.tracer2.name)
                override = prob.growth_rate if (opts['fast_metals'] and 'growth_rate' in low.slot
                                                and prob.growth_rate is not None) else None
                beta_subst = {}
                plan = self.metal_plan.get(name)
                # `no-metal-decomp = False` (reference model.py:120-123, :181-186): the metal terms are computed per
                # component - on the smooth spectrum, and on the peak spectrum (with the peak's non-linear broadening)
                # times bao_amp - instead of once on the full spectrum: every pair enters twice
                if opts['no_metal_decomp']:
                    components = [('full', prob.pk_full, -1)]
                else:
                    if opts['fast_metals'] or any(kind != 'pipeline' for kind, _ in plan or ()):
                        raise NotImplementedError('no-metal-decomp = False with fast_metals is not accelerated (the '
                                                  "reference's metal caches then ignore the component)")
                    components = [('smooth', prob.pk_smooth, -1), ('peak', pk_peak, low.need('bao_amp'))]
                pair_pid = {}
                entry = 0
                for mi, pair in enumerate(item.metals):
                    n1, n2 = pair.names
                    if opts['single_metal_beta']:
                        # the substitution accumulates over the loop (reference metals.py:289-293)
                        for n in (n1, n2):
                            if n not in main:
                                low.need('beta_metals')
                                beta_subst[n] = 'beta_metals'
                    betas = (beta_subst.get(n1), beta_subst.get(n2))
                    fast = bool(opts['fast_metal_bias'])
                    kind, arg = plan[mi] if plan else ('pipeline', None)
                    for component, pk_lin, amplitude_slot in components:
                        if kind == 'pipeline':
                            pid = self._add_pipeline(
                                low.pipeline(self, pair.pipeline, component, fast_metals=fast, beta_names=betas,
                                             growth_rate_override=override), pair.pipeline, pk_lin)
                            if component != 'peak':
                                self.pipe_index[(name, pair.names)] = pid
                                pair_pid[mi] = pid
                        elif kind == 'share':
                            pid = pair_pid[arg]     # the reference's per-call cache hands this pair the leader's xi
                        else:
                            pid = -1
                        md = MetalDesc()
                        md.pipeline = pid
                        md.amplitude_slot = amplitude_slot
                        # direct_pk (reference model.py:188-207, :120-123): the metal terms are part of the direct model only with
                        # `no-metal-decomp = False`, computed on the caller's spectrum - the smooth-spectrum entries here
                        md.in_direct = int(component == 'smooth')
                        md.tracer[0] = low.tracer(pair.pipeline.tracer1, beta_name=betas[0])
                        md.tracer[1] = low.tracer(pair.pipeline.tracer2, beta_name=betas[1])
                        md.same_tracer = int(n1 == n2)
                        if override is not None:
                            md.growth_rate_slot, md.growth_rate_default = -1, override
                        else:
                            md.growth_rate_slot, md.growth_rate_default = low.s('growth_rate'), DEFAULT_GROWTH_RATE
                        md.extra_bias_slot = -1
                        if (not pair.cross_with_main) and opts['separate_metal_auto_biases'] and n1 != n2:
                            found = [c for c in pair.auto_bias_names if c in low.slot]
                            if not found:
                                raise ValueError(f'Separate metal auto biases is on, but no {pair.auto_bias_names[0]} '
                                                 f'or {pair.auto_bias_names[1]} parameter found for {pair.names}.')
                            md.extra_bias_slot = low.slot[found[0]]
                        md.apply_bias = int(fast)
                        md.multiplicity = 2.0 if pair.double_count else 1.0
                        self._check(lib.vmx_item_add_metal(self._h, iid, C.byref(md)))
                        if component != 'peak':
                            self.metal_source[(name, mi)] = (n_metals_total, pid, pair.matrix is not None)
                        n_metals_total += 1
                        if kind == 'static':
                            vec = _f64(arg)
                            self._check(lib.vmx_item_set_metal_static(self._h, iid, entry, _dp(vec), vec.size))
                        elif kind == 'basis':
                            basis = _f64(arg)
                            assert basis.shape == (3, item.model_grid.size)
                            self._check(lib.vmx_item_set_metal_basis(self._h, iid, entry, _dp(basis), basis.shape[1]))
                        elif getattr(pair, 'kron', None) is not None and self.kron_metals:
                            # new_metals matrices are Kronecker products: two small products per walker instead of one
                            # [n_model]^2 product
                            a_rp = _f64(pair.kron[0])
                            b_rt = None if pair.kron[1] is None else _f64(pair.kron[1])
                            n_rt = item.model_grid.size // a_rp.shape[0]
                            self._check(lib.vmx_item_set_metal_kron(self._h, iid, entry, _dp(a_rp), a_rp.shape[0],
                                                                    None if b_rt is None else _dp(b_rt), n_rt))
                        elif pair.matrix is not None:
                            dense = _f64(pair.matrix.toarray() if hasattr(pair.matrix, 'toarray') else pair.matrix)
                            self._check(lib.vmx_item_set_matrix(self._h, iid, MAT_METAL, entry, dense.shape[0],
                                                                dense.shape[1], _dp(dense)))
                        entry += 1

            if item.inst_sys_table is not None:
                vec = _f64(static_terms.instrumental_systematics_template(item))
                self._check(lib.vmx_item_set_additive_template(self._h, iid, _dp(vec), vec.size,
                                                               low.s('desi_inst_sys_amp'),
                                                               static_terms.DESI_INST_SYS_DEFAULT_AMP))
            for term in item.broadband:
                grid = item.model_grid if term.pos == 'pre' else item.dist_grid
                pos = BB_POS[(term.pos, term.kind)]
                if term.func == 'broadband_sky':
                    slots = np.array([low.need(term.name + '-scale-sky'), low.need(term.name + '-sigma-sky')],
                                     dtype=np.int32)
                    window = ((grid.rp >= 0.) & (grid.rp < grid.rp_binsize)).astype(np.float64)
                    basis = _f64(np.stack([grid.rt, window]))
                    func = BB_SKY
                else:
                    if term.coords == 'r,mu':
                        r1, r2 = grid.r / 100., grid.mu
                    else:
                        r1 = grid.r / 100. * grid.mu
                        r2 = grid.r / 100. * np.sqrt(1 - grid.mu**2)
                    p1 = np.arange(term.r1[0], term.r1[1] + 1, term.r1[2])
                    p2 = np.arange(term.r2[0], term.r2[1] + 1, term.r2[2])
                    slots, rows = [], []
                    for i in p1:
                        for j in p2:
                            slots.append(low.need(f'{term.name} ({i},{j})'))
                            rows.append(r1**i * r2**j)
                    slots = np.array(slots, dtype=np.int32)
                    basis = _f64(np.stack(rows))
                    func = BB_POLY
                if slots.size > 16:
                    raise NotImplementedError('more than 16 coefficients in one broadband term')
                self._check(lib.vmx_item_add_broadband(self._h, iid, pos, func, slots.size, _ip(slots),
                                                       _dp(basis), basis.shape[1]))

            if item.distortion is not None:
                dm = item.distortion
                if hasattr(dm, 'tocsr') and dm.nnz < self.csr_threshold * dm.shape[0] * dm.shape[1]:
                    # the reference's own representation (scipy csr_array, vega/data.py:342-346), kept as is
                    csr = dm.tocsr().copy()
                    csr.sum_duplicates()        # canonical form: sorted, one entry per (row, column)
                    csr.sort_indices()
                    ptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
                    idx = np.ascontiguousarray(csr.indices, dtype=np.int32)
                    val = _f64(csr.data)
                    self._check(lib.vmx_item_set_matrix_csr(self._h, iid, csr.shape[0], csr.shape[1],
                                                            ptr.ctypes.data_as(C.POINTER(C.c_int64)), _ip(idx), _dp(val)))
                    self.csr_items.append(name)
                else:
                    dense = _f64(dm.toarray() if hasattr(dm, 'toarray') else dm)
                    self._check(lib.vmx_item_set_matrix(self._h, iid, MAT_DISTORTION, 0, dense.shape[0],
                                                        dense.shape[1], _dp(dense)))
            idx = np.flatnonzero(item.model_mask).astype(np.int32)
            if idx.size != item.data_size:
                raise ValueError(f'{name}: model mask keeps {idx.size} bins but the data mask {item.data_size}')
            self._check(lib.vmx_item_set_mask(self._h, iid, _ip(idx), idx.size))
            self._check(lib.vmx_item_set_data(self._h, iid, _dp(_f64(item.masked_data_vec)), idx.size))
            if getattr(item, 'marg_diff2coeff', None) is not None:
                # best-fit template coefficients = M . residual (reference vega_interface.py:546-579)
                m = _f64(item.marg_diff2coeff)
                self._check(lib.vmx_item_set_marg_matrix(self._h, iid, _dp(m), m.shape[0], m.shape[1]))
            if item.cov is not None and prob.global_cov is None and self.global_chi2_matrix is None:
                cinv = _f64(item.chi2_matrix)      # C^-1, or P^T C^-1 P with marginalize-in-fit (setup.py)
                self._check(lib.vmx_item_set_matrix(self._h, iid, MAT_INVCOV, 0, cinv.shape[0], cinv.shape[1],
                                                    _dp(cinv)))
            self.model_slices[name] = slice(off, off + item.dist_grid.size)
            off += item.dist_grid.size

        if prob.global_cov is not None or self.global_chi2_matrix is not None:
            g = self.global_chi2_matrix if self.global_chi2_matrix is not None else _f64(prob.global_masks()['chi2_matrix'])
            self._check(lib.vmx_set_global_invcov(self._h, _dp(g), g.shape[0]))
        for pname, (mean, sigma) in prob.priors.items():
            self._check(lib.vmx_add_prior(self._h, low.need(pname), float(mean), float(sigma)))
        # applicability guard of the mu node rule: the box it is validated on (mu_quadrature.rule_box)
        from .mu_quadrature import rule_box
        box = rule_box(low.names)
        if box:
            slots = np.array([low.slot[n] for n in box], dtype=np.int32)
            lo = _f64([box[n][0] for n in box])
            hi = _f64([box[n][1] for n in box])
            self._check(lib.vmx_set_mu_rule_box(self._h, slots.size, _ip(slots), _dp(lo), _dp(hi)))
        self.mu_rule_box = box
        if not self.static_poly:
            self._check(lib.vmx_set_static_poly(self._h, 0))
        self._check(lib.vmx_finalize(self._h, self.n_params, self.max_batch))
        self.model_size = self._check(lib.vmx_model_size(self._h))
        # chi2-only evaluations run as a static quadratic form around the configured parameter values when the
        # configuration allows it (include/vegamx.h: vmx_set_quadratic_form)
        self.quadratic_form = bool(self._check(lib.vmx_set_quadratic_form(self._h, _dp(_f64(low.theta0)))))

    # ---- evaluation
    def theta_from_params(self, params=None):
        theta = self.low.theta0.copy()
        if params:
            for name, value in params.items():
                if name == 'peak':
                    continue
                if name not in self.low.slot:
                    raise KeyError(f'unknown parameter {name!r}: rebuild the engine with extra_names')
                theta[self.low.slot[name]] = value
        return theta

    def eval(self, theta, want_model=False):
        """theta [B, n_params] -> (chi2 [B], status [B], model [B, model_size] or None)."""
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            theta = theta[None, :]
        B = theta.shape[0]
        if theta.shape[1] != self.n_params:
            raise ValueError(f'theta must have {self.n_params} columns')
        if not want_model and B <= 8:
            # small chi2-only calls are latency-bound: persistent in / out buffers with their pointers made once (building
            # three ctypes pointers costs as much as two of the chain's kernels)
            io = self._small_io
            if io is None:
                t_in, c_out, s_out = np.empty((8, self.n_params)), np.empty(8), np.empty(8, dtype=np.int32)
                io = self._small_io = (t_in, c_out, s_out, _dp(t_in), _dp(c_out), _ip(s_out))
            io[0][:B] = theta
            if self.lib.vmx_eval(self._h, io[3], B, io[4], None, io[5]) < 0:
                raise EngineError(self.lib.vmx_last_error().decode())
            return io[1][:B].copy(), io[2][:B].copy(), None
        theta = _f64(theta)
        chi2 = np.empty(B)
        status = np.empty(B, dtype=np.int32)
        model = np.empty((B, self.model_size)) if want_model else None
        self._check(self.lib.vmx_eval(self._h, _dp(theta), B, _dp(chi2), _dp(model) if want_model else None,
                                      _ip(status)))
        return chi2, status, model

    def eval_device(self, d_theta_ptr, B, d_chi2_ptr, d_model_ptr=None, d_status_ptr=None):
        self._check(self.lib.vmx_eval_device(self._h, d_theta_ptr, B, d_chi2_ptr, d_model_ptr, d_status_ptr))

    def sync(self):
        self._check(self.lib.vmx_sync(self._h))

    def eval_device_mocks(self, d_theta_ptr, B, d_chi2_ptr, d_mock_ptr, d_status_ptr=None):
        """``eval_device`` for walkers that bring their own mock rows: ``d_mock_ptr`` int32 [B] in device memory, per walker the
        pool row it is compared with (include/vegamx.h: vmx_eval_device_mocks)."""
        self._check(self.lib.vmx_eval_device_mocks(self._h, d_theta_ptr, B, d_chi2_ptr, d_status_ptr, d_mock_ptr))

    def set_mock_factor(self, name, chol, fiducial_masked):
        """Cholesky factor of item ``name``'s (scaled) masked covariance and its fiducial model on the masked bins: the mocks of a
        ``fit_migrad(..., mock_stream=...)`` run are made from them on the device (include/vegamx.h: vmx_item_set_mock_factor)."""
        qi = self.item_names.index(name)
        chol, fid = _f64(chol), _f64(fiducial_masked)
        self._check(self.lib.vmx_item_set_mock_factor(self._h, qi, _dp(chol), _dp(fid), fid.size))

    def pinned_empty(self, shape):
        """A float64 array of ``shape`` in page-locked host memory (include/vegamx.h: vmx_host_alloc) - for buffers the library
        copies from asynchronously.  Kept per engine and reused while it is large enough; the memory lives until `close()`, so the
        array must not be used after that."""
        count = int(np.prod(shape))
        kept = self.__dict__.get('_pinned')
        if kept is None or kept[1] < count:
            if kept is not None:
                self._check(self.lib.vmx_host_free(self._h, kept[0]))
                self._pinned = None
            ptr = C.c_void_p()
            self._check(self.lib.vmx_host_alloc(self._h, C.byref(ptr), count * 8))
            self._pinned = kept = (ptr, count)
        flat = np.ctypeslib.as_array(C.cast(kept[0], C.POINTER(C.c_double)), shape=(kept[1],))
        return flat[:count].reshape(shape)

    def get_mock_pool(self, name, n_mocks):
        """The first ``n_mocks`` rows of item ``name``'s mock pool, as the device holds them."""
        qi = self.item_names.index(name)
        n = self.prob.items[name].data_size
        out = np.empty((n_mocks, n))
        self._check(self.lib.vmx_item_get_mock_pool(self._h, qi, _dp(out), n_mocks, n))
        return out

    def fit_migrad(self, plan, theta0, mock_rows=None, const_hint=-1, chunk=0, lanes=0, mock_stream=None):
        """MIGRAD fits of ``theta0.shape[0]`` parameter rows on the device (include/vegamx.h: vmx_fit_migrad).  ``plan``:
        :meth:`vega_amd.migrad.MigradMinimizer.plan` with ``free`` = indices into the engine's parameter columns; ``mock_rows``:
        the pool row every fit is fitted to (None: the items' data).  Returns (per-stage result dicts, statistics of the run)."""
        stages = plan['stages']
        if not 1 <= len(stages) <= VMX_FIT_MAX_STAGES:
            raise ValueError(f'1 .. {VMX_FIT_MAX_STAGES} Minuit objects per fit')
        theta0 = _f64(theta0)
        F = theta0.shape[0]
        if theta0.ndim != 2 or theta0.shape[1] != self.n_params:
            raise ValueError(f'theta0 must have {self.n_params} columns')
        spec = FitSpec()
        spec.n_stages, spec.n_params = len(stages), self.n_params
        spec.iterate, spec.maxfcn, spec.up, spec.tol = int(plan['iterate']), int(plan['maxfcn']), float(plan['up']), float(plan['tol'])
        for k, st in enumerate(stages):
            free = np.asarray(st['free'], dtype=int)
            if not 1 <= free.size <= VMX_FIT_MAXN:
                raise NotImplementedError(f'{free.size} free parameters: the device-resident fits take 1 .. {VMX_FIT_MAXN} per Minuit object')
            cs = spec.stage[k]
            cs.n = free.size
            for i, (j, lim, err) in enumerate(zip(free, st['limits'], st['errors'])):
                lo, hi = (None, None) if lim is None else lim
                lo = None if lo is None or not np.isfinite(lo) else float(lo)
                hi = None if hi is None or not np.isfinite(hi) else float(hi)
                cs.col[i], cs.has_lo[i], cs.has_hi[i] = int(j), int(lo is not None), int(hi is not None)
                cs.lo[i], cs.hi[i], cs.err[i] = lo or 0., hi or 0., float(err)
        outs, res = [], (FitResultArrays * len(stages))()
        for k, st in enumerate(stages):
            n = len(st['free'])
            o = dict(x=np.zeros((F, n)), ext=np.zeros((F, n)), V=np.zeros((F, n, n)), fval=np.zeros(F), edm=np.zeros(F),
                     flags=np.zeros(F, dtype=np.int32), nfcn=np.zeros(F, dtype=np.int64), n_iter=np.zeros(F, dtype=np.int32))
            outs.append(o)
            r = res[k]
            r.x, r.ext, r.V, r.fval, r.edm = _dp(o['x']), _dp(o['ext']), _dp(o['V']), _dp(o['fval']), _dp(o['edm'])
            r.flags, r.nfcn, r.n_iter = _ip(o['flags']), o['nfcn'].ctypes.data_as(C.POINTER(C.c_int64)), _ip(o['n_iter'])
        rows = None if mock_rows is None else np.ascontiguousarray(mock_rows, dtype=np.int32)
        if rows is not None and rows.shape != (F,):
            raise ValueError('mock_rows: one pool row per fit')
        opt = FitOptions(int(const_hint), int(chunk), int(lanes), 0, None)
        if mock_stream is not None:
            # the mocks are made while the fits run: `draws` [F, stride] float64 filled by a producer thread, `counter` int32 [1]
            # counting the complete rows (include/vegamx.h: vmx_mock_stream)
            draws, counter = mock_stream['draws'], mock_stream['counter']
            if draws.dtype != np.float64 or not draws.flags.c_contiguous or draws.shape[0] != F or counter.dtype != np.int32:
                raise ValueError('mock_stream: draws float64 [n_fits, stride] C-contiguous, counter int32')
            ms = MockStream(F, int(mock_stream.get('wave', 0)), _dp(draws), draws.shape[1], _ip(counter),
                            float(mock_stream.get('timeout', 0.)))
            opt.mocks = C.pointer(ms)
        stats = FitStats()
        self._check(self.lib.vmx_fit_migrad(self._h, C.byref(spec), F, _dp(theta0), None if rows is None else _ip(rows),
                                            C.byref(opt), res, C.byref(stats)))
        info = {name: getattr(stats, name) for name, _ in FitStats._fields_ if not name.endswith('by_batch')}
        info['calls_by_batch'] = dict(zip(FIT_BATCH_BINS, list(stats.calls_by_batch)))
        info['evaluations_by_batch'] = dict(zip(FIT_BATCH_BINS, list(stats.evaluations_by_batch)))
        return outs, info

    def set_constant_nl_hint(self, on=True, gaussian=False):
        """For ``eval_device``: the caller asserts that the Arinyo parameters - with ``gaussian`` also the smoothing,
        peak-broadening and Gaussian velocity-dispersion parameters - are identical for all walkers of a batch
        (violations are flagged per walker, never silently wrong)."""
        self._check(self.lib.vmx_set_constant_nl_hint(self._h, 0 if not on else 2 if gaussian else 1))

    def stream_handle(self):
        """hipStream_t of the engine as an integer (for ``torch.cuda.ExternalStream``)."""
        return int(self.lib.vmx_stream(self._h))

    def last_stream_handle(self):
        """hipStream_t of the last ``eval_device`` (with two lanes: the lane it ran on)."""
        return int(self.lib.vmx_last_stream(self._h))

    def set_lanes(self, lanes):
        """1 or 2 batches in flight for chi2-only ``eval_device`` calls (include/vegamx.h: vmx_set_lanes); the second lane
        borrows every static tensor of the engine."""
        self._check(self.lib.vmx_set_lanes(self._h, int(lanes)))
        self.lanes = int(lanes)

    def set_data(self, name, masked_data):
        qi = self.item_names.index(name)
        d = _f64(masked_data)
        self._check(self.lib.vmx_item_set_data(self._h, qi, _dp(d), d.size))

    def set_mock_pool(self, name, pool):
        """Masked mock data vectors [n_mocks, n_masked] of item ``name`` (Monte-Carlo fits)."""
        qi = self.item_names.index(name)
        pool = _f64(np.atleast_2d(pool))
        self._check(self.lib.vmx_item_set_mock_pool(self._h, qi, _dp(pool), pool.shape[0], pool.shape[1]))

    def set_mock_index(self, index=None):
        """Per-walker pool row used as data in the following evaluations (None: the items' own data)."""
        if index is None:
            self._check(self.lib.vmx_set_mock_index(self._h, None, 0))
        else:
            idx = np.ascontiguousarray(index, dtype=np.int32)
            self._check(self.lib.vmx_set_mock_index(self._h, _ip(idx), idx.size))

    def set_invcov(self, name, invcov):
        qi = self.item_names.index(name)
        m = _f64(invcov)
        self._check(self.lib.vmx_item_set_matrix(self._h, qi, MAT_INVCOV, 0, m.shape[0], m.shape[1], _dp(m)))

    def debug_read(self, what, index=0, count=0):
        out = np.empty(count)
        n = self.lib.vmx_debug_read(self._h, what, index, _dp(out), out.size)
        if n < 0:
            raise EngineError(self.lib.vmx_last_error().decode())
        return out[:n]

    def pk_multipoles(self, B=1):
        """P_ell(k) of the last evaluation (batch size B): dict pipeline id -> [B, 4, nk] for the pipelines that form
        their multipoles per walker (the stage tap behind `model_pk`, reference model.py:106-107)."""
        nk = self.prob.k.size
        nkp = (nk + sum(self.fftlog_pads) + 31) // 32 * 32          # (a row: the samples, then the FFTLog's power-law pads)
        out = {}
        n_cols = None
        for pid in range(self.n_pipelines):
            col = self.lib.vmx_pipeline_column(self._h, pid)
            if -3 < col < 0:            # (-1 / -2: an error code; <= -3 encodes "no column" and the column count)
                raise EngineError(self.lib.vmx_last_error().decode())
            if col < -2:
                n_cols = -3 - col
                continue
            out[pid] = col
        if n_cols is None:
            n_cols = len(out)
        pl = self.debug_read(0, 0, 4 * B * n_cols * nkp).reshape(4, n_cols, B, nkp)      # pipeline-major columns
        return {pid: np.ascontiguousarray(pl[:, col, :, :nk].transpose(1, 0, 2)) for pid, col in out.items()}

    def metal_xi(self, item_name, pair_index):
        """Correlation of one metal pair in the last evaluation (walker 0): after its metal matrix, before the
        bias product and the multiplicity."""
        g, pid, has_matrix = self.metal_source[(item_name, pair_index)]
        n = self.prob.items[item_name].model_grid.size
        cap = self.max_batch * ((n + 31) // 32 * 32)
        if has_matrix:
            return self.debug_read(3, g, cap)[:n].copy()
        return self.debug_read(1, pid, cap)[:n].copy()

    def set_direct_pk(self, pk=None):
        """direct_pk mode: per-walker linear spectra [B, nk] (None: back to the fiducial template)."""
        if pk is None:
            self._check(self.lib.vmx_set_direct_pk(self._h, None, 0, 0))
            return
        pk = _f64(np.atleast_2d(pk))
        self._check(self.lib.vmx_set_direct_pk(self._h, _dp(pk), pk.shape[0], pk.shape[1]))

    QUADRATIC_FORM_KINDS = {'auto': 0, 'q': 1, 'factored': 2}

    def set_quadratic_form_kind(self, kind='auto'):
        """'auto' (the cheaper form per engine), 'q' (the half-form matrix Q', nq^2 flops per walker) or 'factored'
        (|| U r0 - F dx ||^2, 2 n_masked nq flops: model grids much finer than the data grid) - include/vegamx.h:
        vmx_set_quadratic_form_kind."""
        self._check(self.lib.vmx_set_quadratic_form_kind(self._h, self.QUADRATIC_FORM_KINDS[kind]))

    def last_form(self):
        """Form the last evaluation took: 'full' (distortion + C^-1 products), 'q' or 'factored'."""
        return ('full', 'q', 'factored')[int(self.debug_read(4, 0, 9)[8])]

    def set_quadratic_form(self, on=True):
        """Switch the quadratic form of chi2-only evaluations on (expansion point: the configured parameter values) or
        off (every evaluation runs the full chain).  Returns whether the form is in use."""
        ref = _f64(self.low.theta0)
        self.quadratic_form = bool(self._check(self.lib.vmx_set_quadratic_form(self._h, _dp(ref) if on else None))) and on
        return self.quadratic_form

    def mu_nodes(self):
        """(mu, w) of the extra nodes of the mu rule as the engine holds them."""
        n = self._check(self.lib.vmx_get_mu_nodes(self._h, None, None, 0))
        mu, w = np.empty(n), np.empty(n)
        self._check(self.lib.vmx_get_mu_nodes(self._h, _dp(mu), _dp(w), n))
        return mu, w

    def set_mu_quadrature(self, node_rule=True):
        """True: the node rule that reproduces the reference's 1000-point mu sums from 178 evaluations (default);
        False: the 1000-point loop itself.  Returns the setting in effect."""
        return bool(self._check(self.lib.vmx_set_mu_quadrature(self._h, int(bool(node_rule)))))

    def set_linear_spectra(self, pk_full, pk_smooth):
        """Replace the template's linear spectra (``Model.compute(pars, pk_full, pk_smooth)``, reference
        model.py:157-187); the peak spectrum is their difference (model.py:177)."""
        pk_full, pk_smooth = _f64(pk_full), _f64(pk_smooth)
        if pk_full.shape != (self.prob.k.size,) or pk_smooth.shape != pk_full.shape:
            raise ValueError('pk_full and pk_smooth live on the template k grid')
        peak = _f64(pk_full - pk_smooth)
        self._check(self.lib.vmx_set_linear_spectra(self._h, _dp(peak), _dp(pk_smooth), _dp(pk_full), pk_full.size))

    def marg_coeff(self, name, B):
        """Marginalisation-template coefficients [B, n_templates] of item ``name`` for the last evaluation (its batch
        size B): the static map of the residual, applied by the engine's product kernels."""
        qi = self.item_names.index(name)
        nt = self.prob.items[name].marg_diff2coeff.shape[0]
        out = np.empty((B, nt))
        self._check(self.lib.vmx_marg_coeff(self._h, qi, _dp(out), B))
        return out

    def set_parameter_transform(self, scale=None, shift=None):
        """Parameter-level blinding: every walker becomes scale * theta + shift (per column) before the model and the
        priors read it; None switches it off."""
        if scale is None:
            self._check(self.lib.vmx_set_parameter_transform(self._h, None, None))
            return
        scale, shift = _f64(scale), _f64(shift)
        if scale.shape != (self.n_params,) or shift.shape != (self.n_params,):
            raise ValueError('scale and shift hold one value per parameter column')
        self._check(self.lib.vmx_set_parameter_transform(self._h, _dp(scale), _dp(shift)))

    def set_metal_beta_override(self, beta=None):
        """Set-up hook: every tracer of a bias-free metal pipeline takes ``beta`` (None: back to the parameters)."""
        self._check(self.lib.vmx_set_metal_beta_override(self._h, int(beta is not None), 0.0 if beta is None else float(beta)))

    def matvec_device(self, d_A, rows, cols_padded, d_x, B, d_y):
        self._check(self.lib.vmx_matvec_device(self._h, d_A, rows, cols_padded, d_x, B, d_y))

    def matmul_host(self, A, X):
        """X @ A.T on the GPU for host arrays A [rows, cols], X [B, cols] (the product kernels of the chain)."""
        A, X = _f64(A), _f64(X)
        Y = np.empty((X.shape[0], A.shape[0]))
        self._check(self.lib.vmx_matmul_host(self._h, _dp(A), A.shape[0], A.shape[1], _dp(X), X.shape[0], _dp(Y)))
        return Y

    def set_profiling(self, on):
        self._check(self.lib.vmx_set_profiling(self._h, int(bool(on))))

    def set_profiling_classes(self, names, stride=1):
        """Time only the named kernel classes (see timings()) while profiling is on - every ``stride``-th launch of them
        (1 .. 16)."""
        index = {self.lib.vmx_kernel_name(i).decode(): i for i in range(VMX_N_KERNELS)}
        mask = 0
        for name in names:
            mask |= 1 << index[name]
        self._check(self.lib.vmx_set_profiling_mask(self._h, mask | ((min(max(int(stride), 1), 16) - 1) << 28)))

    def timings(self, reset=True):
        ms = np.zeros(VMX_N_KERNELS)
        n = np.zeros(VMX_N_KERNELS, dtype=np.int64)
        self._check(self.lib.vmx_get_timings(self._h, _dp(ms), n.ctypes.data_as(C.POINTER(C.c_int64)), int(reset)))
        return {self.lib.vmx_kernel_name(i).decode(): (float(ms[i]), int(n[i])) for i in range(VMX_N_KERNELS)}
