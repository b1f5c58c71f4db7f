"""Drop-in host interface: the reference's ``VegaInterface`` surface over the vegamx engine.

Mirrors the call signatures, return types and error behaviour of
``VegaInterface.compute_model / chi2 / log_lik`` (reference vega/vega_interface.py:208-387) so
that a minimiser or sampler written against the reference keeps working, and adds the batched
entry points ``chi2_batch`` / ``compute_model_batch`` the GPU is built for.  All arithmetic of the
per-evaluation hot path runs in libvegamx.so; this module only marshals parameters.
"""
import copy

import numpy as np

from . import metals_plan
from .engine_group import make_engine
from .setup import build_problem


class _DataView:
    """The attributes of the reference's ``Data`` object that its callers read
    (reference vega/data.py:169-250, :397, :410)."""

    def __init__(self, item):
        self._item = item
        self.masked_mc_mock = None
        self.scaled_inv_masked_cov = None
        self.scaled_log_cov_det = None

    data_vec = property(lambda self: self._item.data_vec)
    masked_data_vec = property(lambda self: self._item.masked_data_vec)
    data_mask = property(lambda self: self._item.data_mask)
    model_mask = property(lambda self: self._item.model_mask)
    inv_masked_cov = property(lambda self: self._item.inv_masked_cov)
    log_cov_det = property(lambda self: self._item.log_cov_det)
    data_size = property(lambda self: self._item.data_size)
    full_data_size = property(lambda self: self._item.data_vec.size)
    effective_data_size = property(lambda self: self._item.effective_data_size)
    variance = property(lambda self: self._item.variance)
    nb = property(lambda self: self._item.nb)


class _ModelView:
    """``vega.models[name]``: the per-correlation entry points of the reference's ``Model`` that sit below
    ``compute_model`` (reference vega/model.py:157-207), over the same engine.  ``pars`` is a complete local parameter
    dictionary as ``_get_lcl_prms`` hands it to the reference's models - blinding offsets already applied - so no
    transform is applied here."""

    def __init__(self, vega, name):
        self._vega, self._name = vega, name

    def compute(self, pars, pk_full, pk_smooth):
        """Full correlation function of this item from the peak / smooth decomposition of the given linear spectra
        (reference vega/model.py:157-187).  Like the reference it leaves ``pars['peak'] = False`` behind."""
        xi = self._vega._item_model(self._name, pars, pk_full, pk_smooth)
        pars['peak'] = False
        return xi

    def compute_direct(self, pars, pk_full):
        """Model straight from a full linear spectrum, no decomposition (reference vega/model.py:188-207)."""
        xi = self._vega._item_model(self._name, pars, pk_full, None)
        pars['peak'] = False
        return xi


class VegaInterface:
    """GPU-backed stand-in for ``vega.VegaInterface`` restricted to the model + chi2 hot path."""

    def __init__(self, main_path, search_dirs=(), max_batch=256, device=0, problem=None,
                 extra_names=(), kron_metals=True, csr_threshold=None, coordinates=None):
        # coordinates: {name: vega_amd.Coordinates(...)} for correlations WITHOUT a data file - the reference's
        # `corr_item.init_coordinates(...)` (vega/correlation_item.py:120-136); such a problem computes models only
        self.problem = problem if problem is not None else build_problem(main_path, search_dirs, coordinates=coordinates)
        # (reference vega_interface.py:110-118: one correlation without data switches every `data[name]` to None)
        self._has_data = all(getattr(item, 'has_data', True) for item in self.problem.items.values())
        self.main_config = self.problem.main_config
        self.params = self.problem.params
        self.sample_params = self.problem.sample_params
        self.priors = self.problem.priors
        self.corr_items = self.problem.items
        self.data = {name: _DataView(item) if self._has_data else None for name, item in self.problem.items.items()}
        self.fiducial = {'k': self.problem.k, 'pk_full': self.problem.pk_full,
                         'pk_smooth': self.problem.pk_smooth, 'z_eff': self.problem.z_eff,
                         'z_fiducial': self.problem.z_fid, 'Omega_m': self.problem.omega_m,
                         'Omega_de': self.problem.omega_de, 'growth_rate': self.problem.growth_rate}
        self._use_global_cov = self.problem.global_cov is not None
        self.model_pk = bool(self.main_config is not None and 'control' in self.main_config
                             and self.main_config['control'].getboolean('model_pk', False))
        self.monte_carlo = False
        self._mc_active = False
        self._analysis = None
        self._engine_args = dict(max_batch=max_batch, device=device, extra_names=extra_names, kron_metals=kron_metals,
                                 csr_threshold=csr_threshold)
        # parameter-level blinding (reference vega_interface.py:123-127, :853-886): checked before anything is computed
        from .setup import init_blinding
        self._blind, _ = init_blinding(self.problem.items, self.sample_params)
        self._rnsps = None
        self.engine = make_engine(self.problem, **self._engine_args)
        self.param_names = self.engine.names
        # fast_metals: the reference fills its metal caches at the first evaluation (metals_plan.py)
        self._metals_frozen = not metals_plan.needs_freeze(self.problem)
        self._pinned_slots = np.zeros(0, dtype=np.int64)
        self._pinned_values = np.zeros(0)
        self._pinned_names = []
        # marginalisation templates fitted on the fly / reported (reference vega_interface.py:281-292, :546-579)
        self.marginalize_in_fit = any(it.marginalize_in_fit for it in self.problem.items.values())
        self._marg_names = [n for n, it in self.problem.items.items() if it.marg_diff2coeff is not None]
        self._random_marg_coeff = None
        self.models = {name: _ModelView(self, name) for name in self.problem.items}
        # a fit's results (reference vega_interface.py:198, :581-643; written by run_vega through vega.output)
        from .output import Output
        self.output = Output(self.main_config['output'] if self.main_config is not None and 'output' in self.main_config
                             else None, self.problem.items)
        self.bestfit = self.minimizer = None
        self.bestfit_model = self.bestfit_corr_stats = None
        self.chisq = self.reduced_chisq = self.p_value = self.total_data_size = None

    @property
    def analysis(self):
        """The Monte-Carlo / scan driver (the reference's ``vega.analysis``); ``vega.output`` writes its results."""
        return self._analysis

    @analysis.setter
    def analysis(self, driver):
        self._analysis = driver
        if getattr(self, 'output', None) is not None:
            self.output.analysis = driver

    # ------------------------------------------------------------------ blinding
    def set_blinding_offsets(self, offsets):
        """Install the parameter offsets ``{name: v}`` of a blinding file (the reference's ``_rnsps``, read by
        utils.get_blinding:320-372 from collaboration files it only names for some strategies): from then on every
        evaluation - scalar, batched or on device buffers - sees p + pi - exp(v^2) for those parameters and 1 for
        the full-shape scale parameters, model and priors alike (vega_interface.py:389-421).  None removes them."""
        if offsets is not None and not self._blind:
            # the reference asserts the same inconsistency (vega_interface.py:409-413)
            raise AssertionError('Blinding offsets (_rnsps) are set but blinding flag is False.')
        self._rnsps = dict(offsets) if offsets is not None else None
        self._push_blinding()

    def _push_blinding(self):
        if self._rnsps is None:
            self.engine.set_parameter_transform(None, None)
            return
        from .setup import blinding_transform
        self.engine.set_parameter_transform(*blinding_transform(self.param_names, self._rnsps))

    def _blinded(self, theta):
        if self._rnsps is None:
            return theta
        from .setup import blinding_transform
        scale, shift = blinding_transform(self.param_names, self._rnsps)
        return np.where(scale == 1.0, theta + shift, scale * theta + shift)

    # ------------------------------------------------------------------ parameter marshalling
    def _theta(self, params=None):
        return self.engine.theta_from_params(params)

    def theta_matrix(self, params_list):
        """Stack parameter dictionaries (or pass through a [B, n_params] array)."""
        if isinstance(params_list, np.ndarray):
            return np.atleast_2d(params_list)
        return np.stack([self._theta(p) for p in params_list])

    def freeze_metals(self, params=None):
        """`fast_metals`: evaluate once with every metal pair on its own pipeline, keep the metal x metal
        correlations of that evaluation as static vectors and rebuild the engine without their pipelines (and with
        the main x metal pairs the reference's per-call cache merges sharing one).  Called automatically by the
        first evaluation, with that evaluation's parameters - the moment the reference fills its caches.  The
        engine object is replaced: re-read ``self.engine`` afterwards."""
        if self._metals_frozen:
            return
        theta = np.asarray(params, dtype=np.float64) if isinstance(params, np.ndarray) else self._theta(params)
        boot = self.engine
        boot.eval(theta[None, :])
        plan, pinned = metals_plan.fast_metal_plan(self.problem, dict(zip(boot.names, theta)), boot.metal_xi)
        boot.close()
        self.engine = make_engine(self.problem, metal_plan=plan, **self._engine_args)
        assert self.engine.names == self.param_names
        self._push_blinding()
        self._pinned_slots = np.array([self.engine.low.slot[n] for n in pinned], dtype=np.int64)
        self._pinned_values = np.array(list(pinned.values()), dtype=np.float64)
        self._pinned_names = list(pinned)
        self._metals_frozen = True
        self._mc_active = False

    def freeze_static_metals(self, params=None):
        """Opt-in set-up step for samplers and minimisers: metal pairs whose P(k,mu) is the bias-free Kaiser
        polynomial times static factors (no HCD / UV / non-linear / smoothing / velocity-dispersion term, unscaled
        coordinates, unsampled redshift-evolution exponents) are replaced by their exact static form
        xi = Y0 + (beta1 + beta2) Y1 + beta1 beta2 Y2 (metals_plan.static_basis_plan) - no P(k,mu), FFTLog, bin
        evaluation or metal-matrix product per walker for them any more.  Results are unchanged to rounding as
        long as the pinned parameters (the pairs' `alpha_<tracer>`) keep the values they have in ``params``
        (default: the configured ones); evaluations that move them raise.  The engine object is replaced."""
        theta = np.asarray(params, dtype=np.float64) if isinstance(params, np.ndarray) else self._theta(params)
        self.freeze_metals(theta)
        old = self.engine
        plan, pinned = metals_plan.static_basis_plan(self.problem, old, theta, base_plan=old.metal_plan)
        if not pinned and plan == old.metal_plan:
            return
        old.close()
        self.engine = make_engine(self.problem, metal_plan=plan, **self._engine_args)
        assert self.engine.names == self.param_names
        self._push_blinding()
        merged = dict(zip(self._pinned_names, self._pinned_values))
        merged.update(pinned)
        self._pinned_names = list(merged)
        self._pinned_slots = np.array([self.engine.low.slot[n] for n in merged], dtype=np.int64)
        self._pinned_values = np.array(list(merged.values()), dtype=np.float64)
        self._mc_active = False

    def _check_pinned(self, theta):
        """Main x metal pairs that shared the reference's per-call cache at the first evaluation share a pipeline
        here; that stays equivalent only while the unsampled parameters behind their equal betas keep their
        values."""
        if self._pinned_slots.size and not np.array_equal(
                np.broadcast_to(self._pinned_values, (theta.shape[0], self._pinned_values.size)),
                theta[:, self._pinned_slots]):
            raise ValueError('frozen metal terms: ' + ', '.join(self._pinned_names) + ' must keep the values they had '
                             'when the terms were frozen (or be listed in [sample] beforehand)')

    def _direct(self, direct_pk):
        """Context: evaluate with the caller's linear spectrum (one [nk] vector, or [B, nk] for a batch) in place of
        the fiducial template - the reference's ``direct_pk`` argument (vega_interface.py:208-248)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if direct_pk is None:
                yield
                return
            if any(item.metals and not item.metal_opts['no_metal_decomp'] for item in self.problem.items.values()):
                # the metal terms are then computed on the caller's spectrum too (reference model.py:120-123 through
                # compute_direct, :188-207): their pipelines cannot sit on the static basis of the template's spectra
                if getattr(self.engine, 'static_poly', True):
                    self._rebuild_engine(static_poly=False)
            engine = self.engine
            try:
                engine.set_direct_pk(direct_pk)
                yield
            finally:
                engine.set_direct_pk(None)
        return ctx()

    def _rebuild_engine(self, **changes):
        """Replace the engine by one built with other options (the metal plan, blinding offsets and pins carry over; data,
        mocks and covariances are sent again at the next evaluation)."""
        old = self.engine
        plan = old.metal_plan
        old.close()
        self._engine_args.update(changes)
        self.engine = make_engine(self.problem, metal_plan=plan, **self._engine_args)
        assert self.engine.names == self.param_names
        self._push_blinding()
        if self._pinned_names:
            self._pinned_slots = np.array([self.engine.low.slot[n] for n in self._pinned_names], dtype=np.int64)
        self._mc_active = False

    def _sync_monte_carlo(self):
        """chi2 reads the current mock and the scaled inverse covariance in Monte-Carlo mode
        (reference vega/vega_interface.py:296-297, :311-313)."""
        # (reference vega_interface.py:265, :342: `assert self._has_data` - a correlation without a data file has no chi2)
        assert self._has_data, 'a correlation without a data file: models only (chi2, log_lik and fits need data)'
        if self.monte_carlo == self._mc_active and not self.monte_carlo:
            return
        if self.monte_carlo and self._use_global_cov:
            # reference vega_interface.py:296-297: the global mock of the analysis object, in item order
            mock = getattr(getattr(self, 'analysis', None), 'current_mc_mock', None)
            if mock is None:
                raise ValueError('monte_carlo is set but analysis.current_mc_mock is None')
            from .montecarlo import split_global
            for name, part in split_global(self.problem, np.asarray(mock, dtype=np.float64)[None, :]).items():
                self.engine.set_data(name, part[0])
            self._mc_active = True
            return
        for name, view in self.data.items():
            if self.monte_carlo:
                if view.masked_mc_mock is None:
                    raise ValueError(f'monte_carlo is set but data[{name!r}].masked_mc_mock is None')
                self.engine.set_data(name, view.masked_mc_mock)
                if view.scaled_inv_masked_cov is not None and not self._use_global_cov:
                    item = self.problem.items[name]
                    scaled = np.asarray(view.scaled_inv_masked_cov, dtype=np.float64)
                    if item.marginalize_in_fit and item.marg_diff2coeff is not None:
                        # the templates are fitted to the residual against the mock with the map of the UNSCALED covariance
                        # (reference vega_interface.py:282-292, :546-579), chi2 takes the scaled one (:311-313): Q = P^T C_s^-1 P
                        P = item.marg_projector()
                        scaled = P.T.dot(scaled).dot(P)
                    self.engine.set_invcov(name, scaled)
            else:
                self.engine.set_data(name, view.masked_data_vec)
                if self.problem.items[name].cov is not None and not self._use_global_cov:
                    self.engine.set_invcov(name, self.problem.items[name].chi2_matrix)
        self._mc_active = self.monte_carlo

    # ------------------------------------------------------------------ reference surface
    def compute_model(self, params=None, run_init=True, direct_pk=None, marg_coeff=None):
        """dict name -> model correlation function (distorted grid), as the reference returns.  ``marg_coeff``
        (dict name -> template coefficients, as ``chi2(..., return_marg_coeff=True)`` returns them) adds the
        marginalisation templates to the items that have them (reference vega_interface.py:208-248)."""
        self.freeze_metals(params)
        self._check_pinned(self._theta(params)[None, :])
        with self._direct(direct_pk):
            _, status, model = self.engine.eval(self._theta(params)[None, :], want_model=True)
        if status[0]:
            from .errors import VegaModelError
            raise VegaModelError(f'model evaluation failed (status {int(status[0])})')
        if self.model_pk:
            # `model_pk = True` ([control]; reference vega_interface.py:66, model.py:106-107, :186): the models are the
            # multipoles P_ell(k) of the core power spectrum, bao_amp * peak + smooth, [n_ell, nk] per correlation
            pl = self.engine.pk_multipoles(1)
            bao = self._blinded(self._theta(params))[self.engine.low.slot['bao_amp']]
            out = {}
            for name, item in self.problem.items.items():
                n_ell = item.core.xi.ell_max // 2 + 1
                peak, smooth = pl[self.engine.pipe_index[(name, 'peak')]][0], pl[self.engine.pipe_index[(name, 'smooth')]][0]
                out[name] = (smooth if direct_pk is not None else bao * peak + smooth)[:n_ell].copy()
            return out
        out = {name: model[0, sl].copy() for name, sl in self.engine.model_slices.items()}
        if marg_coeff is not None:
            for name, item in self.problem.items.items():
                if item.marg_templates is not None:
                    out[name] += self._templates_times(name, np.asarray(marg_coeff[name], dtype=np.float64)[None, :])[0]
        return out

    def _templates_times(self, name, coeff):
        """coeff [B, n_templates] -> distorted templates . coeff [B, n_dist], on the engine's product kernels."""
        dense = self._template_cache.get(name) if hasattr(self, '_template_cache') else None
        if dense is None:
            tm = self.problem.items[name].marg_templates
            dense = np.ascontiguousarray(tm.toarray() if hasattr(tm, 'toarray') else tm, dtype=np.float64)
            self.__dict__.setdefault('_template_cache', {})[name] = dense
        return self.engine.matmul_host(dense, coeff)

    @property
    def mc_config(self):
        """The ``[monte carlo]`` set-up ({'params', 'sample'}) or None (reference vega_interface.py:140-150)."""
        return self.problem.mc_config

    @property
    def corr_num_marg_modes(self):
        """Modes the small-scale marginalisation removes, per correlation (reference vega_interface.py:181-184)."""
        return {name: item.num_marg_modes for name, item in self.problem.items.items()}

    def compute_marg_coeff(self, model_cf):
        """Best-fit coefficients of the marginalisation templates for given models (reference
        VegaInterface.compute_marg_coeff, vega/vega_interface.py:546-579): per correlation that has templates, the static
        map `marg_diff2coeff` applied to the residual of the data - or of the installed Monte-Carlo mock - against the model
        on the fitted bins.  (A host product on a model the caller already holds; inside chi2 / log_lik the engine forms the
        coefficients itself: ``return_marg_coeff=True``.)"""
        out = {}
        for name, item in self.problem.items.items():
            if item.marg_diff2coeff is None:
                continue
            data = self.data[name].masked_mc_mock if self.monte_carlo else item.masked_data_vec
            out[name] = np.asarray(item.marg_diff2coeff.dot(np.asarray(data) - np.asarray(model_cf[name])[item.model_mask]))
        return out

    def set_fast_metals(self):
        """`fast_metals` is a property of the lowered problem here ([model] fast_metals, frozen at the first evaluation:
        :meth:`freeze_metals`); the reference's switch (vega/vega_interface.py:657-664, not called by its own drivers any
        more, :588) has nothing to flip afterwards."""
        self.freeze_metals()

    def _marg_coeff(self, B):
        """dict name -> [B, n_templates] for the items with marginalisation templates, from the residuals of the
        evaluation that has just run (reference vega_interface.py:546-579: per item, ignoring a global covariance)."""
        return {name: self.engine.marg_coeff(name, B) for name in self._marg_names}

    def chi2(self, params=None, direct_pk=None, return_marg_coeff=False):
        """float chi2; 1e100 when the model cannot be evaluated (reference :268-279).  With ``return_marg_coeff`` the
        reference's tuple ``(chi2, {name: coefficients})``; after a model error ``(1e100, first coefficients ever
        computed, or None)`` (reference :273-279, :285-286)."""
        self.freeze_metals(params)
        self._check_pinned(self._theta(params)[None, :])
        self._sync_monte_carlo()
        # the coefficients are a map of the residuals, which only the full chain forms (a chi2-only evaluation may take
        # the static quadratic form, include/vegamx.h): ask for the model then
        need_coeff = return_marg_coeff or (self.marginalize_in_fit and self._random_marg_coeff is None)
        with self._direct(direct_pk):
            chi2, status, _ = self.engine.eval(self._theta(params)[None, :], want_model=bool(need_coeff and self._marg_names))
        if status[0]:               # the engine's chi2 is the 1e100 sentinel
            return (float(chi2[0]), self._random_marg_coeff) if return_marg_coeff else float(chi2[0])
        if need_coeff:
            coeff = {name: c[0] for name, c in self._marg_coeff(1).items()}
            if self._random_marg_coeff is None:
                self._random_marg_coeff = coeff
            if return_marg_coeff:
                return float(chi2[0]), coeff
        return float(chi2[0])

    def log_lik(self, params=None, direct_pk=None, return_marg_coeff=False):
        """Gaussian log-likelihood with its normalisation (reference :327-387).  With ``return_marg_coeff`` the
        reference's ``(log_lik, coefficients of all items stacked in sorted name order)`` - the closure the PolyChord
        adapter builds (reference vega/samplers/polychord.py:106-113) - or ``(log_lik, None)``."""
        if not return_marg_coeff:
            return float(self._log_norm() - 0.5 * self.chi2(params, direct_pk))
        chi2, coeff = self.chi2(params, direct_pk, True)
        log_lik = float(self._log_norm() - 0.5 * chi2)
        if coeff is None:
            return log_lik, None
        names = sorted(coeff)
        if len(names) > 1:
            return log_lik, np.hstack([coeff[n] for n in names])
        return log_lik, (coeff[names[0]] if names else np.array([]))

    def _item_model(self, name, pars, pk_full, pk_smooth):
        """One item's model for a complete local parameter dictionary and caller-supplied spectra
        (``pk_smooth = None``: the direct form).  The whole engine evaluates; the item's slice is returned."""
        from .errors import VegaModelError
        theta = self.engine.theta_from_params({k: v for k, v in pars.items() if k != 'peak'})
        self.freeze_metals(theta)
        self._check_pinned(theta[None, :])
        eng = self.engine
        own = pk_smooth is not None and not (np.array_equal(pk_full, self.problem.pk_full)
                                             and np.array_equal(pk_smooth, self.problem.pk_smooth))
        if own and eng.metal_plan:
            raise NotImplementedError('frozen metal terms were extracted with the fiducial spectra')
        if self._rnsps is not None:
            eng.set_parameter_transform(None, None)     # `pars` already carries the offsets
        try:
            if own:
                eng.set_linear_spectra(pk_full, pk_smooth)
            with self._direct(pk_full if pk_smooth is None else None):
                _, status, model = eng.eval(theta[None, :], want_model=True)
        finally:
            if own:
                eng.set_linear_spectra(self.problem.pk_full, self.problem.pk_smooth)
            if self._rnsps is not None:
                self._push_blinding()
        if status[0]:
            raise VegaModelError(f'model evaluation failed (status {int(status[0])})')
        return model[0, eng.model_slices[name]].copy()

    def _log_norm(self):
        log_norm = 0.
        for name, item in self.problem.items.items():
            log_norm -= 0.5 * item.data_size * np.log(2 * np.pi)
            if not self._use_global_cov:
                if self.monte_carlo and self.data[name].scaled_log_cov_det is not None:
                    log_norm -= 0.5 * self.data[name].scaled_log_cov_det
                else:
                    log_norm -= 0.5 * item.log_cov_det
        if self._use_global_cov:
            log_norm -= 0.5 * self.problem.global_masks()['log_det']
        for (_, sigma) in self.priors.values():
            log_norm += -0.5 * np.log(2 * np.pi) - np.log(sigma)
        return log_norm

    def compute_prior_chi2(self, params=None):
        theta = self._blinded(self._theta(params))
        total = 0.
        for name, (mean, sigma) in self.priors.items():
            total += (theta[self.engine.low.slot[name]] - mean)**2 / sigma**2
        return total

    # ------------------------------------------------------------------ batched surface
    def chi2_batch_direct(self, params_list, direct_pk):
        """chi2 of B parameter points, each with its own linear spectrum direct_pk[b] (e.g. one Boltzmann-code run per
        point): the batched form of ``chi2(params, direct_pk=...)``.  B <= max_batch."""
        theta = self.theta_matrix(params_list)
        direct_pk = np.atleast_2d(np.asarray(direct_pk, dtype=np.float64))
        if direct_pk.shape[0] != theta.shape[0] or theta.shape[0] > self.engine.max_batch:
            raise ValueError('one spectrum per parameter point, at most max_batch points')
        self.freeze_metals(theta[0])
        self._check_pinned(theta)
        self._sync_monte_carlo()
        with self._direct(direct_pk):
            return self.engine.eval(theta)[0]

    def chi2_batch(self, params_list, return_status=False, return_marg_coeff=False):
        """chi2 for many parameter points: list of dicts or [B, n_params] array (column order
        ``self.param_names``).  Points are evaluated in chunks of ``max_batch``.  ``return_marg_coeff`` appends
        ``{name: [B, n_templates]}`` (rows of walkers whose model failed are NaN)."""
        theta = self.theta_matrix(params_list)
        self.freeze_metals(theta[0])        # fast_metals: the first point of the first batch plays the first call
        self._check_pinned(theta)
        self._sync_monte_carlo()
        out = np.empty(theta.shape[0])
        status = np.empty(theta.shape[0], dtype=np.int32)
        coeff = {name: np.empty((theta.shape[0], self.problem.items[name].marg_diff2coeff.shape[0]))
                 for name in self._marg_names} if return_marg_coeff else None
        mb = self.engine.max_batch
        for lo in range(0, theta.shape[0], mb):
            c, s, _ = self.engine.eval(theta[lo:lo + mb], want_model=bool(return_marg_coeff and self._marg_names))
            out[lo:lo + mb] = c
            status[lo:lo + mb] = s
            if return_marg_coeff:
                for name, block in self._marg_coeff(c.size).items():
                    block[s != 0] = np.nan
                    coeff[name][lo:lo + mb] = block
        res = (out, status) if return_status else (out,)
        if return_marg_coeff:
            res = res + (coeff,)
        return res if len(res) > 1 else res[0]

    def chi2_batch_device(self, theta, out=None):
        """chi2 of walkers that are already in HBM: ``theta`` a CUDA float64 tensor [n, n_params] (column order
        ``self.param_names``) on the engine's device -> CUDA tensor [n].  Nothing crosses PCIe: chunks of ``max_batch``
        go through ``vmx_eval_device`` on the engine's stream, which is ordered after the caller's current torch stream
        and before whatever that stream does next (events, no host synchronisation).  Failed walkers carry the
        reference's 1e100 sentinel.  Frozen-metal pins are the caller's contract here (no host copy to check)."""
        import torch
        if not (theta.is_cuda and theta.dtype == torch.float64 and theta.dim() == 2 and theta.is_contiguous()
                and theta.shape[1] == len(self.param_names)):
            raise ValueError(f'theta: contiguous CUDA float64 tensor [n, {len(self.param_names)}]')
        if not self._metals_frozen:
            self.freeze_metals(theta[0].cpu().numpy())      # fast_metals: one walker plays the reference's first call
        self._sync_monte_carlo()
        eng = self.engine
        n = theta.shape[0]
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=theta.device)
        current = torch.cuda.current_stream(theta.device)
        if getattr(eng, 'lanes', 1) > 1:
            current.synchronize()       # (two lanes, two streams: the walkers are complete before either lane reads them)
        else:
            torch.cuda.ExternalStream(eng.stream_handle(), device=theta.device).wait_event(current.record_event())
        mb = eng.max_batch
        if hasattr(eng, 'eval_device_tensor'):      # (one engine per transform setting: engine_group.py)
            for lo in range(0, n, mb):
                eng.eval_device_tensor(theta[lo:lo + mb], out[lo:lo + mb])
            return out
        for lo in range(0, n, mb):
            hi = min(lo + mb, n)
            eng.eval_device(theta[lo:hi].data_ptr(), hi - lo, out[lo:hi].data_ptr())
            # (with two lanes consecutive chunks run on alternating streams: the caller's stream waits for each)
            current.wait_event(torch.cuda.ExternalStream(eng.last_stream_handle(), device=theta.device).record_event())
        return out

    def log_lik_batch(self, params_list):
        return self._log_norm() - 0.5 * self.chi2_batch(params_list)

    def compute_model_batch(self, params_list):
        theta = self.theta_matrix(params_list)
        self.freeze_metals(theta[0])
        self._check_pinned(theta)
        blocks = []
        mb = self.engine.max_batch
        for lo in range(0, theta.shape[0], mb):
            blocks.append(self.engine.eval(theta[lo:lo + mb], want_model=True)[2])
        model = np.concatenate(blocks)
        return {name: model[:, sl] for name, sl in self.engine.model_slices.items()}

    # ------------------------------------------------------------------ fits (SURVEY section 8f "next" #1, #2)
    def minimize(self, params=None, tol=0.1, method='migrad'):
        """Fit the sampled parameters to the data (reference VegaInterface.minimize -> Minimizer.minimize,
        vega/vega_interface.py:581, vega/minimizer.py:39-103).  Returns a FitResult with one fit; the best fit
        is also kept in ``self.bestfit``."""
        from .montecarlo import MonteCarlo
        self.freeze_metals()
        self._sync_monte_carlo()
        # `params` overrides entries of the sampling table, as Minimizer.minimize(params) does for chi2 scans
        # (reference vega/minimizer.py:39-64: values / errors / limits / fix)
        sample = {key: dict(self.sample_params.get(key, {})) for key in ('limits', 'values', 'errors', 'fix')}
        for key in sample:
            if params is not None and key in params:
                sample[key].update({n: v for n, v in params[key].items() if n in sample['limits']})
        driver = MonteCarlo(self)
        fitter = driver.minimizer(sample, tol=tol, method=method)
        from .minimizer import MinimizerView
        self.bestfit = fitter.minimize(n_fits=1, fixed=driver._fixed)
        self.minimizer = MinimizerView(self.bestfit)        # (the reference's names: minimizer.values[par], .fmin.fval, .minuit.valid)
        self._bestfit_statistics()
        return self.bestfit

    def _bestfit_statistics(self, print_func=None):
        """What the reference's ``minimize`` leaves behind next to the fit (vega/vega_interface.py:593-643): the best-fit
        model (through the engine), every correlation's own chi2 / reduced chi2 / p-value on its fitted bins and the best-fit
        coefficients of its marginalisation templates (added to its best-fit model), and the totals.  The per-correlation
        chi2 is a statistic of the written result - one product per correlation on the host, once per fit - not the
        fitted function (that is the engine's, ``self.chisq = fmin.fval``)."""
        from scipy import stats
        fit = self.bestfit
        values = {**self.params, **fit.as_dict(0)}
        self.bestfit_model = self.compute_model(values, run_init=False)
        self.chisq = float(fit.fval[0])
        if self.model_pk:           # (the models are P_ell(k) there: nothing to compare with the data bins)
            self.bestfit_corr_stats = None
            return
        self.total_data_size = 0
        self.bestfit_corr_stats = {}
        num_pars = len(self.sample_params['limits'])
        for name, item in self.problem.items.items():
            view = self.data[name]
            size = item.effective_data_size
            self.total_data_size += size
            coeff = None
            if self.monte_carlo and self._use_global_cov:
                chisq = 0.                                                     # (as the reference: :603-605)
                diff = None
            else:
                if self.monte_carlo:
                    diff = np.asarray(view.masked_mc_mock) - self.bestfit_model[name][item.model_mask]
                    inv = view.scaled_inv_masked_cov if view.scaled_inv_masked_cov is not None else item.inv_masked_cov
                else:
                    diff = item.masked_data_vec - self.bestfit_model[name][item.model_mask]
                    inv = item.inv_masked_cov
                chisq = float(diff.dot(inv.dot(diff)))
            if item.marg_diff2coeff is not None and diff is not None:
                coeff = np.asarray(item.marg_diff2coeff.dot(diff))
                self.bestfit_model[name] = self.bestfit_model[name] + np.asarray(item.marg_templates.dot(coeff)).ravel()
            dof = size - num_pars
            self.bestfit_corr_stats[name] = {'masked_size': int(size), 'chisq': chisq, 'reduced_chisq': chisq / dof,
                                             'p_value': float(1 - stats.chi2.cdf(chisq, dof)), 'bestfit_marg_coeff': coeff}
            if print_func is not None:
                print_func(f'{name} chi^2/(ndata-nparam): {chisq:.1f}/({size}-{num_pars}) = {chisq / dof:.3f}')
        self.chisq = float(fit.fval[0])
        dof = self.total_data_size - num_pars
        self.reduced_chisq = self.chisq / dof
        self.p_value = float(1 - stats.chi2.cdf(self.chisq, dof))
        if print_func is not None:
            print_func(f'Total chi^2/(ndata-nparam): {self.chisq:.1f}/({self.total_data_size}-{num_pars}) '
                       f'= {self.reduced_chisq:.3f}, PTE={self.p_value:.2f}')
            if not bool(fit.is_valid[0]):
                print_func('Invalid fit!!! Check data, covariance, model and priors.')

    def model_components(self, params=None):
        """The saved components of the reference's models (``save-components``: vega/model.py:41-45, :113-115, :151-153; metal
        terms with the default ``no-metal-decomp = True``: inside the smooth component's final model, :117-119): dict name -> {'xi': {'peak': {'core': ...}, 'smooth': {'core': ...}},
        'xi_distorted': {...}} - `xi`: the raw core correlation of the peak / smooth spectrum on the model grid, `xi_distorted`:
        the component's final model (broadband, distortion).  One evaluation of two walkers, ``bao_amp`` = 0 and 1: the model is
        affine in it (model = bao_amp * peak + smooth, vega/model.py:157-187) and the raw correlations are the per-pipeline
        bins of that evaluation (stage taps).  P(k, mu) grids (`write_pk`) are never formed here."""
        from .engine import Engine
        if self.model_pk or not isinstance(self.engine, Engine):
            raise NotImplementedError('model_components: one engine, correlation-function models')
        for name, item in self.problem.items.items():
            if item.metals and not item.metal_opts['no_metal_decomp']:
                # (`no-metal-decomp = False`: the reference merges every metal pair's saved correlations into the components,
                # vega/model.py:120-130; with the default the metal terms sit inside the smooth component's final model)
                raise NotImplementedError(f'{name}: with no-metal-decomp = False the metal pairs are components of their own')
        eng = self.engine
        self.freeze_metals(params)
        base = {**self.params, **(params or {})}
        theta = self.theta_matrix([{**base, 'bao_amp': 0.}, {**base, 'bao_amp': 1.}])
        self._check_pinned(theta)
        _, status, model = eng.eval(theta, want_model=True)
        if status.any():
            from .errors import VegaModelError
            raise VegaModelError(f'model evaluation failed (status {int(status[status != 0][0])})')
        out = {}
        for name, sl in eng.model_slices.items():
            n = self.problem.items[name].model_grid.size
            n_pad = (n + 31) // 32 * 32
            raw = {comp: eng.debug_read(1, eng.pipe_index[(name, comp)], 2 * n_pad).reshape(2, n_pad)[1, :n].copy()
                   for comp in ('peak', 'smooth')}
            out[name] = {'xi': {'peak': {'core': raw['peak']}, 'smooth': {'core': raw['smooth']}},
                         'xi_distorted': {'peak': {'core': model[1, sl] - model[0, sl]}, 'smooth': {'core': model[0, sl].copy()}}}
        return out

    def compute_sensitivity(self, nominal=None, frac=0.1, verbose=True, print_func=print):
        """Sensitivity of the model to the floating parameters (reference VegaInterface.compute_sensitivity,
        vega/vega_interface.py:956-1075): central differences at value +- frac * error of the four parts of every
        correlation - [0, 0] bao_amp x the peak component's final model, [0, 1] the smooth component's, [1, 0] bao_amp x the
        raw core correlation of the peak spectrum, [1, 1] that of the smooth one - and the Fisher information per bin of every
        parameter pair, distorted and not; kept in ``self.sensitivity`` (keys `nominal`, `partials`, `fisher`).

        All 2 P parameter points go through the engine in batches of eight, each twice: the model is affine in ``bao_amp``
        (model = bao_amp * peak + smooth, reference vega/model.py:157-187), so the two components' final models are
        model(bao_amp = 1) - model(bao_amp = 0) and model(bao_amp = 0); the raw core correlations are the per-pipeline bins of
        the same evaluation (the stage taps).  ``nominal``: {name: (value, error)}; default: the last fit's."""
        from .engine import Engine
        if nominal is None:
            if self.bestfit is None:
                raise RuntimeError('No nominal parameter values provided or saved by minimize()')
            nominal = {n: (float(v), float(e)) for n, v, e in zip(self.bestfit.names, self.bestfit.values[0], self.bestfit.errors[0])}
        if self.model_pk or not isinstance(self.engine, Engine):
            raise NotImplementedError('compute_sensitivity: one engine, correlation-function models')
        eng = self.engine
        self.freeze_metals()
        params = copy.deepcopy(self.params)
        for pname, (pvalue, _) in nominal.items():
            params[pname] = pvalue
        bao_amp = self.params['bao_amp']
        rows = []
        for pname, (pvalue, perror) in nominal.items():
            for sign in (+1, -1):
                for bao in (0., 1.):
                    rows.append({**params, pname: pvalue + sign * frac * perror, 'bao_amp': bao})
        theta = self.theta_matrix(rows)
        self._check_pinned(theta)
        sizes = {name: item.model_grid.size for name, item in self.problem.items.items()}
        model = np.empty((len(rows), eng.model_size))
        raw = {(name, comp): np.empty((len(rows), sizes[name])) for name in sizes for comp in ('peak', 'smooth')}
        chunk = min(8, eng.max_batch)       # (up to 8 walkers the engine keeps the per-pipeline bins of a model evaluation: the stage taps)
        for lo in range(0, len(rows), chunk):
            hi = min(lo + chunk, len(rows))
            _, status, model[lo:hi] = eng.eval(theta[lo:hi], want_model=True)
            if status.any():
                from .errors import VegaModelError
                raise VegaModelError(f'model evaluation failed (status {int(status[status != 0][0])})')
            for (name, comp), out in raw.items():
                n_pad = (sizes[name] + 31) // 32 * 32
                out[lo:hi] = eng.debug_read(1, eng.pipe_index[(name, comp)], (hi - lo) * n_pad).reshape(hi - lo, n_pad)[:, :sizes[name]]
        self.sensitivity = dict(nominal=copy.deepcopy(nominal), partials={n: {} for n in sizes}, fisher={n: {} for n in sizes})
        for pindex, (pname, (pvalue, perror)) in enumerate(nominal.items()):
            if verbose:
                print_func(f'Calculating sensitivity for [{pindex}] {pname} at {pvalue:.4f} ± {perror:.4f}')
            delta = frac * perror
            plus0, plus1, minus0, minus1 = (4 * pindex + i for i in range(4))       # (+, bao 0), (+, bao 1), (-, bao 0), (-, bao 1)
            for name, sl in eng.model_slices.items():
                dist = model[:, sl]
                if dist.shape[1] != sizes[name]:
                    raise ValueError(f'{name}: the distorted and the model grids differ in size (the reference adds both into one array)')
                part = np.zeros((2, 2, sizes[name]))
                part[0, 0] = bao_amp * ((dist[plus1] - dist[plus0]) - (dist[minus1] - dist[minus0]))
                part[0, 1] = dist[plus0] - dist[minus0]
                part[1, 0] = bao_amp * (raw[(name, 'peak')][plus1] - raw[(name, 'peak')][minus1])
                part[1, 1] = raw[(name, 'smooth')][plus1] - raw[(name, 'smooth')][minus1]
                self.sensitivity['partials'][name][pname] = part / (2 * delta)
        if verbose:
            print_func('Computing Fisher information for each pair of parameters...')
        names = list(nominal)
        for i1, p1 in enumerate(names):
            for p2 in names[i1:]:
                for name, item in self.problem.items.items():
                    mask = item.data_mask
                    fisher = np.full((2, sizes[name]), np.nan)
                    for idistort in range(2):
                        d1 = self.sensitivity['partials'][name][p1][idistort].sum(axis=0)
                        d2 = self.sensitivity['partials'][name][p2][idistort].sum(axis=0)
                        fisher[idistort, mask] = d1[mask] * item.inv_masked_cov.dot(d2[mask])
                    self.sensitivity['fisher'][name][(p1, p2)] = fisher
        return self.sensitivity

    def chi2_scan(self, method='migrad'):
        """``[chi2 scan]`` of the reference (vega/analysis.py:53-122; ``vega.analysis.chi2_scan()`` there): every grid point's
        fit in lock-step - see :meth:`vega_amd.montecarlo.MonteCarlo.chi2_scan`."""
        from .montecarlo import MonteCarlo
        if getattr(self, 'analysis', None) is None:
            self.analysis = MonteCarlo(self)
        return self.analysis.chi2_scan(method=method)

    def run_monte_carlo(self, fiducial_model=None, num_mocks=1, seed=0, scale=None, forecast=False,
                        run_mc_fits=True, sample_params=None, method='migrad'):
        """Create ``num_mocks`` mocks around ``fiducial_model`` (default: the model at the current parameters)
        and fit them all in lock-step (reference Analysis.run_monte_carlo, vega/analysis.py:224-308)."""
        from .montecarlo import MonteCarlo
        self.freeze_metals()
        if fiducial_model is None:
            fiducial_model = self.compute_model()
        self.analysis = MonteCarlo(self)
        return self.analysis.run_monte_carlo(fiducial_model, num_mocks=num_mocks, seed=seed, scale=scale,
                                             forecast=forecast, run_mc_fits=run_mc_fits,
                                             sample_params=sample_params, method=method)

    def get_fiducial_for_monte_carlo(self, print_func=print):
        """The fiducial model the mocks are drawn around (reference vega_interface.py:448-503): the [mc parameters]
        on top of a fit to the data when parameters are sampled; `use_measured_fiducial` reads it from files,
        `use_full_pk_for_mc` computes it from the full spectrum directly."""
        if self.problem.mc_config is None:
            raise ValueError('No Monte Carlo config provided: add a [monte carlo] section')
        control = self.main_config['control'] if 'control' in self.main_config else {}
        mc_params = dict(self.problem.mc_config['params'])
        start_from_fit = control.get('mc_start_from_fit', None)
        if start_from_fit is not None:
            # reference vega_interface.py:465-472: the best-fit values of an existing fit file (its BESTFIT table:
            # vega/postprocess/fit_results.py:47-53) under the [mc parameters]
            from .tables import find_file, read_tables
            print_func(f'Reading input fit {start_from_fit}')
            bestfit = None
            for table in read_tables(find_file(start_from_fit, self.problem.search_dirs)):
                if str(table.header.get('EXTNAME', '')).strip().upper() == 'BESTFIT':
                    bestfit = table
            if bestfit is None:
                raise ValueError(f'{start_from_fit}: no BESTFIT table')
            fit_names = [n.decode() if isinstance(n, bytes) else str(n) for n in bestfit.data['names']]
            fit_values = np.asarray(bestfit.data['values'], dtype=float).reshape(len(fit_names), -1)[:, 0]
            mc_params = {**{n.strip(): float(v) for n, v in zip(fit_names, fit_values)}, **mc_params}
            print_func(f'Set template parameters to {mc_params}.')
        elif self.sample_params['limits']:
            print_func('Running initial fit')
            res = self.minimize()
            mc_params = {**res.as_dict(), **mc_params}
            print_func(f'Set template parameters to {mc_params}.')
        if getattr(control, 'getboolean', None) and control.getboolean('use_measured_fiducial', False):
            from .tables import find_file, read_tables
            return {name: np.asarray(read_tables(find_file(control.get(f'mc_fiducial_{name}'),
                                                           self.problem.search_dirs))[0].data['DA'], dtype=float)
                    for name in self.corr_items}
        if getattr(control, 'getboolean', None) and control.getboolean('use_full_pk_for_mc', False):
            return self.compute_model(mc_params, direct_pk=self.fiducial['pk_full'])
        return self.compute_model(mc_params)

    def initialize_monte_carlo(self, scale=None, print_func=print):
        """One mock per correlation around the Monte-Carlo fiducial, installed as the data every following chi2 / fit
        reads (reference VegaInterface.initialize_monte_carlo, vega/vega_interface.py:505-544).  The ONLY place that
        reads ``[control] global_cov_rescale`` (:531-533) - ``run_monte_carlo`` passes its scale through unchanged, as
        ``Analysis.run_monte_carlo`` -> ``create_global_monte_carlo(scale=None)`` does."""
        from .montecarlo import MonteCarlo
        self.freeze_metals()
        fiducial_model = self.get_fiducial_for_monte_carlo(print_func)
        control = self.main_config['control']
        if self.problem.mc_config is not None:
            self.sample_params = self.problem.mc_config['sample']       # "Reset the minimizer" (:523-525)
        forecast = control.getboolean('forecast', False)
        seed = control.getint('mc_seed', 0)
        if self._use_global_cov and scale is None and 'global_cov_rescale' in control:
            scale = control.getfloat('global_cov_rescale')
        self.analysis = MonteCarlo(self)
        mocks = self.analysis.create_mocks(fiducial_model, 1, seed=seed, scale=scale, forecast=forecast,
                                           reseed_per_item=True)
        out = self.analysis.install_mocks(mocks, scale)
        self.monte_carlo = True
        return out

    def close(self):
        self.engine.close()
