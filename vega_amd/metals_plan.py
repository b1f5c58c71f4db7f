"""`fast_metals` of the reference (vega/metals.py:53, :144-207, :280-321), as a plan for the engine.

With `fast_metals = True` the reference
  * fixes the growth rate of the metal terms to the fiducial one (:280-282);
  * computes every metal x metal correlation at the FIRST evaluation and reuses it for ever (:144-169: the cache
    is keyed by the tracer pair only) - here: a static vector the engine multiplies by the bias product;
  * keeps a per-call cache of the undistorted main x metal correlations whose key is (beta1, beta2, every
    parameter value) but NOT the tracer pair (:188-198): a pair whose betas equal those of an earlier pair of the
    same call receives that earlier pair's xi (computed on the earlier pair's coordinates and tracers) and only
    applies its own metal matrix - here: the later pair reads the earlier pair's pipeline.

The plan is made once, with the parameters of the first evaluation, exactly when the reference fills its caches.
"""

import numpy as np

DEFAULT_GROWTH_RATE = 0.970386      # reference vega/utils.py:60


def needs_freeze(problem):
    return any(item.metals and item.metal_opts['fast_metals'] for item in problem.items.values())


def _beta(params, name):
    """beta of a tracer from two of (bias, bias_eta, beta) (reference vega/utils.py:45-82)."""
    beta = params.get('beta_' + name)
    if beta is None:
        bias, bias_eta = params.get('bias_' + name), params.get('bias_eta_' + name)
        if bias is None or bias_eta is None:
            raise KeyError('For each tracer, you need to specify two of these three: (bias, bias_eta, beta). '
                           f'Offending tracer: {name}')
        beta = bias_eta * params.get('growth_rate', DEFAULT_GROWTH_RATE) / bias
    return beta


def _beta_sources(params, name, beta_name):
    """Names of the parameters the beta of a tracer is read from."""
    key = beta_name or ('beta_' + name)
    if key in params:
        return {key}
    return {'bias_' + name, 'bias_eta_' + name, 'growth_rate'}


def fast_metal_plan(problem, params, metal_xi):
    """({item name: [('pipeline', None) | ('share', leader pair index) | ('static', xi vector), ...]}, pinned) for
    the items with `fast_metals`; `metal_xi(item name, pair index)` returns the pair's correlation (after its metal
    matrix) at `params`, from an evaluation in which every pair had its own pipeline.  `pinned` {name: value} are
    the unsampled parameters whose equality makes two pairs share a pipeline: they must keep these values."""
    sampled = set(problem.sample_params.get('limits', {})) if problem.sample_params else set()
    plan, pinned = {}, {}
    for name, item in problem.items.items():
        if not item.metals or not item.metal_opts['fast_metals']:
            continue
        opts = item.metal_opts
        main = (item.tracer1.name, item.tracer2.name)
        local = dict(params)
        if 'growth_rate' in local and problem.growth_rate is not None:
            local['growth_rate'] = problem.growth_rate
        leaders = {}
        subst = {}
        entries = []
        for mi, pair in enumerate(item.metals):
            n1, n2 = pair.names
            if opts['single_metal_beta']:
                for n in (n1, n2):
                    if n not in main:
                        local[f'beta_{n}'] = local['beta_metals']
                        subst[n] = 'beta_metals'
            if not pair.cross_with_main:
                entries.append(('static', metal_xi(name, mi)))
                continue
            sources = _beta_sources(local, n1, subst.get(n1)) | _beta_sources(local, n2, subst.get(n2))
            key = (_beta(local, n1), _beta(local, n2)) + tuple(local[k] for k in sorted(local))
            if key in leaders:
                lead, lead_sources = leaders[key]
                # equal now is equal for ever only if no sampled parameter can separate the two pairs' betas
                if lead_sources == sources or not ((lead_sources ^ sources) & sampled):
                    entries.append(('share', lead))
                    for src in lead_sources ^ sources:
                        pinned[src] = local[src]
                    continue
            else:
                leaders[key] = (mi, sources)
            entries.append(('pipeline', None))
        plan[name] = entries
    return plan, pinned


# ----------------------------------------------------------------------------------------------------------------
# exact static form of polynomial metal pairs
# ----------------------------------------------------------------------------------------------------------------
def _basis_eligible(problem, item, pair, sampled):
    """A pair can be written xi = Y0 + (beta1 + beta2) Y1 + beta1 beta2 Y2 with static Y when its P(k,mu) is the
    bias-free Kaiser polynomial (reference power_spectrum.py:198-222) times static factors only and nothing else
    that is sampled enters its correlation function.  Returns the names of the unsampled parameters the Y depend on
    (to be pinned), or None."""
    if not item.metal_opts['fast_metal_bias'] or not item.metal_opts['no_metal_decomp']:
        return None
    pk, xi = pair.pipeline.pk, pair.pipeline.xi
    if pk.hcd_model is not None or pk.uvb or pk.heii or pk.small_scale_nl is not None:
        return None
    if pk.fullshape_smoothing is not None or pk.velocity_dispersion is not None or not pk.use_gk:
        return None
    if getattr(xi, 'radiation', False) or getattr(xi, 'relativistic', False) or getattr(xi, 'asymmetry', False):
        return None
    if getattr(xi, 'uv_shotnoise', False) or getattr(problem.scale, 'metal_scaling', False):
        return None
    # (new-bias-evolution only changes the static redshift factors the basis already carries)
    pinned = set()
    for tr in (pair.pipeline.tracer1, pair.pipeline.tracer2):
        if xi.evol_model.get(tr.name, 'standard') != 'standard':
            return None
        pinned.add('alpha_' + tr.name)
    if pinned & sampled:
        return None
    return pinned


def static_basis_plan(problem, engine, theta, base_plan=None):
    """({item name: plan entries}, pinned) in which every eligible metal pair that still owns a pipeline in
    ``base_plan`` (default: every pair) becomes ('basis', [Y0, Y1, Y2]).  The vectors are extracted with ``engine``
    (built with ``base_plan``) - three evaluations of ``theta`` with all bias-free metal betas forced to 0, +1, -1:
        xi(beta) = Y0 + 2 beta Y1 + beta^2 Y2   =>   Y0 = xi(0), Y1 = (xi(1) - xi(-1)) / 4, Y2 = (xi(1) + xi(-1)) / 2 - xi(0)
    - so they carry the pair's metal matrix, redshift evolution and growth factors exactly as the chain applies them.
    ``pinned`` {name: value}: the unsampled parameters the vectors depend on."""
    sampled = set(problem.sample_params.get('limits', {})) if problem.sample_params else set()
    theta = np.asarray(theta, dtype=np.float64)
    params = dict(zip(engine.names, theta))
    plan = {name: list(entries) for name, entries in (base_plan or {}).items()}
    targets = []
    for name, item in problem.items.items():
        if not item.metals or item.metal_opts['single_metal_beta']:
            continue        # the substitution order of metals.py:289-293 is kept by the pipelines
        entries = plan.setdefault(name, [('pipeline', None)] * len(item.metals))
        for mi, pair in enumerate(item.metals):
            kind, arg = entries[mi]
            if kind not in ('pipeline', 'share'):
                continue
            # a pair that reads another pair's pipeline (fast_metals sharing) has that pair's P(k,mu) and tracers
            source = item.metals[arg] if kind == 'share' else pair
            need = _basis_eligible(problem, item, source, sampled)
            if need is not None:
                targets.append((name, mi, need))
    if not targets:
        return plan, {}
    xi = {}
    try:
        for beta in (0.0, 1.0, -1.0):
            engine.set_metal_beta_override(beta)
            engine.eval(theta[None, :])
            for name, mi, _ in targets:
                xi[(name, mi, beta)] = engine.metal_xi(name, mi)
    finally:
        engine.set_metal_beta_override(None)
    pinned = {}
    for name, mi, need in targets:
        y0, up, down = xi[(name, mi, 0.0)], xi[(name, mi, 1.0)], xi[(name, mi, -1.0)]
        plan[name][mi] = ('basis', np.stack([y0, (up - down) / 4.0, (up + down) / 2.0 - y0]))
        for src in need:
            pinned[src] = params[src]
    return plan, pinned
