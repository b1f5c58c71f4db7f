"""vega_amd - MI355X-native model + chi2 engine behind Vega's VegaInterface surface."""
from .interface import VegaInterface  # noqa: F401
from .setup import Grid as Coordinates  # noqa: F401  (reference vega/coordinates.py: the grids a caller hands to a model-only correlation)
from .errors import VegaModelError, VegaBoundsError, VegaArinyoError  # noqa: F401


def run_vega(config_path, search_dirs=(), print_func=print, **engine_args):
    """A complete fit from a main config, as the reference's ``vega.run_vega`` (vega/scripts/run_vega.py:7-51) without
    its plots: initialise, optionally switch to a Monte Carlo mock (``[control] run_montecarlo``), minimise, run the
    ``[chi2 scan]`` if there is one, write the result file (``[output] filename``).  Returns the VegaInterface."""
    vega = VegaInterface(config_path, search_dirs=search_dirs, **engine_args)
    vega.compute_model(run_init=False)
    control = vega.main_config['control'] if 'control' in vega.main_config else None
    run_montecarlo = control is not None and control.getboolean('run_montecarlo', False)
    if run_montecarlo and vega.problem.mc_config is not None:
        vega.initialize_monte_carlo(print_func=print_func)
    elif run_montecarlo:
        raise ValueError('You asked to run over a Monte Carlo simulation, but no "[monte carlo]" section provided.')
    if not vega.sample_params['limits']:
        print_func('No sampled parameters. Skipping minimization.')
    else:
        vega.minimize()
        vega._bestfit_statistics(print_func=print_func)
    scan_results = vega.chi2_scan() if 'chi2 scan' in vega.main_config else None
    params = dict(vega.params)
    if vega.minimizer is not None:
        params.update(vega.bestfit.as_dict(0))
    models = vega.model_components(params) if vega.output.output_cf else vega.models
    vega.output.write_results(vega.bestfit_model if vega.bestfit_model is not None else vega.compute_model(params),
                              params, vega.minimizer, vega.bestfit_corr_stats, scan_results, models)
    return vega
