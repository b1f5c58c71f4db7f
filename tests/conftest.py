import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


_SPAWNER = None


def run_programs(commands, envs, timeout=900):
    """Run ``commands`` (argv lists) concurrently with the extra environments ``envs`` through the helper process that
    was started before this session touched the GPU (tests/helpers/spawn_server.py); [(returncode, output)]."""
    import json
    import subprocess
    if _SPAWNER is None:        # CPU-only session: nothing here holds a device, start them directly
        procs = [subprocess.Popen(argv, env={**os.environ, **env}, cwd=str(REPO), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for argv, env in zip(commands, envs)]
        return [(p.wait(timeout=timeout), p.stdout.read()) for p in procs]
    _SPAWNER.stdin.write(json.dumps({'commands': commands, 'env': envs, 'cwd': str(REPO), 'timeout': timeout}) + '\n')
    _SPAWNER.stdin.flush()
    # the helper enforces `timeout` on the programs; should the helper itself die or hang, do not wait for its reply for ever
    import select
    ready, _, _ = select.select([_SPAWNER.stdout], [], [], timeout + 120)
    line = _SPAWNER.stdout.readline() if ready else ''
    if not line:
        raise RuntimeError(f'tests/helpers/spawn_server.py gave no reply within {timeout + 120} s (exit code {_SPAWNER.poll()})')
    reply = json.loads(line)
    return [(r['returncode'], r['output']) for r in reply]


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # multi-process GPU tests start their ranks through a helper that never holds the device: it has to exist before
    # this process initialises the GPU (counting devices does not initialise it)
    global _SPAWNER
    try:
        import torch
        if _SPAWNER is None and torch.cuda.device_count() > 0:
            import subprocess
            _SPAWNER = subprocess.Popen([sys.executable, str(REPO / 'tests' / 'helpers' / 'spawn_server.py')],
                                        stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=str(REPO))
    except Exception:       # pragma: no cover
        _SPAWNER = None
    # torch bundles its own HIP runtime: when a test uses both torch device tensors and libvegamx.so, torch must
    # bring the runtime up first (as bench.py does), otherwise it finds no device in a process that already loaded
    # the system runtime through the engine.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # pragma: no cover - CPU-only container
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


_PROBLEMS = {}


def load_problem(name, **kw):
    """Cached Problem for tests/golden/configs/<name>/main.ini (or an explicit main file)."""
    from vega_amd.setup import build_problem
    key = (name, tuple(sorted(kw.items())))
    if key not in _PROBLEMS:
        main = name if name.endswith('.ini') else f'configs/{name}/main.ini'
        _PROBLEMS[key] = build_problem(main, search_dirs=[GOLDEN], **kw)
    return _PROBLEMS[key]


@pytest.fixture(scope='session')
def problem_loader():
    return load_problem


def fits_ingest_problem(tmp_path):
    """The auto-correlation config pointed at a FITS data file that carries the synthetic distortion matrix and
    covariance (written with the package's own FITS writer in the reference's layout)."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'cf_lya-exp.npz')
    data_path = synthetic.write_data_file(tmp_path / 'cf_lya-synth.fits', source)
    cfg = tmp_path / 'configs' / 'ingest'
    cfg.mkdir(parents=True)
    main = (GOLDEN / 'configs' / 'auto' / 'main.ini').read_text()
    (cfg / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/ingest/lyalya_lyalya.ini', main))
    item = (GOLDEN / 'configs' / 'auto' / 'lyalya_lyalya.ini').read_text()
    (cfg / 'lyalya_lyalya.ini').write_text(re.sub(r'filename = .*', f'filename = {data_path}', item, count=1))
    return build_problem('configs/ingest/main.ini', search_dirs=[tmp_path, GOLDEN])


def dmat_file_problem(tmp_path, coef=2, marg_options=None, config='auto'):
    """The auto-correlation config with its distortion matrix in a separate `distortion-file` (model grid COEFMOD times
    finer than the data grid) and a `covariance-file` - the files tests/golden/make_golden.py::dump_dmat_file gave the
    reference (reference vega/data.py:441-473)."""
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    main = synthetic.dmat_file_configs(tmp_path, GOLDEN, config=config, coef=coef, item_options=marg_options)
    return build_problem(main, search_dirs=[tmp_path, GOLDEN])


def blinding_problem(tmp_path, sample_extra=''):
    """The auto-correlation config on a `desi_dr3` data file (BLINDING header, DA_BLIND column next to DA; the file
    tests/golden/make_golden.py::dump_blinding gave the reference) with a prior on a sampled parameter."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'cf_lya-exp.npz')
    data_path = synthetic.write_data_file(tmp_path / 'cf_lya-blind.fits', source, with_distortion=False,
                                          extra_header={'BLINDING': 'desi_dr3'},
                                          blind_data=synthetic.blinded_data_vector(source[0].data['DA']))
    cfg = tmp_path / 'configs' / 'blind'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (GOLDEN / 'configs' / 'auto' / 'main.ini').read_text()
    main = re.sub(r'ini files = .*', 'ini files = configs/blind/lyalya_lyalya.ini', main)
    main = main.replace('[sample]', '[sample]\n' + sample_extra) + '\n[priors]\nbeta_LYA = gaussian 1.6 0.1\n'
    (cfg / 'main.ini').write_text(main)
    item = (GOLDEN / 'configs' / 'auto' / 'lyalya_lyalya.ini').read_text()
    (cfg / 'lyalya_lyalya.ini').write_text(re.sub(r'filename = .*', f'filename = {data_path}', item, count=1))
    return build_problem('configs/blind/main.ini', search_dirs=[tmp_path, GOLDEN])


def new_bias_evol_problem(tmp_path):
    """The cross-correlation with `new-bias-evolution = True` on a data file that carries a picca cosmology in its
    header (the file tests/golden/make_golden.py::dump_new_bias_evol gave the reference)."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'xcf_lya-exp.npz')
    data_path = synthetic.write_data_file(tmp_path / 'xdata.fits', source, with_distortion=False,
                                          with_covariance=False, extra_header=synthetic.PICCA_COSMOLOGY_HEADER)
    cfg = tmp_path / 'configs' / 'biasevol'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text()
    (cfg / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/biasevol/lyalya_qso.ini', main))
    text = (GOLDEN / 'configs' / 'joint' / 'lyalya_qso.ini').read_text()
    text = re.sub(r'filename = .*', f'filename = {data_path}', text, count=1)
    (cfg / 'lyalya_qso.ini').write_text(text.replace('[model]', '[model]\nnew-bias-evolution = True'))
    return build_problem('configs/biasevol/main.ini', search_dirs=[tmp_path, GOLDEN])


NEW_METALS_CASES = {'auto': ('auto_metals', 'lyalya_lyalya', 'cf_lya-exp.npz', False),
                    'cross': ('joint_metals', 'lyalya_qso', 'xcf_lya-exp.npz', False),
                    'auto_rp': ('auto_metals', 'lyalya_lyalya', 'cf_lya-exp.npz', True)}


def new_metals_problem(tmp_path, tag):
    """One correlation with `new_metals = True`: metal matrices built at set-up from the synthetic stacked-delta file
    and object catalogue (the inputs tests/golden/make_golden.py::dump_new_metals gave the reference)."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    from vega_amd.tables import read_tables
    config, item_name, data_file, rp_only = NEW_METALS_CASES[tag]
    source = read_tables(GOLDEN / 'inputs' / data_file)
    data_path = synthetic.write_data_file(tmp_path / 'data.fits', source, with_distortion=False,
                                          with_covariance=False, extra_header=synthetic.PICCA_COSMOLOGY_HEADER)
    stack = synthetic.write_stacked_deltas(tmp_path / 'stack.fits')
    cat = synthetic.write_object_catalog(tmp_path / 'cat.fits')
    cfg = tmp_path / 'configs' / 'newmetals'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (GOLDEN / 'configs' / config / 'main.ini').read_text()
    (cfg / 'main.ini').write_text(re.sub(r'ini files = .*', f'ini files = configs/newmetals/{item_name}.ini', main))
    text = (GOLDEN / 'configs' / config / f'{item_name}.ini').read_text()
    text = re.sub(r'filename = .*', f'filename = {data_path}\nweights-tracer1 = {stack}\n'
                  f'weights-tracer2 = {cat if "qso" in item_name else stack}', text, count=1)
    text = text.replace('[model]', '[model]\nnew_metals = True' + ('\nrp_only_metal_mats = True' if rp_only else ''))
    (cfg / f'{item_name}.ini').write_text(text + '\n' + synthetic.METAL_MATRIX_SECTION)
    return build_problem('configs/newmetals/main.ini', search_dirs=[tmp_path, GOLDEN]), item_name


def marginalization_problem(tmp_path, options, in_fit=False):
    """fits_ingest_problem with small-scale marginalisation options added to the [model] section (and the templates
    fitted on the fly instead of folded into the covariance when ``in_fit``)."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'cf_lya-exp.npz')
    data_path = synthetic.write_data_file(tmp_path / 'cf_lya-synth.fits', source)
    cfg = tmp_path / 'configs' / 'marg'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (GOLDEN / 'configs' / 'auto' / 'main.ini').read_text()
    if in_fit:
        main = main.replace('[control]', '[control]\nmarginalize-in-fit = True')
    (cfg / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/marg/lyalya_lyalya.ini', main))
    item = (GOLDEN / 'configs' / 'auto' / 'lyalya_lyalya.ini').read_text()
    item = re.sub(r'filename = .*', f'filename = {data_path}', item, count=1).replace('[model]', '[model]\n' + options)
    (cfg / 'lyalya_lyalya.ini').write_text(item)
    return build_problem('configs/marg/main.ini', search_dirs=[tmp_path, GOLDEN])


MARGINALIZATION_CASES = {
    'rtmax': 'marginalize-below-rtmax = 16.0\nmarginalize-prior-sigma = 5.0',
    'allrmin': 'marginalize-all-rmin-cuts = True\nmarginalize-match-data-bins = True',
    'fitscales': 'marginalize-below-rtmax = 12.0\nfit-marginalized-scales = True\nmarginalize-match-data-bins = True',
}


def synth_joint_problem(with_global_cov=False, tmp_path=None):
    """The joint auto + cross config with the seeded dense distortion matrices and covariances of vega_amd.synthetic
    (what tests/golden/make_golden.py injects into the reference); optionally with the synthetic global covariance -
    read back from a `global-cov-file` when ``tmp_path`` is given, set on the Problem otherwise."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    main = 'configs/joint/main.ini'
    dirs = [GOLDEN]
    if with_global_cov and tmp_path is not None:
        base = build_problem(main, search_dirs=dirs)
        grids = [(it.data_grid.rp, it.data_grid.rt) for it in base.items.values()]
        path = synthetic.write_global_covariance(tmp_path / 'global_cov.fits', synthetic.global_covariance(grids))
        cfg = tmp_path / 'configs' / 'globalcov'
        cfg.mkdir(parents=True, exist_ok=True)
        text = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text()
        (cfg / 'main.ini').write_text(text.replace('[data sets]', f'[data sets]\nglobal-cov-file = {path}'))
        main, dirs = 'configs/globalcov/main.ini', [tmp_path, GOLDEN]
    prob = build_problem(main, search_dirs=dirs)
    for item in prob.items.values():
        item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt),
                            inv_masked_cov=synthetic.inverse_masked_covariance(item.data_grid.rp, item.data_grid.rt, item.data_mask))
    if with_global_cov and tmp_path is None:
        prob.global_cov = synthetic.global_covariance([(it.data_grid.rp, it.data_grid.rt)
                                                       for it in prob.items.values()])
    return prob


def model_only_problem(tmp_path):
    """The joint auto + cross config with `has_datafile = False` in both correlations and the coordinates handed in by the caller
    - what tests/golden/make_golden.py::dump_model_only gave the reference (vega/correlation_item.py:40-42, :120-136)."""
    import re
    import numpy as np
    from vega_amd import Coordinates
    from vega_amd.setup import build_problem
    cfg = tmp_path / 'configs' / 'modelonly'
    cfg.mkdir(parents=True)
    main = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text()
    (cfg / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/modelonly/lyalya_lyalya.ini configs/modelonly/lyalya_qso.ini', main))
    for it in ('lyalya_lyalya', 'lyalya_qso'):
        text = (GOLDEN / 'configs' / 'joint' / f'{it}.ini').read_text()
        (cfg / f'{it}.ini').write_text(re.sub(r'filename = .*', 'has_datafile = False', text, count=1))
    exp = np.load(GOLDEN / 'expected_model_only.npz')
    coordinates = {'lyalya_lyalya': Coordinates(0., 120., 100., 30, 25),                 # (z: the effective redshift)
                   'lyalya_qso': Coordinates(-80., 80., 80., 40, 20, z=exp['cross/z'])}
    return build_problem('configs/modelonly/main.ini', search_dirs=[tmp_path, GOLDEN], coordinates=coordinates), coordinates


def config1_problem():
    """BASELINE configs[1] as stated: Lya x Lya auto-correlation only, ell = 0, 2, 4, dense synthetic 2500^2
    distortion matrix and covariance."""
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    prob = build_problem('configs/auto_ell4/main.ini', search_dirs=[GOLDEN])
    for item in prob.items.values():
        item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
    return prob


def options2_problem(tmp_path, which):
    """Configs of tests/golden/make_golden.py::dump_options2: 'cross' (rescale-coords-systematics + old_growth_func +
    fht_lowring = False on the QSO x Lya item), 'auto' (UV shot noise with rescale-coords-systematics), 'model_pk'."""
    import re
    from vega_amd.setup import build_problem
    cfg = tmp_path / 'configs' / f'opt2_{which}'
    cfg.mkdir(parents=True, exist_ok=True)
    if which == 'model_pk':
        main = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text().replace('[control]', '[control]\nmodel_pk = True')
        if '[control]' not in main:
            main += '\n[control]\nmodel_pk = True\n'
        for it in ('lyalya_lyalya', 'lyalya_qso'):
            (cfg / f'{it}.ini').write_text((GOLDEN / 'configs' / 'joint' / f'{it}.ini').read_text())
        main = re.sub(r'ini files = .*', f'ini files = configs/opt2_{which}/lyalya_lyalya.ini configs/opt2_{which}/lyalya_qso.ini', main)
    else:
        item_name = 'lyalya_qso' if which == 'cross' else 'lyalya_lyalya'
        main = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text()
        main = re.sub(r'ini files = .*', f'ini files = configs/opt2_{which}/{item_name}.ini', main)
        text = (GOLDEN / 'configs' / 'joint' / f'{item_name}.ini').read_text()
        if which == 'cross':
            text = text.replace('[model]', '[model]\nrescale-coords-systematics = True\nold_growth_func = True\nfht_lowring = False')
        else:
            text = text.replace('[model]', '[model]\nUVB-shotnoise = True\nrescale-coords-systematics = True')
            main = main.replace('[parameters]', '[parameters]\nuv_shotnoise_amp = 0.02')
        (cfg / f'{item_name}.ini').write_text(text)
    (cfg / 'main.ini').write_text(main)
    return build_problem(f'configs/opt2_{which}/main.ini', search_dirs=[tmp_path, GOLDEN])


def mc_launcher_config(tmp_path, num_mocks=6, seed=3):
    """configs/mc/main.ini under ``tmp_path``: the auto-correlation on a FITS data file that carries distortion matrix
    and covariance, [control] run_montecarlo / num_mc_mocks / mc_seed, a [monte carlo] sampling table and
    [mc parameters] - what scripts/run_mc_sharded.py (the reference's bin/run_vega_mc_mpi.py) reads."""
    import re
    from vega_amd import synthetic
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'cf_lya-exp.npz')
    data_path = synthetic.write_data_file(tmp_path / 'cf_lya-synth.fits', source)
    cfg = tmp_path / 'configs' / 'mc'
    cfg.mkdir(parents=True)
    main = (GOLDEN / 'configs' / 'auto' / 'main.ini').read_text()
    main = re.sub(r'ini files = .*', 'ini files = configs/mc/lyalya_lyalya.ini', main)
    main = re.sub(r'\[control\][^\[]*', '', main)
    main = re.sub(r'\[sample\][^\[]*', '[sample]\nap = 0.5 1.5 1.0 0.01\nat = 0.5 1.5 1.0 0.01\n\n', main)
    main += (f'\n[control]\nrun_montecarlo = True\nnum_mc_mocks = {num_mocks}\nmc_seed = {seed}\n\n[monte carlo]\n'
             'ap = 0.5 1.5 1.0 0.01\nat = 0.5 1.5 1.0 0.01\nbias_eta_LYA = -1.0 0.0 -0.2 0.01\n\n[mc parameters]\nbeta_LYA = 1.8\n')
    main = re.sub(r'(\[output\]\nfilename = ).*', rf'\g<1>{tmp_path}/out/result', main)
    (cfg / 'main.ini').write_text(main)
    item = (GOLDEN / 'configs' / 'auto' / 'lyalya_lyalya.ini').read_text()
    (cfg / 'lyalya_lyalya.ini').write_text(re.sub(r'filename = .*', f'filename = {data_path}', item, count=1))
    return 'configs/mc/main.ini'


def fits_problem(tmp_path):
    """The fit scenario of tests/golden/make_golden.py::dump_fits: the auto-correlation with [sample] ap, at, bias_eta_LYA,
    beta_LYA, a [chi2 scan] over (ap, at), [control] / [monte carlo] / [mc parameters], the synthetic distortion matrix and
    covariance, and as data the reference's own model at the fixture's truth (`expected_fits.npz`: data/<name>)."""
    import re
    import numpy as np
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    cfg = tmp_path / 'configs' / 'fits'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (GOLDEN / 'configs' / 'auto' / 'main.ini').read_text()
    main = re.sub(r'ini files = .*', 'ini files = configs/fits/lyalya_lyalya.ini', main)
    main = re.sub(r'\[control\][^\[]*', '', main)
    main = re.sub(r'\[sample\][^\[]*', '[sample]\nap = 0.5 1.5 1.05 0.01\nat = 0.5 1.5 0.95 0.01\nbias_eta_LYA = True\n'
                  'beta_LYA = True\n\n', main)
    main += ('\n[chi2 scan]\nap = 0.99 1.07 3\nat = 0.93 0.97 2\n\n[control]\nrun_montecarlo = True\nmc_seed = 11\n\n'
             '[monte carlo]\nap = 0.5 1.5 1.05 0.01\nat = 0.5 1.5 0.95 0.01\n\n[mc parameters]\nbeta_LYA = 1.8\n')
    (cfg / 'main.ini').write_text(main)
    (cfg / 'lyalya_lyalya.ini').write_text((GOLDEN / 'configs' / 'auto' / 'lyalya_lyalya.ini').read_text())
    prob = build_problem('configs/fits/main.ini', search_dirs=[tmp_path, GOLDEN])
    exp = np.load(GOLDEN / 'expected_fits.npz')
    for name, item in prob.items.items():
        item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
        item.data_vec = np.array(exp[f'data/{name}'])
    return prob


def pytest_unconfigure(config):
    global _SPAWNER
    if _SPAWNER is not None:
        try:
            _SPAWNER.stdin.close()
            _SPAWNER.wait(timeout=10)
        except Exception:       # pragma: no cover
            _SPAWNER.kill()
        _SPAWNER = None
