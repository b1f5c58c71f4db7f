"""Generate the committed golden fixtures from the UNMODIFIED reference.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py

What it does
  1. converts the reference's test inputs (template P(k), data vectors + grids, metal grids,
     picca benchmark vectors) into ``tests/golden/inputs/*.npz`` bundles (data, not source);
  2. derives the test configs under ``tests/golden/configs/`` from the reference's own test
     configs, only rewriting file names to point at those bundles;
  3. imports the reference (``/root/reference/vega``) under the import stand-ins of
     ``tools/refshim`` (packages absent from this image; ``mcfit`` -> ``oracle/fftlog.py``) and
     dumps its outputs - models, chi2, log-likelihood and per-stage taps - for the fiducial
     point and for seeded walkers into ``tests/golden/expected_*.npz``.

The reference's ``allclose``-keyed grid caches are reset between walkers
so that every dumped value is an exact recomputation.
"""
import configparser
import io
import os
import re
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path('/root/reference')

sys.path[:0] = [str(REPO / 'tools' / 'refshim'), str(REPO), str(REF), str(REF / 'tests')]

from vega_amd import fitslite                       # noqa: E402
from vega_amd.tables import Table, write_bundle     # noqa: E402
from vega_amd import synthetic                      # noqa: E402

WALKER_SEED = 20260803
N_WALKERS = 8


# ----------------------------------------------------------------------------- inputs
def convert_inputs():
    out = HERE / 'inputs'
    out.mkdir(exist_ok=True)
    files = {
        'vega/models/PlanckDR16/PlanckDR16.fits': 'PlanckDR16.npz',
        'tests/data/cf_lya-exp.fits.gz': 'cf_lya-exp.npz',
        'tests/data/cf_lyb-exp.fits.gz': 'cf_lyb-exp.npz',
        'tests/data/xcf_lya-exp.fits.gz': 'xcf_lya-exp.npz',
        'tests/data/xcf_lyb-exp.fits.gz': 'xcf_lyb-exp.npz',
        'tests/data/metal_dmat_lya.fits.gz': 'metal_dmat_lya.npz',
        'tests/data/metal_dmat_lyb.fits.gz': 'metal_dmat_lyb.npz',
        'tests/data/metal_xdmat_lya.fits.gz': 'metal_xdmat_lya.npz',
        'tests/data/metal_xdmat_lyb.fits.gz': 'metal_xdmat_lyb.npz',
        'tests/data/dr16_simple_auto.fits': 'dr16_simple_auto.npz',
        'tests/data/dr16_simple_cross.fits': 'dr16_simple_cross.npz',
        'tests/data/picca_bench_data.fits': 'picca_bench_data.npz',
    }
    for src, dst in files.items():
        hdul = fitslite.open(REF / src)
        tabs = []
        for hdu in hdul[1:]:
            cols = {}
            for name in hdu.columns.names:
                col = hdu.data[name]
                # metal files carry bookkeeping columns the hot path never reads
                if col.dtype.kind in 'US':
                    continue
                cols[name] = col
            tabs.append(Table(hdu.header, cols))
        write_bundle(out / dst, tabs)
        print('wrote', out / dst)
    # HCD Voigt-profile table (data) used by model-hcd = fvoigt
    (out / 'fvoigt_models').mkdir(exist_ok=True)
    np.save(out / 'fvoigt_models' / 'Fvoigt_exp.npy', np.loadtxt(REF / 'vega/models/fvoigt_models/Fvoigt_exp.txt'))
    # DESI instrumental-systematics table (data)
    import shutil
    (out / 'instrumental_systematics').mkdir(exist_ok=True)
    shutil.copy(REF / 'vega/models/instrumental_systematics/desi-instrument-syst-for-forest-auto-correlation.csv',
                out / 'instrumental_systematics')


# ----------------------------------------------------------------------------- configs
_RENAMES = [
    (r'PlanckDR16/PlanckDR16\.fits', 'inputs/PlanckDR16.npz'),
    (r'data/([\w\-]+)\.fits(\.gz)?', r'inputs/\1.npz'),
]


def _rewrite(text, ini_dir_map):
    for pat, rep in _RENAMES:
        text = re.sub(pat, rep, text)
    for old, new in ini_dir_map.items():
        text = text.replace(old, new)
    return text


def derive_configs():
    cfg_out = HERE / 'configs'
    sets = {
        'full4': (REF / 'tests/full_configs', ['main.ini', 'lyalya_lyalya.ini', 'lyalya_lyalyb.ini',
                                                'lyalya_qso.ini', 'lyalyb_qso.ini'],
                  {'full_configs/': 'configs/full4/'}),
        'picca': (REF / 'examples/picca_benchmarks/configs/vega',
                  ['main.ini', 'main_cross.ini']
                  + [f'auto_test_{i}.ini' for i in (0, 1, 2, 4, 5, 6, 7)]
                  + [f'cross_test_{i}.ini' for i in (0, 1, 2, 4, 5, 6, 7)],
                  {'examples/picca_benchmarks/configs/vega/': 'configs/picca/'}),
    }
    for name, (src_dir, files, dirmap) in sets.items():
        dst = cfg_out / name
        dst.mkdir(parents=True, exist_ok=True)
        for f in files:
            text = _rewrite((src_dir / f).read_text(), dirmap)
            (dst / f).write_text(text)

    # joint auto+cross (BASELINE configs 3/4) and auto-only (configs 1/2) mains, with and
    # without the [metals] sections, derived from the full4 set
    main = (cfg_out / 'full4' / 'main.ini').read_text()
    for tag, items in {'joint': ['lyalya_lyalya', 'lyalya_qso'], 'auto': ['lyalya_lyalya']}.items():
        for metals in (True, False):
            sub = f'{tag}' + ('_metals' if metals else '')
            d = cfg_out / sub
            d.mkdir(exist_ok=True)
            line = 'ini files = ' + ' '.join(f'configs/{sub}/{it}.ini' for it in items)
            (d / 'main.ini').write_text(re.sub(r'ini files = .*', line, main))
            for it in items:
                text = (cfg_out / 'full4' / f'{it}.ini').read_text()
                if not metals:
                    text = re.sub(r'\[metals\][^\[]*', '', text)
                (d / f'{it}.ini').write_text(text)
    # BASELINE configs[1]: the auto-correlation alone with ell = 0, 2, 4
    d = cfg_out / 'auto_ell4'
    d.mkdir(exist_ok=True)
    (d / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/auto_ell4/lyalya_lyalya.ini', main))
    text = re.sub(r'\[metals\][^\[]*', '', (cfg_out / 'full4' / 'lyalya_lyalya.ini').read_text())
    (d / 'lyalya_lyalya.ini').write_text(text.replace('[model]', '[model]\nell_max = 4'))
    # auto-correlation with the UV shot-noise and DESI instrumental-systematics terms switched on
    d = cfg_out / 'auto_extras'
    d.mkdir(exist_ok=True)
    (d / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/auto_extras/lyalya_lyalya.ini', main)
                                .replace('[parameters]', '[parameters]\nuv_shotnoise_amp = 0.02\ndesi_inst_sys_amp = 0.0004'))
    text = re.sub(r'\[metals\][^\[]*', '', (cfg_out / 'full4' / 'lyalya_lyalya.ini').read_text())
    (d / 'lyalya_lyalya.ini').write_text(text.replace('[model]', '[model]\nUVB-shotnoise = True\n'
                                                      'desi-instrumental-systematics = True'))
    # auto-correlation with the mock-binning factor (power_spectrum.py:143-160), line-of-sight size scaled by the growth rate
    d = cfg_out / 'auto_mockbin'
    d.mkdir(exist_ok=True)
    (d / 'main.ini').write_text(re.sub(r'ini files = .*', 'ini files = configs/auto_mockbin/lyalya_lyalya.ini', main))
    text = re.sub(r'\[metals\][^\[]*', '', (cfg_out / 'full4' / 'lyalya_lyalya.ini').read_text())
    (d / 'lyalya_lyalya.ini').write_text(text.replace('[model]', '[model]\nmock-bin-size = 2.0\n'
                                                      'mock-los-smoothing = growth'))
    # joint + metals with the reference's `fast_metals` switch (metal x metal xi frozen at the first evaluation)
    d = cfg_out / 'joint_metals_fast'
    d.mkdir(exist_ok=True)
    items = ['lyalya_lyalya', 'lyalya_qso']
    line = 'ini files = ' + ' '.join(f'configs/joint_metals_fast/{it}.ini' for it in items)
    (d / 'main.ini').write_text(re.sub(r'ini files = .*', line, main))
    for it in items:
        text = (cfg_out / 'joint_metals' / f'{it}.ini').read_text()
        (d / f'{it}.ini').write_text(text.replace('[model]', '[model]\nfast_metals = True'))
    print('wrote configs under', cfg_out)


# ----------------------------------------------------------------------------- reference runs
def _reference():
    import matplotlib
    matplotlib.use('Agg')
    from vega import VegaInterface
    return VegaInterface


def _reset_caches(vega):
    """Defeat the allclose / first-call caches so each call recomputes (quirks 2, 3)."""
    for model in vega.models.values():
        pks = [model.Pk_core]
        if model.metals is not None:
            pks += list(model.metals.Pk_metal.values())
        for pk in pks:
            pk._arinyo_pars = None
            pk._peak_nl_pars = None
            pk._L0_hcd_cache = None
            pk._F_hcd = None


def _ref_main(tmp, items, metals):
    """Write a reference-side main.ini selecting ``items`` of tests/full_configs."""
    main = (REF / 'tests/full_configs/main.ini').read_text()
    paths = []
    for it in items:
        text = (REF / 'tests/full_configs' / f'{it}.ini').read_text()
        if not metals:
            text = re.sub(r'\[metals\][^\[]*', '', text)
        p = Path(tmp) / f'{it}.ini'
        p.write_text(text)
        paths.append(str(p))
    main = re.sub(r'ini files = .*', 'ini files = ' + ' '.join(paths), main)
    mp = Path(tmp) / 'main.ini'
    mp.write_text(main)
    return str(mp)


def make_walkers(params, n, seed=WALKER_SEED):
    """Seeded walkers around the fiducial: every parameter perturbed by 2 % (0.01 absolute for
    zero-valued ones); the huge ``qso_rad_lifetime`` sentinel is kept."""
    rng = np.random.default_rng(seed)
    names = sorted(params)
    out = []
    for _ in range(n):
        w = {}
        for name in names:
            v = params[name]
            g = rng.standard_normal()
            if name == 'qso_rad_lifetime':
                w[name] = v
            elif v == 0:
                w[name] = 0.01 * g
            else:
                w[name] = v * (1 + 0.02 * g)
        out.append(w)
    return names, out


def _taps_from_model(model, with_metals):
    """Stage outputs the reference keeps when ``save-components`` is on."""
    taps = {}
    for comp in ('peak', 'smooth'):
        taps[f'{comp}/pk_mean'] = np.mean(model.pk[comp]['core'])
        taps[f'{comp}/xi_core'] = model.xi[comp]['core']
        taps[f'{comp}/xi_distorted'] = model.xi_distorted[comp]['core']
        taps[f'{comp}/pk_ells'] = model.PktoXi.compute_pk_ells(model.pk[comp]['core'])
    return taps


def dump_full4(VegaInterface):
    os.chdir(REF / 'tests')
    vega = VegaInterface('full_configs/main.ini')
    out = {'log_lik': vega.log_lik(), 'chi2': vega.chi2(),
           'pinned_log_lik': -8766.997108462287}
    model = vega.compute_model(run_init=False)
    for name, xi in model.items():
        out[f'model/{name}'] = xi
    np.savez_compressed(HERE / 'expected_full4.npz', **out)
    print('full4: log_lik', out['log_lik'], 'chi2', out['chi2'])


def dump_subset(VegaInterface, tag, items, metals, synth=False):
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, items, metals))
        out = {}
        if synth:
            # synthetic distortion matrix + covariance (none ship with the reference)
            from scipy.sparse import csr_array
            for name in items:
                data = vega.data[name]
                grid = data.model_coordinates
                dm = synthetic.distortion_matrix(grid.rp_grid, grid.rt_grid)
                cov = synthetic.covariance(data.data_coordinates.rp_grid,
                                           data.data_coordinates.rt_grid)
                data._distortion_mat = csr_array(dm)
                data._cov_mat = cov
                data._inv_masked_cov = None
                data._log_cov_det = None
        names, walkers = make_walkers(vega.params, N_WALKERS)
        out['param_names'] = np.array(names)
        out['theta_fid'] = np.array([vega.params[n] for n in names])
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])

        _reset_caches(vega)
        out['fid/chi2'] = vega.chi2()
        out['fid/log_lik'] = vega.log_lik()
        model = vega.compute_model(run_init=False)
        for name in items:
            out[f'fid/model/{name}'] = model[name]

        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w))
            _reset_caches(vega)
            model = vega.compute_model(w, run_init=False)
            for name in items:
                out[f'walker{i}/model/{name}'] = model[name]
        out['chi2'] = np.array(chi2s)

        # stage taps at the fiducial point
        if not synth:
            vega.fiducial['save-components'] = True
            if metals:
                # components cannot be saved in fast-metal-bias mode (reference metals.py:67-69,242)
                for name in items:
                    out[f'fid/xi_metals/{name}'] = _metals_only(vega, name)
            with tempfile.TemporaryDirectory() as tmp2:
                vega_t = VegaInterface(_ref_main(tmp2, items, False))
                vega_t.fiducial['save-components'] = True
                vega_t.compute_model(run_init=True)
                for name in items:
                    for key, val in _taps_from_model(vega_t.models[name], False).items():
                        out[f'fid/taps/{name}/{key}'] = val
                    # FFTLog outputs per multipole for the smooth component
                    m = vega_t.models[name]
                    pk_ells = out[f'fid/taps/{name}/smooth/pk_ells']
                    for i, ell in enumerate(m.PktoXi.ell_vals):
                        r_fft, xi_fft = m.PktoXi.fftlog_objects[ell](pk_ells[i], extrap=False)
                        out[f'fid/taps/{name}/smooth/r_fft_{ell}'] = r_fft
                        out[f'fid/taps/{name}/smooth/xi_fft_{ell}'] = xi_fft
        np.savez_compressed(HERE / f'expected_{tag}.npz', **out)
        print(tag, 'fid chi2', out['fid/chi2'], 'walker chi2', out['chi2'][:3])


def _metals_only(vega, name):
    import copy
    pars = copy.deepcopy(vega.params)
    pars['peak'] = False
    _reset_caches(vega)
    return vega.models[name].metals.compute(pars, vega.fiducial['pk_full'], 'full')


def dump_picca(VegaInterface):
    os.chdir(REF)
    out = {}
    for kind, main in (('auto', 'main.ini'), ('cross', 'main_cross.ini')):
        vega = VegaInterface(f'examples/picca_benchmarks/configs/vega/{main}')
        vega.fiducial['Omega_de'] = None
        xi = vega.compute_model(run_init=True)
        for name, vec in xi.items():
            out[f'{kind}_{name}'] = vec
    np.savez_compressed(HERE / 'expected_picca.npz', **out)
    print('picca: dumped', len(out), 'vectors')


def dump_mc(VegaInterface):
    """Two Monte-Carlo mocks of the joint config with the synthetic covariance, drawn exactly as
    Analysis.run_monte_carlo does (np.random.seed once, then create_monte_carlo_sim per mock), plus the chi2 of
    the fiducial parameters against each mock through the reference's monte_carlo switch."""
    from scipy.sparse import csr_array
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya', 'lyalya_qso']
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, items, False))
        for name in items:
            data = vega.data[name]
            data._distortion_mat = csr_array(synthetic.distortion_matrix(data.model_coordinates.rp_grid,
                                                                          data.model_coordinates.rt_grid))
            data._cov_mat = synthetic.covariance(data.data_coordinates.rp_grid, data.data_coordinates.rt_grid)
            data._inv_masked_cov = None
            data._log_cov_det = None
        fid = vega.compute_model(run_init=False)
        out = {}
        np.random.seed(7)
        vega.monte_carlo = True
        for i in range(2):
            vega.analysis.create_monte_carlo_sim(fid, seed=None, scale=None)
            for name in items:
                out[f'mock{i}/{name}'] = vega.data[name].masked_mc_mock.copy()
            _reset_caches(vega)
            out[f'mock{i}/chi2_fid'] = vega.chi2()
        np.savez_compressed(HERE / 'expected_mc.npz', **out)
        print('mc: chi2 of the fiducial against the two mocks', out['mock0/chi2_fid'], out['mock1/chi2_fid'])


def dump_extras(VegaInterface):
    """Additive terms outside the default test configuration: UV-background shot noise and the DESI
    instrumental-systematics model, both switched on for the auto-correlation (fiducial point + one walker)."""
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nUVB-shotnoise = True\n'
                                                 'desi-instrumental-systematics = True'))
        mp = Path(main)
        mp.write_text(mp.read_text().replace('[parameters]', '[parameters]\nuv_shotnoise_amp = 0.02\n'
                                             'desi_inst_sys_amp = 0.0004'))
        vega = VegaInterface(main)
        out = {'fid/chi2': vega.chi2(), 'fid/model': vega.compute_model(run_init=False)['lyalya_lyalya']}
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 5)
        _reset_caches(vega)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[walkers[0][n] for n in names]])
        out['walker0/chi2'] = vega.chi2(walkers[0])
        _reset_caches(vega)
        out['walker0/model'] = vega.compute_model(walkers[0], run_init=False)['lyalya_lyalya']
        np.savez_compressed(HERE / 'expected_extras.npz', **out)
        print('extras: chi2', out['fid/chi2'], out['walker0/chi2'])


def dump_mockbin(VegaInterface):
    """`mock-bin-size` with `mock-los-smoothing = growth` (reference power_spectrum.py:143-160): fiducial point and
    one walker."""
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nmock-bin-size = 2.0\n'
                                                 'mock-los-smoothing = growth'))
        vega = VegaInterface(main)
        out = {'fid/chi2': vega.chi2(), 'fid/model': vega.compute_model(run_init=False)['lyalya_lyalya']}
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 6)
        walkers[0]['growth_rate'] = vega.params['growth_rate']      # template growth rate: not a free parameter
        _reset_caches(vega)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[walkers[0][n] for n in names]])
        out['walker0/chi2'] = vega.chi2(walkers[0])
        _reset_caches(vega)
        out['walker0/model'] = vega.compute_model(walkers[0], run_init=False)['lyalya_lyalya']
        np.savez_compressed(HERE / 'expected_mockbin.npz', **out)
        print('mockbin: chi2', out['fid/chi2'], out['walker0/chi2'])


def dump_mockbin_sampled(VegaInterface):
    """`mock-bin-size` whose line-of-sight size follows a parameter that VARIES between the walkers (reference
    power_spectrum.py:143-160, :494-501): `mock-los-smoothing = amplitude` with four values of `los_smooth_amp`, and
    `= growth` with four growth rates (which also enter the bias relations, utils.py:45-82)."""
    os.chdir(REF / 'tests')
    out = {}
    for mode, name in (('amplitude', 'los_smooth_amp'), ('growth', 'growth_rate')):
        with tempfile.TemporaryDirectory() as tmp:
            main = _ref_main(tmp, ['lyalya_lyalya'], False)
            item = Path(tmp) / 'lyalya_lyalya.ini'
            item.write_text(item.read_text().replace('[model]', '[model]\nmock-bin-size = 2.0\n'
                                                     f'mock-los-smoothing = {mode}'))
            mp = Path(main)
            if mode == 'amplitude':
                mp.write_text(mp.read_text().replace('[parameters]', '[parameters]\nlos_smooth_amp = 0.3'))
            vega = VegaInterface(main)
            names, walkers = make_walkers(vega.params, 4, seed=WALKER_SEED + 31)
            for i, w in enumerate(walkers):
                w['growth_rate'] = vega.params['growth_rate']
                w[name] = (0.05, 0.3, 0.7, 1.2)[i] if mode == 'amplitude' else vega.params['growth_rate'] * (0.85, 0.95, 1.05, 1.2)[i]
            out[f'{mode}/param_names'] = np.array(names)
            out[f'{mode}/theta'] = np.array([[w[n] for n in names] for w in walkers])
            chi2, models = [], []
            for w in walkers:
                _reset_caches(vega)
                chi2.append(vega.chi2(w))
                _reset_caches(vega)
                models.append(np.array(vega.compute_model(w, run_init=False)['lyalya_lyalya']))
            out[f'{mode}/chi2'] = np.array(chi2)
            out[f'{mode}/model'] = np.array(models)
            print('mockbin sampled', mode, chi2)
    np.savez_compressed(HERE / 'expected_mockbin_sampled.npz', **out)


def dump_fht_extrap(VegaInterface):
    """`fht_extrap = True` (reference vega/pktoxi.py:41,141): the FFTLog input padded with power laws.  The option only gives
    numbers for spectra whose last samples are not smoothed to zero: no small-scale non-linear term, no full-shape smoothing,
    no peak broadening (sigmaNL = 0), no binning kernel - with the test configuration's own model it returns NaN, recorded here as well.
    (The transform is the repo's restatement of mcfit behind tools/refshim: what this pins is the reference's call path.)"""
    os.chdir(REF / 'tests')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        text = item.read_text()
        item.write_text(text.replace('[model]', '[model]\nfht_extrap = True'))
        vega = VegaInterface(main)
        with np.errstate(all='ignore'):
            out['default_model/chi2'] = vega.chi2()
        text = re.sub(r'small scale nl *=.*\n', '', text)
        text = re.sub(r'fullshape smoothing *=.*\n', '', text)
        # (nor the binning kernel: its sincs change sign from sample to sample at the template's last wavenumbers, and a
        # power law through two such samples grows like |ratio|^617)
        text = text.replace('[model]', '[model]\nmodel binning = False')
        item.write_text(text.replace('[model]', '[model]\nfht_extrap = True'))
        vega = VegaInterface(main)
        names, walkers = make_walkers(vega.params, 3, seed=WALKER_SEED + 41)
        for w in walkers:
            w['growth_rate'] = vega.params['growth_rate']
            w['sigmaNL_par'] = 0.0
            w['sigmaNL_per'] = 0.0
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2, models = [], []
        for w in walkers:
            _reset_caches(vega)
            chi2.append(vega.chi2(w))
            _reset_caches(vega)
            models.append(np.array(vega.compute_model(w, run_init=False)['lyalya_lyalya']))
        out['chi2'] = np.array(chi2)
        out['model'] = np.array(models)
        # the same walkers with zero padding: how much the option moves the model
        item.write_text(text)
        vega = VegaInterface(main)
        _reset_caches(vega)
        out['zero_pad/chi2_0'] = vega.chi2(walkers[0])
    print('fht_extrap: chi2', out['chi2'], 'zero padding', out['zero_pad/chi2_0'], 'default model', out['default_model/chi2'])
    np.savez_compressed(HERE / 'expected_fht_extrap.npz', **out)


def dump_fits_ingest(VegaInterface):
    """Static-state ingestion (reference vega/data.py:285-473, utils.py:271-298): a data FITS file that carries the
    synthetic distortion matrix (`DM`) and covariance (`CO`) as vector columns - written by
    vega_amd.synthetic.write_data_file from the reference's own test file - read by the unmodified reference;
    chi2 / log-likelihood / model at the fiducial point and one walker."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
        data_path = synthetic.write_data_file(Path(tmp) / 'cf_lya-synth.fits', source)
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1))
        vega = VegaInterface(main)
        data = vega.data['lyalya_lyalya']
        assert data.has_distortion and data.cov_mat is not None
        out = {'fid/chi2': vega.chi2(), 'fid/log_lik': vega.log_lik(),
               'fid/model': vega.compute_model(run_init=False)['lyalya_lyalya'],
               'log_cov_det': data.log_cov_det, 'data_size': data.data_size}
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 3)
        _reset_caches(vega)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[walkers[0][n] for n in names]])
        out['walker0/chi2'] = vega.chi2(walkers[0])
        np.savez_compressed(HERE / 'expected_fits_ingest.npz', **out)
        print('fits ingest: chi2', out['fid/chi2'], 'log_lik', out['fid/log_lik'], out['walker0/chi2'])


def dump_blinding(VegaInterface):
    """Blinding (reference vega/data.py:305-339, vega_interface.py:389-421, :853-886, utils.py:375-393): a `desi_dr3`
    data file - the DA_BLIND column replaces DA - read by the unmodified reference, first without parameter
    offsets (the reference has no offsets file for that strategy), then with `_rnsps` set by hand to pin
    apply_blinding for model and priors alike."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
        data_path = synthetic.write_data_file(Path(tmp) / 'cf_lya-blind.fits', source, with_distortion=False,
                                              extra_header={'BLINDING': 'desi_dr3'},
                                              blind_data=synthetic.blinded_data_vector(source[0].data['DA']))
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        mp = Path(main)
        mp.write_text(mp.read_text() + '\n[priors]\nbeta_LYA = gaussian 1.6 0.1\n')
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1))
        vega = VegaInterface(main)
        assert vega._blind and vega._rnsps is None and vega.data['lyalya_lyalya'].blind
        out = {'plain/chi2': vega.chi2(), 'plain/log_lik': vega.log_lik()}
        vega._rnsps = synthetic.blinding_offsets()
        _reset_caches(vega)
        out['offsets/chi2'] = vega.chi2()
        out['offsets/log_lik'] = vega.log_lik()
        out['offsets/prior_chi2'] = vega.compute_prior_chi2()
        out['offsets/model'] = vega.compute_model(run_init=False)['lyalya_lyalya']
        names, walkers = make_walkers(vega.params, 2, seed=WALKER_SEED + 6)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        vals = []
        for w in walkers:
            _reset_caches(vega)
            vals.append(vega.chi2(w))
        out['offsets/walker_chi2'] = np.array(vals)
        # sampling a blinded parameter on such data is an error in the reference
        mp.write_text(mp.read_text().replace('[sample]', '[sample]\ngrowth_rate = True'))
        try:
            VegaInterface(main)
            out['sampled_blinded_error'] = np.array('')
        except ValueError as err:
            out['sampled_blinded_error'] = np.array(str(err))
        np.savez_compressed(HERE / 'expected_blinding.npz', **out)
        print('blinding:', {k: v for k, v in out.items() if np.ndim(v) == 0})


def dump_metal_decomp(VegaInterface):
    """`no-metal-decomp = False` (reference model.py:120-123, :181-186): the 15 metal pairs of the auto-correlation
    computed per component (smooth spectrum; peak spectrum with its broadening, times bao_amp) instead of once on
    the full spectrum.  Fiducial point and two walkers."""
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], True)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nno-metal-decomp = False'))
        vega = VegaInterface(main)
        assert not vega.models['lyalya_lyalya'].no_metal_decomp
        out = {'fid/chi2': vega.chi2(), 'fid/model': vega.compute_model(run_init=False)['lyalya_lyalya']}
        names, walkers = make_walkers(vega.params, 2, seed=WALKER_SEED + 7)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        vals = []
        for w in walkers:
            _reset_caches(vega)
            vals.append(vega.chi2(w))
        out['chi2'] = np.array(vals)
        _reset_caches(vega)
        out['walker0/model'] = vega.compute_model(walkers[0], run_init=False)['lyalya_lyalya']
        np.savez_compressed(HERE / 'expected_metal_decomp.npz', **out)
        print('metal decomp: chi2', out['fid/chi2'], out['chi2'])


NEW_METALS_CASES = {'auto': ('lyalya_lyalya', 'cf_lya-exp.fits.gz', False), 'cross': ('lyalya_qso', 'xcf_lya-exp.fits.gz', False),
                    'auto_rp': ('lyalya_lyalya', 'cf_lya-exp.fits.gz', True)}


def dump_new_metals(VegaInterface):
    """`new_metals = True` (reference vega/metals.py:83-112, :389-752): metal matrices built at set-up from a stacked-
    delta file / an object catalogue (synthetic, vega_amd.synthetic) instead of read from a picca file - full
    (rp, rt) matrices for an auto- and a cross-correlation, the rp-only form for the auto-correlation.  picca being
    absent, the reference runs on the restated wavelengths / comoving distance of tools/refshim/picca.  Dumps chi2 at
    the fiducial point and one walker, and per pair the effective coordinates and the matrix applied to a probe."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    out = {}
    for tag, (item_name, data_file, rp_only) in NEW_METALS_CASES.items():
        with tempfile.TemporaryDirectory() as tmp:
            source = read_tables(REF / 'tests/data' / data_file)
            data_path = synthetic.write_data_file(Path(tmp) / 'data.fits', source, with_distortion=False,
                                                  with_covariance=False, extra_header=synthetic.PICCA_COSMOLOGY_HEADER)
            stack = synthetic.write_stacked_deltas(Path(tmp) / 'stack.fits')
            cat = synthetic.write_object_catalog(Path(tmp) / 'cat.fits')
            main = _ref_main(tmp, [item_name], True)
            item = Path(tmp) / f'{item_name}.ini'
            text = re.sub(r'filename = .*', f'filename = {data_path}\nweights-tracer1 = {stack}\n'
                          f'weights-tracer2 = {cat if "qso" in item_name else stack}', item.read_text(), count=1)
            text = text.replace('[model]', '[model]\nnew_metals = True' + ('\nrp_only_metal_mats = True' if rp_only else ''))
            item.write_text(text + '\n' + synthetic.METAL_MATRIX_SECTION)
            vega = VegaInterface(main)
            metals = vega.models[item_name].metals
            assert metals.new_metals and metals.rp_only_metal_mats == rp_only
            out[f'{tag}/chi2'] = vega.chi2()
            out[f'{tag}/model'] = vega.compute_model(run_init=False)[item_name]
            names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 8)
            _reset_caches(vega)
            out[f'{tag}/param_names'] = np.array(names)
            out[f'{tag}/theta'] = np.array([[walkers[0][n] for n in names]])
            out[f'{tag}/walker0/chi2'] = vega.chi2(walkers[0])
            pairs = list(metals.rp_metal_dmats)
            out[f'{tag}/pairs'] = np.array(['|'.join(p_) for p_ in pairs])
            probe = np.cos(0.013 * np.arange(metals.size))
            for i, pair in enumerate(pairs):
                xi_obj = metals.Xi_metal[pair]
                out[f'{tag}/pair{i}/r'] = xi_obj._r
                out[f'{tag}/pair{i}/mu'] = xi_obj._mu
                out[f'{tag}/pair{i}/z'] = xi_obj._z
                out[f'{tag}/pair{i}/applied'] = metals.apply_metal_matrix(probe, pair)
            print('new_metals', tag, out[f'{tag}/chi2'], out[f'{tag}/walker0/chi2'], len(pairs), 'pairs')
    np.savez_compressed(HERE / 'expected_new_metals.npz', **out)


def dump_new_bias_evol(VegaInterface):
    """`new-bias-evolution = True` (reference correlation_func.py:238-299) on the cross-correlation, the cosmology
    taken from the data file's header (picca's D_H restated in tools/refshim/picca): fiducial point, one walker."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        source = read_tables(REF / 'tests/data/xcf_lya-exp.fits.gz')
        data_path = synthetic.write_data_file(Path(tmp) / 'xdata.fits', source, with_distortion=False,
                                              with_covariance=False, extra_header=synthetic.PICCA_COSMOLOGY_HEADER)
        main = _ref_main(tmp, ['lyalya_qso'], False)
        item = Path(tmp) / 'lyalya_qso.ini'
        text = re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1)
        item.write_text(text.replace('[model]', '[model]\nnew-bias-evolution = True'))
        vega = VegaInterface(main)
        assert vega.models['lyalya_qso'].Xi_core._use_new_bias_evol
        out = {'fid/chi2': vega.chi2(), 'fid/model': vega.compute_model(run_init=False)['lyalya_qso']}
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 9)
        _reset_caches(vega)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[walkers[0][n] for n in names]])
        out['walker0/chi2'] = vega.chi2(walkers[0])
        np.savez_compressed(HERE / 'expected_new_bias_evol.npz', **out)
        print('new bias evol: chi2', out['fid/chi2'], out['walker0/chi2'])


def dump_marginalization(VegaInterface):
    """Small-scale marginalisation (reference vega/correlation_item.py:175-268, vega/data.py:96-128, :762-828): the
    covariance of the FITS data file of dump_fits_ingest updated with the distorted templates of the bins at
    rt < 16 Mpc/h (and, second variant, of every bin the r-min cut removes, matched to data bins)."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    out = {}
    for tag, opts in (('rtmax', 'marginalize-below-rtmax = 16.0\nmarginalize-prior-sigma = 5.0'),
                      ('allrmin', 'marginalize-all-rmin-cuts = True\nmarginalize-match-data-bins = True'),
                      ('fitscales', 'marginalize-below-rtmax = 12.0\nfit-marginalized-scales = True\n'
                                    'marginalize-match-data-bins = True')):
        with tempfile.TemporaryDirectory() as tmp:
            source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
            data_path = synthetic.write_data_file(Path(tmp) / 'cf_lya-synth.fits', source)
            main = _ref_main(tmp, ['lyalya_lyalya'], False)
            item = Path(tmp) / 'lyalya_lyalya.ini'
            text = re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1)
            item.write_text(text.replace('[model]', '[model]\n' + opts))
            vega = VegaInterface(main)
            data = vega.data['lyalya_lyalya']
            out[f'{tag}/chi2'] = vega.chi2()
            out[f'{tag}/log_lik'] = vega.log_lik()
            out[f'{tag}/num_marg_modes'] = data.num_marg_modes
            out[f'{tag}/cov_update_trace'] = np.trace(data.cov_marg_update)
            # what the result file writes as <name>_VAR (vega/output.py:197): `Data.variance` is a live view of the covariance's
            # diagonal (data.py:85), so the update added in place at data.py:107 shows in it
            out[f'{tag}/variance'] = np.array(data.variance)
            print('marginalization', tag, out[f'{tag}/chi2'], out[f'{tag}/log_lik'], data.num_marg_modes)
            # the same templates fitted on the fly instead (control: marginalize-in-fit)
            mp = Path(main)
            mp.write_text(mp.read_text().replace('[control]', '[control]\nmarginalize-in-fit = True'))
            vega = VegaInterface(main)
            out[f'{tag}/infit/variance'] = np.array(vega.data['lyalya_lyalya'].variance)
            out[f'{tag}/infit/chi2'] = vega.chi2()
            out[f'{tag}/infit/log_lik'] = vega.log_lik()
            names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 4)
            _reset_caches(vega)
            out[f'{tag}/infit/param_names'] = np.array(names)
            out[f'{tag}/infit/theta'] = np.array([[walkers[0][n] for n in names]])
            out[f'{tag}/infit/walker0/chi2'] = vega.chi2(walkers[0])
            print('  in fit:', out[f'{tag}/infit/chi2'], out[f'{tag}/infit/log_lik'], out[f'{tag}/infit/walker0/chi2'])
    np.savez_compressed(HERE / 'expected_marginalization.npz', **out)


def dump_model_only(VegaInterface):
    """Correlations WITHOUT a data file (reference vega/correlation_item.py:40-42, :120-136, vega/vega_interface.py:110-137,
    :208-235): `has_datafile = False`, the caller hands the coordinates to every correlation item, `compute_model` builds
    the models on them - no distortion matrix, no mask, no chi2.  The auto-correlation on a 30 x 25 grid of 4 Mpc/h bins with a
    constant redshift (`Coordinates(..., z_eff=...)`) and the cross-correlation on a 40 x 20 grid with its own redshifts;
    fiducial point and three walkers."""
    from vega.coordinates import Coordinates
    os.chdir(REF / 'tests')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya', 'lyalya_qso'], False)
        for it in ('lyalya_lyalya', 'lyalya_qso'):
            item = Path(tmp) / f'{it}.ini'
            item.write_text(re.sub(r'filename = .*', 'has_datafile = False', item.read_text(), count=1))
        vega = VegaInterface(main)
        assert not vega._has_data and vega.data['lyalya_lyalya'] is None
        z_eff = vega.fiducial['z_eff']
        auto = Coordinates(0., 120., 100., 30, 25, z_eff=z_eff)
        rng = np.random.default_rng(77)
        cross = Coordinates(-80., 80., 80., 40, 20)
        cross = Coordinates(-80., 80., 80., 40, 20, z_grid=z_eff + 0.05 * rng.standard_normal(cross.rp_grid.size))
        vega.corr_items['lyalya_lyalya'].init_coordinates(auto)
        vega.corr_items['lyalya_qso'].init_coordinates(cross)
        out['cross/z'] = np.asarray(cross.z_grid)
        fid = vega.compute_model()
        for name, xi in fid.items():
            out[f'fid/{name}'] = np.asarray(xi)
        names, walkers = make_walkers(vega.params, 3, seed=WALKER_SEED + 21)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        for i, w in enumerate(walkers):
            for name, xi in vega.compute_model(w).items():
                out[f'walker{i}/{name}'] = np.asarray(xi)
        print('model only:', {n: (v.shape, float(np.abs(v).max())) for n, v in fid.items()})
    np.savez_compressed(HERE / 'expected_model_only.npz', **out)


def dump_dmat_file(VegaInterface):
    """The ingestion branch DESI production files use (reference vega/data.py:441-473): the distortion matrix in its own
    file with a model grid COEFMOD = 2 times finer than the data grid (DM is 2500 x 10000, HDU 2 carries the 100 x 100
    model-grid coordinates, the distorted-model grid is the regular 50 x 50 one) and the covariance in a `covariance-file`,
    both written by vega_amd.synthetic.write_dmat_file_case; fiducial point + 8 walkers through the unmodified reference,
    plus the small-scale marginalisation on that finer grid (templates of COEFMOD^2 model bins per distorted bin,
    vega/data.py:762-828)."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
        dmat, cov = synthetic.write_dmat_file_case(tmp, source, coef=2)
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        base = item.read_text().replace('[data]', f'[data]\ndistortion-file = {dmat}\ncovariance-file = {cov}', 1)
        item.write_text(base)
        vega = VegaInterface(main)
        data = vega.data['lyalya_lyalya']
        assert data.coeff_binning_model == 2 and data.distortion_mat.shape == (2500, 10000)
        assert data.model_coordinates.rp_grid.size == 10000 and data.dist_model_coordinates.rp_grid.size == 2500
        out['fid/chi2'] = vega.chi2()
        out['fid/log_lik'] = vega.log_lik()
        out['fid/model'] = vega.compute_model(run_init=False)['lyalya_lyalya']
        out['log_cov_det'] = data.log_cov_det
        names, walkers = make_walkers(vega.params, N_WALKERS, seed=WALKER_SEED + 11)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2, loglik, models = [], [], []
        for w in walkers:
            _reset_caches(vega)
            chi2.append(vega.chi2(w))
            _reset_caches(vega)
            loglik.append(vega.log_lik(w))
            _reset_caches(vega)
            models.append(vega.compute_model(w, run_init=False)['lyalya_lyalya'])
        out['walkers/chi2'] = np.array(chi2)
        out['walkers/log_lik'] = np.array(loglik)
        out['walkers/model'] = np.array(models)
        print('dmat file: chi2', out['fid/chi2'], 'log_lik', out['fid/log_lik'], out['walkers/chi2'][:3])
        item.write_text(base.replace('[model]', '[model]\nmarginalize-below-rtmax = 16.0\nmarginalize-prior-sigma = 5.0'))
        vega = VegaInterface(main)
        data = vega.data['lyalya_lyalya']
        out['marg/chi2'] = vega.chi2()
        out['marg/log_lik'] = vega.log_lik()
        out['marg/num_marg_modes'] = data.num_marg_modes
        out['marg/cov_update_trace'] = np.trace(data.cov_marg_update)
        _reset_caches(vega)
        out['marg/walker0/chi2'] = vega.chi2(walkers[0])
        print('  marginalised:', out['marg/chi2'], out['marg/log_lik'], data.num_marg_modes, out['marg/walker0/chi2'])
    np.savez_compressed(HERE / 'expected_dmat_file.npz', **out)


def dump_marg_mc(VegaInterface):
    """`marginalize-in-fit` together with a rescaled covariance in Monte Carlo (reference vega/vega_interface.py:282-292,
    :311-313; vega/data.py:711-722): the template coefficients come from the residual against the MOCK through the map built
    with the unscaled covariance, chi2 uses scaled_inv_masked_cov = C^-1 / scale.  One mock of scale 4 on the data file of
    dump_marginalization; chi2 at the fiducial point and one walker, with the coefficients."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
        data_path = synthetic.write_data_file(Path(tmp) / 'cf_lya-synth.fits', source)
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        text = re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1)
        item.write_text(text.replace('[model]', '[model]\nmarginalize-below-rtmax = 16.0\nmarginalize-prior-sigma = 5.0'))
        mp = Path(main)
        mp.write_text(mp.read_text().replace('[control]', '[control]\nmarginalize-in-fit = True'))
        vega = VegaInterface(main)
        assert vega.marginalize_in_fit
        fid = vega.compute_model(run_init=False)
        np.random.seed(23)
        vega.monte_carlo = True
        vega.analysis.create_monte_carlo_sim(fid, seed=None, scale=4.0)
        data = vega.data['lyalya_lyalya']
        out['scale'] = 4.0
        out['mock'] = data.masked_mc_mock.copy()
        np.testing.assert_allclose(data.scaled_inv_masked_cov, data.inv_masked_cov / 4.0)
        _reset_caches(vega)
        chi2, coeff = vega.chi2(return_marg_coeff=True)
        out['fid/chi2'], out['fid/coeff'] = chi2, coeff['lyalya_lyalya']
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 17)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[walkers[0][n] for n in names]])
        _reset_caches(vega)
        out['walker0/chi2'] = vega.chi2(walkers[0])
        print('marg + rescaled MC: chi2', out['fid/chi2'], out['walker0/chi2'], 'data size', data.data_size)
    np.savez_compressed(HERE / 'expected_marg_mc.npz', **out)


def direct_pk_vector(k, pk_full):
    """A stand-in for a Boltzmann-code spectrum: the fiducial one tilted and rescaled."""
    return 1.07 * pk_full * (k / 0.1)**0.03


def dump_direct_pk(VegaInterface):
    """`direct_pk` (reference vega_interface.py:208-248 -> model.py:188-207): chi2 and model from a caller-supplied
    full spectrum, joint fit with metals configured (which the direct path leaves out with the default
    no-metal-decomp) and a post-distortion additive broadband (which enters once instead of (1 + bao_amp) times)."""
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya', 'lyalya_qso']
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, items, True))
        pk = direct_pk_vector(vega.fiducial['k'], vega.fiducial['pk_full'])
        out = {'direct_pk': pk, 'fid/chi2': vega.chi2(direct_pk=pk)}
        model = vega.compute_model(run_init=False, direct_pk=pk)
        for name in items:
            out[f'fid/model/{name}'] = model[name]
        names, walkers = make_walkers(vega.params, 2, seed=WALKER_SEED + 8)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w, direct_pk=pk * (1 + 0.01 * (i + 1))))
        out['chi2'] = np.array(chi2s)
        np.savez_compressed(HERE / 'expected_direct_pk.npz', **out)
        print('direct_pk: chi2', out['fid/chi2'], out['chi2'])


def dump_direct_pk_metals(VegaInterface):
    """`direct_pk` together with `no-metal-decomp = False` (reference vega/model.py:188-207 -> :120-123): the metal terms are
    then part of the direct model, computed on the caller's spectrum (with the default decomposition they are left out:
    dump_direct_pk).  The auto-correlation with its 15 metal pairs; fiducial point (chi2, model) and two walkers."""
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], True)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nno-metal-decomp = False'))
        vega = VegaInterface(main)
        assert not vega.models['lyalya_lyalya'].no_metal_decomp
        pk = direct_pk_vector(vega.fiducial['k'], vega.fiducial['pk_full'])
        out = {'direct_pk': pk, 'fid/chi2': vega.chi2(direct_pk=pk),
               'fid/model': vega.compute_model(run_init=False, direct_pk=pk)['lyalya_lyalya'],
               'plain/chi2': vega.chi2()}
        names, walkers = make_walkers(vega.params, 2, seed=WALKER_SEED + 19)
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w, direct_pk=pk * (1 + 0.01 * (i + 1))))
        out['chi2'] = np.array(chi2s)
        np.savez_compressed(HERE / 'expected_direct_pk_metals.npz', **out)
        print('direct_pk + metals: chi2', out['fid/chi2'], out['chi2'], 'plain', out['plain/chi2'])


def dump_fast_metals(VegaInterface):
    """`fast_metals = True` (reference metals.py:53,144-169,280-282): metal x metal correlations are computed at
    the FIRST evaluation and reused for ever after.  Sequence dumped: chi2 at the fiducial point (fills the cache),
    then 4 walkers (chi2 and model each), all against that frozen cache."""
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya', 'lyalya_qso']
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, items, True)
        for it in items:
            p = Path(tmp) / f'{it}.ini'
            p.write_text(p.read_text().replace('[model]', '[model]\nfast_metals = True'))
        vega = VegaInterface(main)
        out = {'fid/chi2': vega.chi2()}
        model = vega.compute_model(run_init=False)
        for name in items:
            out[f'fid/model/{name}'] = model[name]
        names, walkers = make_walkers(vega.params, 4, seed=WALKER_SEED + 9)
        # the mode assumes that only the metal biases change between evaluations (metals.py:147-148): the
        # metal betas stay at their fiducial values, everything else is perturbed
        for w in walkers:
            for n in names:
                if n.startswith('beta_') and n not in ('beta_LYA', 'beta_QSO', 'beta_hcd'):
                    w[n] = vega.params[n]
        out['param_names'] = np.array(names)
        out['theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w))
            _reset_caches(vega)
            model = vega.compute_model(w, run_init=False)
            for name in items:
                out[f'walker{i}/model/{name}'] = model[name]
        out['chi2'] = np.array(chi2s)
        np.savez_compressed(HERE / 'expected_joint_metals_fast.npz', **out)
        print('fast_metals: fid chi2', out['fid/chi2'], 'walkers', out['chi2'])


def _inject_synthetic(vega, items):
    """The seeded distortion matrices and covariances of vega_amd.synthetic in place of the identity stand-ins the
    reference uses for its test files (vega/data.py:77-80)."""
    from scipy.sparse import csr_array
    for name in items:
        data = vega.data[name]
        data._distortion_mat = csr_array(synthetic.distortion_matrix(data.model_coordinates.rp_grid,
                                                                      data.model_coordinates.rt_grid))
        data._cov_mat = synthetic.covariance(data.data_coordinates.rp_grid, data.data_coordinates.rt_grid)
        data._inv_masked_cov = None
        data._log_cov_det = None


def dump_config1(VegaInterface):
    """BASELINE configs[1] as stated: Lya x Lya auto-correlation only, ell = 0, 2, 4 (`ell_max = 4`), dense synthetic
    2500^2 distortion matrix and covariance; chi2 / log-likelihood / model at the fiducial point and for 4 walkers,
    one evaluation at a time."""
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya']
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, items, False)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nell_max = 4'))
        vega = VegaInterface(main)
        _inject_synthetic(vega, items)
        assert vega.models['lyalya_lyalya'].PktoXi.ell_vals == (0, 2, 4)
        names, walkers = make_walkers(vega.params, 4, seed=WALKER_SEED + 21)
        out = {'param_names': np.array(names), 'theta': np.array([[w[n] for n in names] for w in walkers])}
        _reset_caches(vega)
        out['fid/chi2'] = vega.chi2()
        out['fid/log_lik'] = vega.log_lik()
        out['fid/model/lyalya_lyalya'] = vega.compute_model(run_init=False)['lyalya_lyalya']
        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w))
            _reset_caches(vega)
            out[f'walker{i}/model/lyalya_lyalya'] = vega.compute_model(w, run_init=False)['lyalya_lyalya']
        out['chi2'] = np.array(chi2s)
        np.savez_compressed(HERE / 'expected_config1.npz', **out)
        print('config1: fid chi2', out['fid/chi2'], 'walkers', out['chi2'])


def dump_marg_coeff(VegaInterface):
    """The marginalisation coefficients through the public surface (reference vega/vega_interface.py:208-325,
    :546-579; the PolyChord adapter calls log_lik(..., return_marg_coeff=True), vega/samplers/polychord.py:106-113):
    `chi2 / log_lik(return_marg_coeff=True)` and `compute_model(marg_coeff=...)` on the data file of
    dump_marginalization, templates folded into the covariance and fitted on the fly, at the fiducial point, for one
    walker, and for a walker whose model cannot be evaluated (the first coefficients ever computed come back)."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    out = {}
    opts = 'marginalize-below-rtmax = 16.0\nmarginalize-prior-sigma = 5.0'
    for mode in ('cov', 'infit'):
        with tempfile.TemporaryDirectory() as tmp:
            source = read_tables(REF / 'tests/data/cf_lya-exp.fits.gz')
            data_path = synthetic.write_data_file(Path(tmp) / 'cf_lya-synth.fits', source)
            main = _ref_main(tmp, ['lyalya_lyalya'], False)
            item = Path(tmp) / 'lyalya_lyalya.ini'
            text = re.sub(r'filename = .*', f'filename = {data_path}', item.read_text(), count=1)
            item.write_text(text.replace('[model]', '[model]\n' + opts))
            if mode == 'infit':
                mp = Path(main)
                mp.write_text(mp.read_text().replace('[control]', '[control]\nmarginalize-in-fit = True'))
            vega = VegaInterface(main)
            chi2, coeff = vega.chi2(return_marg_coeff=True)
            out[f'{mode}/fid/chi2'] = chi2
            out[f'{mode}/fid/coeff'] = coeff['lyalya_lyalya']
            ll, flat = vega.log_lik(return_marg_coeff=True)
            out[f'{mode}/fid/log_lik'] = ll
            out[f'{mode}/fid/coeff_flat'] = flat
            out[f'{mode}/fid/model_plain'] = vega.compute_model(run_init=False)['lyalya_lyalya']
            out[f'{mode}/fid/model_with_templates'] = vega.compute_model(run_init=False, marg_coeff=coeff)['lyalya_lyalya']
            names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 4)
            _reset_caches(vega)
            out[f'{mode}/param_names'] = np.array(names)
            out[f'{mode}/theta'] = np.array([[walkers[0][n] for n in names]])
            chi2, coeff = vega.chi2(walkers[0], return_marg_coeff=True)
            out[f'{mode}/walker0/chi2'] = chi2
            out[f'{mode}/walker0/coeff'] = coeff['lyalya_lyalya']
            # a model error (rescaled separations beyond the FFTLog range): chi2 = 1e100 and the first coefficients
            _reset_caches(vega)
            bad_chi2, bad_coeff = vega.chi2({'ap': 1e3}, return_marg_coeff=True)
            out[f'{mode}/bad/chi2'] = bad_chi2
            out[f'{mode}/bad/coeff'] = bad_coeff['lyalya_lyalya']
            print('marg coeff', mode, out[f'{mode}/fid/chi2'], out[f'{mode}/fid/coeff'][:3], bad_chi2)
    np.savez_compressed(HERE / 'expected_marg_coeff.npz', **out)


def dump_global_mc(VegaInterface):
    """Monte-Carlo mocks from a global covariance (reference vega/analysis.py:164-222, used at :270-276; chi2 reads
    analysis.current_mc_mock, vega/vega_interface.py:294-304): the joint config with the synthetic distortion matrices
    and a `global-cov-file` written by vega_amd.synthetic; two mocks drawn as Analysis.run_monte_carlo draws them
    (np.random.seed once, then create_global_monte_carlo per mock) and the chi2 of the fiducial parameters against
    each; plus chi2 / log-likelihood against the data with that covariance."""
    from vega_amd.tables import read_tables
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya', 'lyalya_qso']
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, items, False)
        grids = []
        for it, f in zip(items, ('cf_lya-exp', 'xcf_lya-exp')):
            t = read_tables(REF / f'tests/data/{f}.fits.gz')[0].data
            grids.append((np.asarray(t['RP'], dtype=float), np.asarray(t['RT'], dtype=float)))
        gc_path = synthetic.write_global_covariance(Path(tmp) / 'global_cov.fits', synthetic.global_covariance(grids))
        mp = Path(main)
        mp.write_text(mp.read_text().replace('[data sets]', f'[data sets]\nglobal-cov-file = {gc_path}'))
        vega = VegaInterface(main)
        assert vega._use_global_cov
        _inject_synthetic(vega, items)
        out = {'data/chi2': vega.chi2(), 'data/log_lik': vega.log_lik()}
        fid = vega.compute_model(run_init=False)
        np.random.seed(7)
        vega.monte_carlo = True
        for i in range(2):
            mock = vega.analysis.create_global_monte_carlo(fid, seed=None, scale=None)
            out[f'mock{i}/global'] = mock.copy()
            _reset_caches(vega)
            out[f'mock{i}/chi2_fid'] = vega.chi2()
            out[f'mock{i}/log_lik_fid'] = vega.log_lik()
        # a rescaled covariance: a fresh interface (the Cholesky factor is cached with its first scale)
        vega = VegaInterface(main)
        _inject_synthetic(vega, items)
        fid = vega.compute_model(run_init=False)
        np.random.seed(7)
        out['scaled/mock0/global'] = vega.analysis.create_global_monte_carlo(fid, seed=None, scale=0.25).copy()
        np.savez_compressed(HERE / 'expected_global_mc.npz', **out)
        print('global mc: data chi2', out['data/chi2'], 'mocks', out['mock0/chi2_fid'], out['mock1/chi2_fid'])


def dump_model_compute(VegaInterface):
    """`vega.models[name].compute(pars, pk_full, pk_smooth)` (reference vega/model.py:157-187) - the per-correlation
    entry point below compute_model - with the fiducial spectra and with caller-supplied ones (the stand-in of
    direct_pk_vector applied to both), joint config with metals."""
    import copy
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya', 'lyalya_qso']
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, items, True))
        k = vega.fiducial['k']
        pk_full = direct_pk_vector(k, vega.fiducial['pk_full'])
        pk_smooth = direct_pk_vector(k, vega.fiducial['pk_smooth'])
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 31)
        out = {'pk_full': pk_full, 'pk_smooth': pk_smooth, 'param_names': np.array(names),
               'theta': np.array([[walkers[0][n] for n in names]])}
        for name in items:
            _reset_caches(vega)
            pars = copy.deepcopy(vega.params)
            out[f'fiducial_spectra/{name}'] = vega.models[name].compute(pars, vega.fiducial['pk_full'],
                                                                        vega.fiducial['pk_smooth'])
            _reset_caches(vega)
            pars = copy.deepcopy(walkers[0])
            out[f'own_spectra/{name}'] = vega.models[name].compute(pars, pk_full, pk_smooth)
        np.savez_compressed(HERE / 'expected_model_compute.npz', **out)
        print('model compute: dumped', [k for k in out if '/' in k])


def dump_options2(VegaInterface):
    """Model options of the xi stage that round 1 rejected: `rescale-coords-systematics` (QSO radiation and UV shot
    noise evaluated on the rescaled coordinates, reference vega/correlation_func.py:470-475, :681-684),
    `old_growth_func` (:75-80, :405-444), `fht_lowring = False` (vega/pktoxi.py:42,53), and `model_pk` (the models are
    the multipoles of the core power spectrum: vega/vega_interface.py:66, vega/model.py:106-107)."""
    os.chdir(REF / 'tests')
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_qso'], False)
        item = Path(tmp) / 'lyalya_qso.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nrescale-coords-systematics = True\n'
                                                 'old_growth_func = True\nfht_lowring = False'))
        vega = VegaInterface(main)
        out['cross/fid/chi2'] = vega.chi2()
        out['cross/fid/model'] = vega.compute_model(run_init=False)['lyalya_qso']
        names, walkers = make_walkers(vega.params, 2, seed=WALKER_SEED + 41)
        out['cross/param_names'] = np.array(names)
        out['cross/theta'] = np.array([[w[n] for n in names] for w in walkers])
        chi2s = []
        for i, w in enumerate(walkers):
            _reset_caches(vega)
            chi2s.append(vega.chi2(w))
            _reset_caches(vega)
            out[f'cross/walker{i}/model'] = vega.compute_model(w, run_init=False)['lyalya_qso']
        out['cross/chi2'] = np.array(chi2s)
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya'], False)
        mp = Path(main)
        mp.write_text(mp.read_text().replace('[parameters]', '[parameters]\nuv_shotnoise_amp = 0.02'))
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nUVB-shotnoise = True\n'
                                                 'rescale-coords-systematics = True'))
        vega = VegaInterface(main)
        out['auto/fid/chi2'] = vega.chi2()
        out['auto/fid/model'] = vega.compute_model(run_init=False)['lyalya_lyalya']
        w = {'ap': 1.03, 'at': 0.96, 'uv_shotnoise_amp': 0.03}
        _reset_caches(vega)
        out['auto/walker/chi2'] = vega.chi2(w)
    with tempfile.TemporaryDirectory() as tmp:
        main = _ref_main(tmp, ['lyalya_lyalya', 'lyalya_qso'], False)
        mp = Path(main)
        mp.write_text(mp.read_text().replace('[control]', '[control]\nmodel_pk = True'))
        vega = VegaInterface(main)
        model = vega.compute_model(run_init=False)
        for name in model:
            out[f'model_pk/fid/{name}'] = np.asarray(model[name])
        w = {'ap': 1.03, 'bias_eta_LYA': -0.21, 'beta_LYA': 1.5, 'bao_amp': 0.8, 'sigmaNL_par': 7.0}
        _reset_caches(vega)
        model = vega.compute_model(w, run_init=False)
        for name in model:
            out[f'model_pk/walker/{name}'] = np.asarray(model[name])
    np.savez_compressed(HERE / 'expected_options2.npz', **out)
    print('options2: cross chi2', out['cross/fid/chi2'], out['cross/chi2'], 'auto', out['auto/fid/chi2'],
          'model_pk shape', out['model_pk/fid/lyalya_lyalya'].shape)


FITS_SAMPLE = ('[sample]\nap = 0.5 1.5 1.05 0.01\nat = 0.5 1.5 0.95 0.01\nbias_eta_LYA = True\nbeta_LYA = True\n\n')
FITS_SCAN = '[chi2 scan]\nap = 0.99 1.07 3\nat = 0.93 0.97 2\n\n'
FITS_MC = ('[control]\nrun_montecarlo = True\nmc_seed = 11\n\n[monte carlo]\nap = 0.5 1.5 1.05 0.01\nat = 0.5 1.5 0.95 0.01\n\n'
           '[mc parameters]\nbeta_LYA = 1.8\n\n')


FITS_TRUTH = {'ap': 1.03, 'at': 0.97, 'bias_eta_LYA': -0.21, 'beta_LYA': 1.6}


def _fit_scenario(vega, items):
    """Synthetic distortion matrix + covariance, and as DATA the reference's own model at FITS_TRUTH (the test file's data
    vector is a noiseless model without distortion - against the synthetic matrices a fit would run into the limits)."""
    _inject_synthetic(vega, items)
    truth = vega.compute_model(dict(FITS_TRUTH), run_init=False)
    for name in items:
        data = vega.data[name]
        vec = np.array(truth[name])
        mask = data.dist_model_coordinates.get_mask_to_other(data.data_coordinates)
        data._data_vec = vec[mask] if vec.size != data.data_vec.size else vec
        data._masked_data_vec = None
    _reset_caches(vega)
    return {name: np.array(vega.data[name].data_vec) for name in items}


def dump_fits(VegaInterface):
    """The reference's fit DRIVERS on the auto-correlation (no metals): `minimize` (bias pre-fit + full fit,
    vega/minimizer.py:39-103), `Analysis.chi2_scan` (vega/analysis.py:53-122), `initialize_monte_carlo`
    (vega/vega_interface.py:505-544) and `Analysis.run_monte_carlo` (vega/analysis.py:224-308) - walked by the unmodified
    reference with the repo's MIGRAD restatement behind the iminuit surface (tools/refshim/iminuit).  What these fixtures
    pin is everything AROUND the minimiser: which parameters are pinned at which grid values, start values, the order
    of the scan, seeding and scaling of the mocks, which results are kept."""
    os.chdir(REF / 'tests')
    items = ['lyalya_lyalya']
    with tempfile.TemporaryDirectory() as tmp:
        main_path = Path(_ref_main(tmp, items, False))
        main = main_path.read_text()
        main = re.sub(r'\[sample\][^\[]*', FITS_SAMPLE, main)
        main = re.sub(r'\[control\][^\[]*', '', main)
        main_path.write_text(main + '\n' + FITS_SCAN + FITS_MC)
        vega = VegaInterface(str(main_path))
        data_vecs = _fit_scenario(vega, items)
        out = {'sample_names': np.array(list(vega.sample_params['limits']))}
        for name in items:
            out[f'data/{name}'] = data_vecs[name]
        vega.minimize()
        m = vega.minimizer
        out['fit/names'] = np.array(list(m.values))
        out['fit/values'] = np.array(list(m.values.values()))
        out['fit/errors'] = np.array([m.errors[n] for n in m.values])
        out['fit/covariance'] = np.array(m.covariance)
        out['fit/fval'] = m.fmin.fval
        out['fit/nfcn'] = m.fmin.nfcn
        print('fits: minimize', dict(m.values), m.fmin)
        scan = vega.analysis.chi2_scan()
        out['scan/grid_names'] = np.array(list(vega.analysis.grids))
        for k, g in vega.analysis.grids.items():
            out[f'scan/grid/{k}'] = g
        keys = list(scan[0])
        out['scan/keys'] = np.array(keys)
        out['scan/results'] = np.array([[r[k] for k in keys] for r in scan])
        print('fits: scan', out['scan/results'])
        # Monte Carlo: fiducial from a fit to the data + [mc parameters], one mock per correlation installed as data
        vega2 = VegaInterface(str(main_path))
        _fit_scenario(vega2, items)
        mocks = vega2.initialize_monte_carlo()
        for name, mock in mocks.items():
            out[f'mcinit/mock/{name}'] = np.array(mock)
        _reset_caches(vega2)
        out['mcinit/chi2'] = vega2.chi2()
        out['mcinit/log_lik'] = vega2.log_lik()
        vega3 = VegaInterface(str(main_path))
        _fit_scenario(vega3, items)
        fid = vega3.get_fiducial_for_monte_carlo()
        vega3.monte_carlo = True                # (bin/run_vega_mc_mpi.py:44)
        for name in items:
            out[f'mc/fiducial/{name}'] = np.array(fid[name])
        vega3.analysis.run_monte_carlo(fid, num_mocks=2, seed=5)
        an = vega3.analysis
        out['mc/names'] = np.array(list(an.mc_bestfits))
        out['mc/bestfits'] = np.array([an.mc_bestfits[n] for n in an.mc_bestfits])
        out['mc/chisq'] = np.array(an.mc_chisq)
        out['mc/valid'] = np.array(an.mc_valid_minima)
        for name in items:
            out[f'mc/mocks/{name}'] = np.array(an.mc_mocks[name])
        np.savez_compressed(HERE / 'expected_fits.npz', **out)
        print('fits: mc bestfits', out['mc/bestfits'], out['mc/chisq'])



def dump_fit_stats(VegaInterface):
    """What the reference's `minimize` leaves next to the fit (vega/vega_interface.py:593-643) on its own test configuration
    (tests/full_configs/main.ini, the fit tests/test_vega.py:16-18 pins): the best-fit model, every correlation's masked
    size / chi2 / reduced chi2 / p-value, the totals - the numbers `Output.write_results` puts into the MODEL_<name> headers."""
    os.chdir(REF / 'tests')
    vega = VegaInterface('full_configs/main.ini')
    vega.minimize()
    out = {'names': np.array(list(vega.corr_items)), 'fval': vega.minimizer.fmin.fval,
           'chisq': vega.chisq, 'reduced_chisq': vega.reduced_chisq, 'p_value': vega.p_value,
           'total_data_size': vega.total_data_size,
           'fit/names': np.array(list(vega.minimizer.values)), 'fit/values': np.array(list(vega.minimizer.values.values()))}
    for name, st in vega.bestfit_corr_stats.items():
        for key in ('masked_size', 'chisq', 'reduced_chisq', 'p_value'):
            out[f'stats/{name}/{key}'] = st[key]
        assert st['bestfit_marg_coeff'] is None
        out[f'model/{name}'] = np.array(vega.bestfit_model[name])
    np.savez_compressed(HERE / 'expected_fit_stats.npz', **out)
    print('fit stats:', {k: out[k] for k in ('fval', 'chisq', 'reduced_chisq', 'p_value', 'total_data_size')},
          {n: vega.bestfit_corr_stats[n]['chisq'] for n in vega.bestfit_corr_stats})


SENSITIVITY_NOMINAL = {'bias_eta_LYA': (-0.2, 0.004), 'ap': (1.04, 0.02), 'sigmaNL_par': (6.4, 0.5), 'drp_QSO': (0.2, 0.1)}


def dump_sensitivity(VegaInterface):
    """`VegaInterface.compute_sensitivity` (vega/vega_interface.py:956-1075) of the unmodified reference on its own test
    configuration: the partial derivatives of the four parts of every correlation (final and raw, peak and smooth) and the
    Fisher information per bin of every parameter pair, at SENSITIVITY_NOMINAL."""
    os.chdir(REF / 'tests')
    # (the auto + cross items WITHOUT their metal terms = tests/golden/configs/joint: with metals the reference's own function
    # stops at `assert not fast_metals` in Metals.compute_metal_corr_slow, vega/metals.py:242, unless fast_metal_bias is off)
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, ['lyalya_lyalya', 'lyalya_qso'], False))
        vega.compute_sensitivity(nominal=dict(SENSITIVITY_NOMINAL), frac=0.1, verbose=False)
    sens = vega.sensitivity
    out = {'names': np.array(list(vega.corr_items)), 'params': np.array(list(SENSITIVITY_NOMINAL)),
           'nominal': np.array([SENSITIVITY_NOMINAL[p] for p in SENSITIVITY_NOMINAL])}
    for name in vega.corr_items:
        for pname, arr in sens['partials'][name].items():
            out[f'partials/{name}/{pname}'] = arr
        for (p1, p2), arr in sens['fisher'][name].items():
            if 'ap' in (p1, p2):            # (the others follow from the partials the same way: kept out of the fixture for its size)
                out[f'fisher/{name}/{p1}/{p2}'] = arr
        out[f'fisher_keys/{name}'] = np.array(['/'.join(k) for k in sens['fisher'][name]])
    np.savez_compressed(HERE / 'expected_sensitivity.npz', **out)
    print('sensitivity:', {k: float(np.nansum(v)) for k, v in out.items() if k.startswith('fisher/lyalya_lyalya/')})


def dump_components(VegaInterface):
    """The saved components of the reference's models (`save-components`, vega/model.py:41-45, :113-115, :151-153) on the
    auto + cross items without metal terms, at one walker: the raw core correlations of the peak / smooth spectrum and the
    components' final (distorted) models - what `write_cf` puts into the Xi_<name> HDUs (vega/output.py:375-440)."""
    os.chdir(REF / 'tests')
    with tempfile.TemporaryDirectory() as tmp:
        vega = VegaInterface(_ref_main(tmp, ['lyalya_lyalya', 'lyalya_qso'], False))
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 29)
        vega.fiducial['save-components'] = True
        full = vega.compute_model(walkers[0], run_init=True)
        out = {'names': np.array(list(vega.corr_items)), 'param_names': np.array(names),
               'theta': np.array([[walkers[0][n] for n in names]])}
        for name, model in vega.models.items():
            out[f'model/{name}'] = np.array(full[name])
            for part in ('peak', 'smooth'):
                out[f'xi/{name}/{part}'] = np.array(model.xi[part]['core'])
                out[f'xi_distorted/{name}/{part}'] = np.array(model.xi_distorted[part]['core'])
        # ... and the auto-correlation WITH its metal terms (tests/golden/configs/auto_metals): the reference saves components
        # only with `fast_metal_bias = False` (vega/metals.py:242); with the default `no-metal-decomp = True` the model keeps
        # the 'core' entries alone, the metal terms sit inside the smooth component's final model (vega/model.py:117-119)
        main = _ref_main(tmp, ['lyalya_lyalya'], True)
        item = Path(tmp) / 'lyalya_lyalya.ini'
        item.write_text(item.read_text().replace('[model]', '[model]\nfast_metal_bias = False'))
        vega = VegaInterface(main)
        names, walkers = make_walkers(vega.params, 1, seed=WALKER_SEED + 31)
        vega.fiducial['save-components'] = True
        full = vega.compute_model(walkers[0], run_init=True)
        out['metals/param_names'] = np.array(names)
        out['metals/theta'] = np.array([[walkers[0][n] for n in names]])
        model = vega.models['lyalya_lyalya']
        assert list(model.xi['smooth']) == ['core']
        out['metals/model'] = np.array(full['lyalya_lyalya'])
        for part in ('peak', 'smooth'):
            out[f'metals/xi/{part}'] = np.array(model.xi[part]['core'])
            out[f'metals/xi_distorted/{part}'] = np.array(model.xi_distorted[part]['core'])
    np.savez_compressed(HERE / 'expected_components.npz', **out)
    print('components:', {k: float(np.abs(v).max()) for k, v in out.items() if 'xi' in k})


def dump_pk_kat(VegaInterface):
    """A few full P(k,mu) grids reduced to the checksums reference tests/test_pk.py uses."""
    # The known answers themselves are constants of the reference's test and live in
    # tests/test_oracle.py; nothing to generate here.


if __name__ == '__main__':
    what = sys.argv[1:] or ['inputs', 'configs', 'full4', 'joint', 'picca', 'mc', 'extras', 'fast_metals', 'mockbin', 'fits_ingest', 'marginalization', 'direct_pk', 'blinding', 'metal_decomp', 'new_metals', 'new_bias_evol', 'config1', 'marg_coeff', 'global_mc', 'model_compute', 'options2', 'fits', 'dmat_file', 'marg_mc', 'direct_pk_metals', 'mockbin_sampled', 'fht_extrap', 'fit_stats', 'sensitivity', 'components', 'model_only']
    if 'inputs' in what:
        convert_inputs()
    if 'configs' in what:
        derive_configs()
    VI = _reference() if set(what) & {'full4', 'joint', 'picca', 'mc', 'extras', 'fast_metals', 'mockbin', 'fits_ingest', 'marginalization', 'direct_pk', 'blinding', 'metal_decomp', 'new_metals', 'new_bias_evol', 'config1', 'marg_coeff', 'global_mc', 'model_compute', 'options2', 'fits', 'dmat_file', 'marg_mc', 'direct_pk_metals', 'mockbin_sampled', 'fht_extrap', 'fit_stats', 'sensitivity', 'components', 'model_only'} else None
    if 'full4' in what:
        dump_full4(VI)
    if 'joint' in what:
        dump_subset(VI, 'joint_metals', ['lyalya_lyalya', 'lyalya_qso'], True)
        dump_subset(VI, 'joint', ['lyalya_lyalya', 'lyalya_qso'], False)
        dump_subset(VI, 'joint_synth', ['lyalya_lyalya', 'lyalya_qso'], False, synth=True)
    if 'picca' in what:
        dump_picca(VI)
    if 'mc' in what:
        dump_mc(VI)
    if 'extras' in what:
        dump_extras(VI)
    if 'fast_metals' in what:
        dump_fast_metals(VI)
    if 'mockbin' in what:
        dump_mockbin(VI)
    if 'fits_ingest' in what:
        dump_fits_ingest(VI)
    if 'marginalization' in what:
        dump_marginalization(VI)
    if 'direct_pk' in what:
        dump_direct_pk(VI)
    if 'blinding' in what:
        dump_blinding(VI)
    if 'metal_decomp' in what:
        dump_metal_decomp(VI)
    if 'new_metals' in what:
        dump_new_metals(VI)
    if 'new_bias_evol' in what:
        dump_new_bias_evol(VI)
    if 'config1' in what:
        dump_config1(VI)
    if 'marg_coeff' in what:
        dump_marg_coeff(VI)
    if 'global_mc' in what:
        dump_global_mc(VI)
    if 'model_compute' in what:
        dump_model_compute(VI)
    if 'options2' in what:
        dump_options2(VI)
    if 'fits' in what:
        dump_fits(VI)
    if 'dmat_file' in what:
        dump_dmat_file(VI)
    if 'marg_mc' in what:
        dump_marg_mc(VI)
    if 'direct_pk_metals' in what:
        dump_direct_pk_metals(VI)
    if 'mockbin_sampled' in what:
        dump_mockbin_sampled(VI)
    if 'fht_extrap' in what:
        dump_fht_extrap(VI)
    if 'fit_stats' in what:
        dump_fit_stats(VI)
    if 'sensitivity' in what:
        dump_sensitivity(VI)
    if 'components' in what:
        dump_components(VI)
    if 'model_only' in what:
        dump_model_only(VI)
