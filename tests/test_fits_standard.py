"""Every FITS file this package writes, checked by a validator written from the FITS standard (tests/fits_standard.py) and not
from the package's own reader - 2880-byte blocks, 80-column cards, the mandatory keywords in their order and fixed format,
TFORMn / TDIMn / NAXIS1 consistency, HIERARCH cards, big-endian payload, zero fill - and decoded independently: the columns the
validator reads are the arrays that were written, and what `fitslite` reads.  The reverse too, in the build container: the
package's reader on the files astropy wrote for the reference (tests/data, vega/models) against the validator's own decoding of
them and against the committed .npz bundles.  Layouts being matched: reference vega/output.py:37-349, :442-520,
vega/data.py:285-473."""
import gzip
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN, load_problem
from fits_standard import FitsError, check_file

REFERENCE = Path('/root/reference')


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in 'US' or b.dtype.kind in 'US':
        return [str(x).strip() for x in a.ravel()] == [str(x).strip() for x in b.ravel()]
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == 'f')


def _against_fitslite(path):
    """The validator's decoding of every table of `path` equals what fitslite hands out (values, names, header values)."""
    from vega_amd import fitslite
    std = check_file(path)
    lite = fitslite.open(path)
    assert len(std) == len(lite)
    for s, l in zip(std, lite):
        for key, value in s['header'].items():
            if key in ('COMMENT', 'HISTORY') or key not in l.header:
                continue
            got = l.header[key]
            assert got == value or (isinstance(value, float) and got == pytest.approx(value, rel=1e-15)), (path, key, value, got)
        if s['columns'] is None:
            continue
        assert list(s['columns']) == list(l.columns.names)
        for name, col in s['columns'].items():
            assert _same(col, l.data[name]), (path, name)
    return std


class _Fit:
    names = ['bias_eta_LYA', 'beta_LYA', "sigma_velo_disp_lorentz_QSO"]
    values = np.array([[-0.2, 1.67, 6.86]])
    errors = np.array([[0.01, 0.05, 0.3]])
    covariance = np.arange(9.).reshape(1, 3, 3)
    fval = np.array([0.64])
    is_valid = np.array([True])
    hesse_failed = np.array([False])
    has_accurate_covar = np.array([False])


def test_result_file_conforms_and_decodes(tmp_path):
    from vega_amd.output import Output
    prob = load_problem('full4')
    rng = np.random.default_rng(3)
    models = {name: rng.standard_normal(item.dist_grid.size) for name, item in prob.items.items()}
    params = {'ap': 1.0, 'at': 0.99, 'bias_eta_LYA': -0.2, 'sigma_velo_disp_lorentz_QSO': 6.86, 'growth_rate': 0.97,
              'bias_eta_SiII(1260)': -1.5e-3, 'tiny': 1e-300, 'huge': -1.25e+200, 'count': 7}
    stats = {name: {'masked_size': int(item.data_size), 'chisq': 0.125 * (i + 1), 'reduced_chisq': 1e-4 * (i + 1),
                    'p_value': 1.0, 'bestfit_marg_coeff': None if i else np.array([0.5, -1.5])}
             for i, (name, item) in enumerate(prob.items.items())}
    scan = [{'ap': 0.9 + 0.1 * i, 'bias_eta_LYA': -0.2, 'beta_LYA': 1.6, 'fval': 3.0 - i} for i in range(3)]
    out = Output({'filename': str(tmp_path / 'result')}, prob.items)
    out.analysis = type('A', (), {'grids': {'ap': np.linspace(0.9, 1.1, 3)}})()
    path = out.write_results(models, params, _Fit(), stats, scan)
    hdus = _against_fitslite(path)
    assert hdus[0]['header']['SIMPLE'] is True and hdus[0]['header']['NAXIS'] == 0 and hdus[0]['header']['EXTEND'] is True
    names = [h['header']['EXTNAME'] for h in hdus[1:]]
    assert names == ['MODEL_' + n.upper() for n in prob.items] + ['BESTFIT', 'SCAN']
    for i, (name, item) in enumerate(prob.items.items()):
        h = hdus[1 + i]
        np.testing.assert_array_equal(h['columns'][name + '_MODEL'], models[name])
        np.testing.assert_array_equal(h['columns'][name + '_MODEL_MASK'], item.model_mask)
        np.testing.assert_array_equal(h['columns'][name + '_RP'], item.dist_grid.rp)
        # the parameters went into HIERARCH cards (lower case, parentheses, more than eight characters) with full precision
        for par, val in params.items():
            assert h['header'][par] == val and type(h['header'][par]) is type(val), par
        assert h['header']['chisq'] == 0.125 * (i + 1) and h['header']['masked_size'] == item.data_size
        assert any(c.startswith('HIERARCH bias_eta_SiII(1260) = ') for c in h['cards'])
    best = hdus[len(prob.items) + 1]
    assert list(best['columns']['names']) == _Fit.names
    np.testing.assert_array_equal(best['columns']['values'], _Fit.values[0])
    np.testing.assert_array_equal(best['columns']['covariance'], _Fit.covariance[0])
    assert best['header']['FVAL'] == 0.64 and best['header']['VALID'] is True and best['header']['ACCURATE'] is False
    assert best['header']['TFORM4'] == '3D'
    np.testing.assert_array_equal(hdus[-1]['columns']['fval'], [3.0, 2.0, 1.0, 0.0])


def test_monte_carlo_file_conforms_and_decodes(tmp_path):
    from vega_amd.output import Output
    prob = load_problem('full4')
    rng = np.random.default_rng(8)
    n_fit, names = 5, ['ap', 'at', 'bias_eta_LYA']
    analysis = SimpleNamespace(
        has_monte_carlo=True,
        mc_mocks={name: rng.standard_normal((n_fit, item.data_size)) for name, item in prob.items.items()},
        mc_bestfits={n: rng.standard_normal((n_fit - 1, 2)) for n in names},
        mc_covariances=[rng.standard_normal((3, 3)) for _ in range(n_fit - 1)],
        mc_chisq=[1.5, np.nan, 2.5, 3.5, 4.5], mc_valid_minima=[True, False, True, True, False],
        mc_valid_hesse=[True, False, True, True, True], mc_failed_mask=[False, True, False, False, False])
    out = Output({'filename': str(tmp_path / 'r.fits')}, prob.items, analysis)
    path = out.write_monte_carlo(cpu_id=3)
    hdus = _against_fitslite(path)
    by_name = {h['header']['EXTNAME']: h for h in hdus[1:]}
    assert {'BESTFIT', 'FITINFO', 'MOCKS'} <= set(by_name)
    # reference vega/output.py:455-470: one row per parameter - its name, its best-fit values over the successful mocks, their
    # errors, its rows of the mocks' covariances side by side
    best = by_name['BESTFIT']['columns']
    assert list(best['names']) == names
    for i, n in enumerate(names):
        np.testing.assert_array_equal(best['values'][i], analysis.mc_bestfits[n][:, 0])
        np.testing.assert_array_equal(best['errors'][i], analysis.mc_bestfits[n][:, 1])
        np.testing.assert_array_equal(best['covariance'][i], np.concatenate([c[:, i] for c in analysis.mc_covariances]))
    np.testing.assert_array_equal(by_name['FITINFO']['columns']['chisq'], analysis.mc_chisq)
    np.testing.assert_array_equal(by_name['FITINFO']['columns']['valid_minima'], analysis.mc_valid_minima)
    np.testing.assert_array_equal(by_name['FITINFO']['columns']['failed_mask'], analysis.mc_failed_mask)
    for name in prob.items:
        np.testing.assert_array_equal(by_name['MOCKS']['columns'][name], analysis.mc_mocks[name])


def test_data_distortion_and_covariance_files_conform(tmp_path):
    """What `vega_amd.synthetic` writes for the ingestion tests - the files the unmodified reference read through astropy for
    tests/golden/expected_fits_ingest.npz and expected_dmat_file.npz (its reader accepted them; here the standard does)."""
    from vega_amd import synthetic
    from vega_amd.tables import read_tables
    source = read_tables(GOLDEN / 'inputs' / 'cf_lya-exp.npz')
    data = synthetic.write_data_file(tmp_path / 'cf.fits', source, extra_header=synthetic.PICCA_COSMOLOGY_HEADER)
    hdus = _against_fitslite(data)
    t1 = hdus[1]
    np.testing.assert_array_equal(t1['columns']['DA'], source[0].data['DA'])
    assert t1['columns']['DM'].shape == (2500, 2500) and t1['columns']['CO'].shape == (2500, 2500)
    assert t1['header']['TFORM' + str(list(t1['columns']).index('DM') + 1)] == '2500D'
    for key in ('RPMIN', 'RPMAX', 'RTMAX', 'NP', 'NT'):
        assert t1['header'][key] == source[0].header[key]
    assert t1['header']['OMEGAM'] == synthetic.PICCA_COSMOLOGY_HEADER['OMEGAM']
    (tmp_path / 'dmat').mkdir()
    synthetic.write_dmat_file_case(tmp_path / 'dmat', source, coef=2)
    written = [path for path in (tmp_path / 'dmat').iterdir() if str(path).endswith(('.fits', '.fits.gz'))]
    assert len(written) == 2           # the distortion file (COEFMOD in its header, HDU 2 with the model grid), the covariance file
    for path in written:
        _against_fitslite(path)
    gc = synthetic.write_global_covariance(tmp_path / 'gc.fits', np.eye(7) * 2.5)
    np.testing.assert_array_equal(_against_fitslite(gc)[1]['columns'][list(_against_fitslite(gc)[1]['columns'])[0]], np.eye(7) * 2.5)
    _against_fitslite(synthetic.write_stacked_deltas(tmp_path / 'deltas.fits', n_pix=50))
    _against_fitslite(synthetic.write_object_catalog(tmp_path / 'cat.fits', n_obj=40))


def test_the_validator_rejects_what_the_standard_forbids(tmp_path):
    """The checks bite: a file the package wrote, damaged in one place at a time."""
    from vega_amd import fitslite
    good = tmp_path / 'good.fits'
    fitslite.write_tables(good, [('T', [('A', 'D', np.arange(4.)), ('B', '2K', np.arange(8).reshape(4, 2))], {'some key': 1.5})])
    check_file(good)
    raw = good.read_bytes()

    def damaged(edit):
        buf = bytearray(raw)
        edit(buf)
        path = tmp_path / 'bad.fits'
        path.write_bytes(bytes(buf))
        with pytest.raises(FitsError):
            check_file(path)

    def card_offset(key):
        return raw.index(key.encode().ljust(8) + b'=')

    damaged(lambda b: b.extend(b'\0' * 100))                                         # not whole blocks
    damaged(lambda b: b.__setitem__(slice(len(raw) - 1, len(raw)), b'x'))              # fill behind the table not zero
    damaged(lambda b: b.__setitem__(slice(card_offset('NAXIS1') + 10, card_offset('NAXIS1') + 30), b'24'.ljust(20)))    # not right-justified
    off = card_offset('PCOUNT'); gc = card_offset('GCOUNT')
    def swap(b):
        b[off:off + 80], b[gc:gc + 80] = bytes(b[gc:gc + 80]), bytes(b[off:off + 80])
    damaged(swap)                                                                        # mandatory keywords out of order
    damaged(lambda b: b.__setitem__(slice(card_offset('TFORM2') + 11, card_offset('TFORM2') + 13), b'3K'))   # fields do not add up to NAXIS1
    damaged(lambda b: b.__setitem__(slice(card_offset('TTYPE1'), card_offset('TTYPE1') + 5), b'ttype'))      # lower-case keyword
    damaged(lambda b: b.__setitem__(card_offset('BITPIX') + 40, 0x07))                  # a control character in a card
    end = raw.index(b'END' + b' ' * 77)
    damaged(lambda b: b.__setitem__(slice(end + 3, end + 4), b'x'))                     # END card not blank behind END
    hier = raw.index(b'HIERARCH some key')
    damaged(lambda b: b.__setitem__(slice(hier + 8, hier + 9), b'_'))                   # HIERARCH without its blank


@pytest.mark.skipif(not (REFERENCE / 'tests' / 'data').is_dir(), reason='the reference tree exists in the build container only')
def test_the_package_reader_on_files_astropy_wrote():
    """fitslite on the FITS files of the reference tree (written by astropy / picca) against the validator's independent decoding
    of the same bytes, and against the committed .npz bundles the tests run on."""
    from vega_amd.tables import read_tables
    files = sorted(list((REFERENCE / 'tests' / 'data').glob('*.fits*')) + list((REFERENCE / 'vega' / 'models').glob('*/*.fits*')))
    assert len(files) >= 12
    for path in files:
        std = _against_fitslite(path)
        assert any(h['columns'] for h in std), path
    converted = {'cf_lya-exp': 'tests/data/cf_lya-exp.fits.gz', 'xcf_lya-exp': 'tests/data/xcf_lya-exp.fits.gz',
                 'metal_dmat_lya': 'tests/data/metal_dmat_lya.fits.gz', 'metal_xdmat_lya': 'tests/data/metal_xdmat_lya.fits.gz'}
    checked = 0
    for stem, rel in converted.items():
        bundle = GOLDEN / 'inputs' / f'{stem}.npz'
        if not bundle.is_file():
            continue
        std = [h for h in check_file(REFERENCE / rel)[1:] if h['columns']]
        for table, h in zip(read_tables(bundle), std):
            for name in table.names:
                assert _same(table.data[name], h['columns'][name]), (stem, name)
            for key, value in table.header.items():
                assert h['header'][key] == value, (stem, key)
        checked += 1
    assert checked >= 2
