"""Table-bundle IO for the engine's static inputs.

Two on-disk forms are accepted everywhere a reference config names a file
(template P(k): reference vega/vega_interface.py:667-703; data / distortion /
covariance: vega/data.py:285-473; metal matrices: vega/data.py:556-687):

* FITS binary tables (``.fits`` / ``.fits.gz``) through :mod:`vega_amd.fitslite`;
* ``.npz`` bundles written by :func:`write_bundle` - one array per column under
  the key ``"<hdu>/<COLUMN>"`` plus a JSON header per HDU under
  ``"<hdu>/__header__"``.  The committed test fixtures use this form.
"""
import json
import os
from pathlib import Path

import numpy as np

from . import fitslite


class Table:
    """One HDU: ``.header`` (dict) and ``.data[column]`` / ``.names``."""

    def __init__(self, header, columns):
        self.header = dict(header)
        self.data = columns
        self.names = list(columns.keys())

    def has(self, name):
        return name in self.data


def read_tables(path):
    """Return the list of tables (HDU 1, 2, ...) held by ``path``."""
    path = str(path)
    if path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as z:
            hdus = sorted({int(key.split('/')[0]) for key in z.files})
            out = []
            for h in hdus:
                header = json.loads(str(z[f'{h}/__header__']))
                order = json.loads(str(z[f'{h}/__order__']))
                out.append(Table(header, {c: z[f'{h}/{c}'] for c in order}))
        return out

    hdul = fitslite.open(path)
    out = []
    for hdu in hdul[1:]:
        if hdu.data is None:
            continue
        cols = {name: hdu.data[name] for name in hdu.columns.names}
        out.append(Table(hdu.header, cols))
    return out


def write_bundle(path, tables, drop_header_prefixes=('TTYPE', 'TFORM', 'NAXIS', 'TDIM')):
    """Write ``tables`` (list of :class:`Table`) as an ``.npz`` bundle."""
    arrays = {}
    for i, tab in enumerate(tables, start=1):
        header = {k: v for k, v in tab.header.items()
                  if not k.startswith(drop_header_prefixes)
                  and isinstance(v, (int, float, str, bool))}
        arrays[f'{i}/__header__'] = np.array(json.dumps(header))
        arrays[f'{i}/__order__'] = np.array(json.dumps(tab.names))
        for c in tab.names:
            arrays[f'{i}/{c}'] = np.asarray(tab.data[c])
    np.savez_compressed(path, **arrays)


def find_file(path, search_dirs=()):
    """Resolve ``path``: as given (absolute / cwd-relative), then under each of
    ``search_dirs`` and the directories named by ``$VEGA_AMD_PATH``.

    Mirrors the role of the reference's ``utils.find_file``
    (reference vega/utils.py:230-268) without its package-relative lookups.
    """
    p = Path(os.path.expandvars(str(path)))
    if p.is_file():
        return p
    dirs = [Path(d) for d in search_dirs]
    dirs += [Path(d) for d in os.environ.get('VEGA_AMD_PATH', '').split(os.pathsep) if d]
    for d in dirs:
        if (d / p).is_file():
            return d / p
    raise RuntimeError(f'The path/file does not exist: {p} (searched {dirs})')
