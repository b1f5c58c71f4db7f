"""Minimal pure-NumPy FITS binary-table reader.

The reference reads its static inputs (template P(k), data vector, distortion
matrix, covariance, coordinate grids, metal matrices) with ``astropy.io.fits``
(reference: vega/data.py:302-394, vega/data.py:441-473, vega/data.py:578-687,
vega/vega_interface.py:667-703).  astropy is not available in this image, so the
engine's host side carries its own reader for the one FITS flavour those files
use: a primary HDU followed by BINTABLE extensions, optionally gzip-compressed.

Only what the hot-path static state needs is supported: TFORM codes
L, B, I, J, K, E, D, A with repeat counts and TDIM-less vector columns.
"""
import builtins
import gzip

import numpy as np

_BLOCK = 2880
_CARD = 80

_TFORM_DTYPES = {
    'L': ('i1', 1), 'B': ('u1', 1), 'I': ('>i2', 2), 'J': ('>i4', 4),
    'K': ('>i8', 8), 'E': ('>f4', 4), 'D': ('>f8', 8), 'A': ('S', 1),
}


def _parse_value(raw):
    raw = raw.strip()
    if not raw:
        return None
    if raw.startswith("'"):
        end = raw.find("'", 1)
        while end != -1 and end + 1 < len(raw) and raw[end + 1] == "'":
            end = raw.find("'", end + 2)
        return raw[1:end].replace("''", "'").rstrip()
    raw = raw.split('/')[0].strip()
    if raw in ('T', 'F'):
        return raw == 'T'
    try:
        return int(raw)
    except ValueError:
        pass
    try:
        return float(raw.replace('D', 'E'))
    except ValueError:
        return raw


def _read_header(buf, pos):
    header = {}
    while True:
        block = buf[pos:pos + _BLOCK]
        if len(block) < _BLOCK:
            raise ValueError('Truncated FITS header')
        pos += _BLOCK
        done = False
        for i in range(0, _BLOCK, _CARD):
            card = block[i:i + _CARD].decode('ascii', errors='replace')
            key = card[:8].strip()
            if key == 'END':
                done = True
                break
            if key == 'HIERARCH' and '=' in card:
                # `HIERARCH long or lower-case keyword = value` (the ESO convention astropy writes for `header['hierarch x']`):
                # kept under the keyword itself, case preserved
                name, _, raw = card[8:].partition('=')
                header[name.strip()] = _parse_value(raw)
            elif card[8:10] == '= ':
                header[key] = _parse_value(card[10:])
        if done:
            return header, pos


def _parse_tform(tform):
    tform = tform.strip()
    i = 0
    while i < len(tform) and tform[i].isdigit():
        i += 1
    repeat = int(tform[:i]) if i else 1
    code = tform[i]
    if code not in _TFORM_DTYPES:
        raise ValueError(f'Unsupported TFORM {tform}')
    return repeat, code


class _Columns:
    def __init__(self, names):
        self.names = list(names)


class _TableData:
    def __init__(self, rec, names, logical=()):
        self._rec = rec
        self._names = names
        self._logical = set(logical)

    def __getitem__(self, name):
        if name not in self._names:
            raise KeyError(name)
        col = self._rec[name]
        if col.dtype.kind == 'S':
            return np.char.decode(col, 'ascii')
        if name in self._logical:
            return col == ord('T')          # TFORM 'L': bool, as astropy hands it out
        # native-endian copy, as astropy hands out
        return np.ascontiguousarray(col.astype(col.dtype.newbyteorder('=')))

    def __len__(self):
        return len(self._rec)


class HDU:
    def __init__(self, header, data=None, names=()):
        self.header = header
        self.data = data
        self.columns = _Columns(names)


class HDUList(list):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass


def open(path):  # noqa: A001 - mirrors astropy.io.fits.open
    """Read every HDU of ``path`` (plain or .gz) into memory."""
    path = str(path)
    if path.endswith('.gz'):
        with gzip.open(path, 'rb') as f:
            buf = f.read()
    else:
        with builtins.open(path, 'rb') as f:
            buf = f.read()

    hdus = HDUList()
    pos = 0
    while pos < len(buf):
        if not buf[pos:pos + _BLOCK].strip(b'\x00 '):
            break
        header, pos = _read_header(buf, pos)
        naxis = header.get('NAXIS', 0)
        size = 0
        if naxis:
            size = abs(header.get('BITPIX', 8)) // 8
            for ax in range(1, naxis + 1):
                size *= header[f'NAXIS{ax}']
            size += header.get('PCOUNT', 0)
        xt = header.get('XTENSION')
        if xt is not None and xt.strip() == 'BINTABLE' and size:
            nrow = header['NAXIS2']
            rowlen = header['NAXIS1']
            names, formats, logical = [], [], []
            for c in range(1, header['TFIELDS'] + 1):
                repeat, code = _parse_tform(header[f'TFORM{c}'])
                if code == 'L':
                    logical.append(header[f'TTYPE{c}'].strip())
                base, _ = _TFORM_DTYPES[code]
                if code == 'A':
                    fmt = f'S{repeat}'
                elif repeat == 1:
                    fmt = base
                else:
                    fmt = (base, (repeat,))
                names.append(header[f'TTYPE{c}'].strip())
                formats.append(fmt)
            dt = np.dtype({'names': names, 'formats': formats})
            if dt.itemsize != rowlen:
                raise ValueError(f'Row length mismatch: {dt.itemsize} vs {rowlen}')
            rec = np.frombuffer(buf, dtype=dt, count=nrow, offset=pos)
            hdus.append(HDU(header, _TableData(rec, names, logical), names))
        else:
            hdus.append(HDU(header))
        pos += ((size + _BLOCK - 1) // _BLOCK) * _BLOCK
    return hdus


# ------------------------------------------------------------------------------------------ writer
def _card(key, value, comment=''):
    if isinstance(value, tuple):            # (value, comment)
        value, comment = value
    if isinstance(value, np.bool_):
        value = bool(value)
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, (int, np.integer)):
        v = f'{int(value):>20}'
    elif isinstance(value, (float, np.floating)):
        v = f'{float(value):>20.13E}'
    else:
        text = "'" + str(value).replace("'", "''").ljust(8) + "'"
        v = f'{text:<20}'
    if len(key) > 8 or key != key.upper() or ' ' in key:
        # HIERARCH card (what astropy makes of `header['hierarch ' + key] = value`, reference vega/output.py:212-228)
        if isinstance(value, bool):
            hv = 'T' if value else 'F'
        elif isinstance(value, (int, np.integer)):
            hv = str(int(value))
        elif isinstance(value, (float, np.floating)):
            hv = repr(float(value)).upper() if np.isfinite(value) else "'" + str(value) + "'"
            if 'E' not in hv and '.' not in hv:
                hv += '.0'
        else:
            hv = "'" + str(value).replace("'", "''") + "'"
        card = f'HIERARCH {key} = {hv}'
        if len(card) > _CARD:
            card = f'HIERARCH {key}={hv}'
        if len(card) > _CARD:
            raise ValueError(f'header keyword {key!r} with its value does not fit one FITS card')
        return card.ljust(_CARD)
    card = f'{key:<8}= {v}'
    if comment:
        card += f' / {comment}'
    return card[:_CARD].ljust(_CARD)


def _header_bytes(cards):
    text = ''.join(cards) + 'END'.ljust(_CARD)
    text += ' ' * (-len(text) % _BLOCK)
    return text.encode('ascii')


def write_tables(path, tables, overwrite=False):
    """Write a FITS file: empty primary HDU + one BINTABLE per entry of ``tables`` =
    [(extname, [(column name, TFORM, array), ...]) or (extname, columns, {header keyword: value}), ...].  TFORM: 'D' / 'nD' (float64), 'K' / 'nK' (int64),
    'L' (logical), 'nA' (strings).  Vector columns take arrays of shape [rows, n].  The layout is the one
    ``astropy.io.fits.BinTableHDU.from_columns`` produces for the same columns (reference vega/output.py)."""
    import os
    if os.path.exists(path) and not overwrite:
        raise OSError(f'File {path!r} already exists.')
    out = [_header_bytes([_card('SIMPLE', True, 'conforms to FITS standard'), _card('BITPIX', 8),
                          _card('NAXIS', 0), _card('EXTEND', True)])]
    for entry in tables:
        extname, columns = entry[0], entry[1]
        extra = entry[2] if len(entry) > 2 else {}
        names, formats, arrays = [], [], []
        nrow = None
        for name, tform, arr in columns:
            repeat, code = _parse_tform(tform)
            arr = np.asarray(arr)
            if code == 'A':
                arr = np.char.encode(np.char.ljust(arr.astype(str), repeat), 'ascii').astype(f'S{repeat}')
                fmt = f'S{repeat}'
            elif code == 'L':
                arr = np.where(arr.astype(bool), ord('T'), ord('F')).astype('i1')
                fmt = 'i1' if repeat == 1 else ('i1', (repeat,))
            else:
                base, _ = _TFORM_DTYPES[code]
                arr = arr.astype(base)
                fmt = base if repeat == 1 else (base, (repeat,))
            if repeat > 1 and code != 'A' and (arr.ndim != 2 or arr.shape[1] != repeat):
                raise ValueError(f'column {name!r}: array shape {arr.shape} does not match TFORM {tform}')
            if nrow is None:
                nrow = arr.shape[0]
            elif arr.shape[0] != nrow:
                raise ValueError(f'column {name!r} has {arr.shape[0]} rows, expected {nrow}')
            names.append(name); formats.append(fmt); arrays.append(arr)
        dt = np.dtype({'names': names, 'formats': formats})
        rec = np.zeros(nrow or 0, dtype=dt)
        for name, arr in zip(names, arrays):
            rec[name] = arr
        cards = [_card('XTENSION', 'BINTABLE', 'binary table extension'), _card('BITPIX', 8), _card('NAXIS', 2),
                 _card('NAXIS1', dt.itemsize), _card('NAXIS2', nrow or 0), _card('PCOUNT', 0), _card('GCOUNT', 1),
                 _card('TFIELDS', len(names))]
        for i, (name, tform, _) in enumerate(columns, start=1):
            cards += [_card(f'TTYPE{i}', name), _card(f'TFORM{i}', tform)]
        cards.append(_card('EXTNAME', str(extname).upper()))          # (astropy stores hdu.name upper case)
        for key, value in extra.items():
            cards.append(_card(key, value))
        data = rec.tobytes()
        out += [_header_bytes(cards), data + b'\x00' * (-len(data) % _BLOCK)]
    with builtins.open(path, 'wb') as f:
        for chunk in out:
            f.write(chunk)
