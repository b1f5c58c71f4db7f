"""An independent FITS validator / decoder for the tests, written from the standard - "Definition of the Flexible Image
Transport System (FITS)", version 4.0 (IAU FITS Working Group, 2018) - and NOT from vega_amd/fitslite.py, whose writer and
reader it checks (astropy is absent from this image, so until now every file the package wrote had only ever been read by the
package's own reader).  Section numbers below are the standard's.

    check_file(path) -> list of HDUs, each {'header': {...}, 'cards': [...], 'columns': {name: ndarray} or None}
                        raises FitsError with the offending offset / card on the first violation

What is checked:
  3.1      the file is a whole number of 2880-byte blocks (3.3.2 / 3.5: nothing but blocks)
  4.1.1    header = 80-character card images, ASCII text 0x20-0x7E only; the END card is 'END' + 77 blanks; the rest of the
           last header block is blank
  4.1.2    keyword names: columns 1-8, left-justified, upper-case letters / digits / '-' / '_' only; value indicator '= ' in
           columns 9-10
  4.2      fixed-format values of the mandatory keywords: logical T / F in column 30, integers right-justified ending in
           column 30, strings opening with a quote in column 11 and at least eight characters between the quotes for XTENSION;
           free-format strings, integers, reals (upper-case E or D exponent) and logicals elsewhere; '/' comments
  4.2.1.2  CONTINUE long strings: '&' at the end of the continued string, CONTINUE cards with a string in columns 11-
  4.4.1    mandatory keywords and their order: SIMPLE, BITPIX, NAXIS, NAXISn, END in the primary header (4.4.1.1), XTENSION,
           BITPIX, NAXIS, NAXISn, PCOUNT, GCOUNT in extensions (4.4.1.2)
  7.3.1    binary tables: XTENSION = 'BINTABLE', BITPIX = 8, NAXIS = 2, NAXIS1, NAXIS2, PCOUNT, GCOUNT = 1, TFIELDS in this
           order; TFORMn for n = 1 .. TFIELDS of the form rTa with T in LXBIJKAEDCMPQ; the fields' bytes add up to NAXIS1
  7.3.2    TTYPEn are strings; TDIMn = '(l,m,...)' with a product that does not exceed the repeat count
  7.3.3    the data are NAXIS1 x NAXIS2 + PCOUNT bytes, big-endian (5.2, 5.3), the remainder of the last block zero bytes
  HIERARCH the ESO convention (registered with the FITS Support Office): 'HIERARCH' in columns 1-8, a blank, keyword tokens,
           '=' and a free-format value
"""
import gzip
import re

import numpy as np

BLOCK, CARD = 2880, 80
_KEYWORD = re.compile(r'^[A-Z0-9_-]{1,8} *$')
_INT = re.compile(r'^[+-]?[0-9]+$')
_REAL = re.compile(r'^[+-]?([0-9]+\.?[0-9]*|\.[0-9]+)([ED][+-]?[0-9]+)?$')
# (type, bytes per element, numpy dtype) of 7.3.1's table of TFORM codes
_TFORM = {'L': (1, 'S1'), 'X': (1, 'u1'), 'B': (1, 'u1'), 'I': (2, '>i2'), 'J': (4, '>i4'), 'K': (8, '>i8'), 'A': (1, 'S1'),
          'E': (4, '>f4'), 'D': (8, '>f8'), 'C': (8, '>c8'), 'M': (16, '>c16'), 'P': (8, '>i4'), 'Q': (16, '>i8')}


class FitsError(AssertionError):
    pass


def _string_at(card, start):
    """A character string whose opening quote is at `start` (4.2.1): (value with '' undone and trailing blanks removed, index
    after the closing quote)."""
    if card[start] != "'":
        raise FitsError(f'string value must open with a quote in column {start + 1}: {card!r}')
    i, out = start + 1, []
    while True:
        j = card.find("'", i)
        if j < 0:
            raise FitsError(f'string value without closing quote: {card!r}')
        out.append(card[i:j])
        if card[j + 1:j + 2] == "'":
            out.append("'")
            i = j + 2
            continue
        text = ''.join(out)
        # (trailing blanks are not significant; a string of blanks only stands for one blank, the null string for itself)
        return (text.rstrip(' ') or (' ' if text else '')), j + 1


def _free_value(text, card):
    """A free-format value field (4.2): string, logical, integer or real; returns (value, rest)."""
    text = text.lstrip(' ')
    if not text:
        return None, ''
    if text[0] == "'":
        value, end = _string_at(text, 0)
        return value, text[end:]
    token = text.split('/', 1)[0].strip(' ')
    rest = text[len(text.split('/', 1)[0]):]
    if token in ('T', 'F'):
        return token == 'T', rest
    if _INT.match(token):
        return int(token), rest
    if _REAL.match(token):
        return float(token.replace('D', 'E')), rest
    raise FitsError(f'not a FITS value ({token!r}): {card!r}')


def _comment_ok(rest, card):
    rest = rest.strip(' ')
    if rest and not rest.startswith('/'):
        raise FitsError(f'text after the value that is not a / comment: {card!r}')


def _parse_header(buf, pos, primary):
    cards, header, order = [], {}, []
    ended = False
    continued = None
    while not ended:
        block = buf[pos:pos + BLOCK]
        if len(block) != BLOCK:
            raise FitsError(f'header block at byte {pos} is truncated')
        pos += BLOCK
        for k in range(0, BLOCK, CARD):
            raw = block[k:k + CARD]
            if any(c < 0x20 or c > 0x7E for c in raw):
                raise FitsError(f'header card with a byte outside ASCII text at byte {pos - BLOCK + k}')
            card = raw.decode('ascii')
            if ended:
                if card.strip(' '):
                    raise FitsError(f'non-blank card after END: {card!r}')
                continue
            if card.startswith('END') and not card[3:].strip(' '):
                if card != 'END' + ' ' * 77:
                    raise FitsError('the END card must be END followed by 77 blanks')
                ended = True
                continue
            cards.append(card)
            key = card[:8]
            if key == 'CONTINUE':
                if continued is None:
                    raise FitsError(f'CONTINUE without a string ending in &: {card!r}')
                if card[8:10] != '  ':
                    raise FitsError(f'CONTINUE must have blanks in columns 9-10: {card!r}')
                value, end = _string_at(card, 10)
                _comment_ok(card[end:], card)
                whole = header[continued][:-1] + value
                header[continued] = whole
                continued = continued if value.endswith('&') else None
                if continued is None and header[order[-1]].endswith('&'):
                    pass
                continue
            if continued is not None:
                raise FitsError(f'a string ending in & must be followed by a CONTINUE card: {card!r}')
            if key == 'HIERARCH':
                if card[8] != ' ' or '=' not in card:
                    raise FitsError(f'HIERARCH card: a blank in column 9 and an = sign: {card!r}')
                name, _, field = card[9:].partition('=')
                name = name.strip(' ')
                if not name or any(c < ' ' or c > '~' for c in name):
                    raise FitsError(f'HIERARCH keyword: {card!r}')
                value, rest = _free_value(field, card)
                _comment_ok(rest, card)
                header[name] = value
                order.append(name)
                continue
            if key.strip(' ') in ('COMMENT', 'HISTORY', ''):
                continue
            if not _KEYWORD.match(key) or key[0] == ' ':
                raise FitsError(f'keyword name (columns 1-8, upper case, left-justified): {card!r}')
            name = key.rstrip(' ')
            if card[8:10] != '= ':
                raise FitsError(f'value indicator "= " in columns 9-10: {card!r}')
            value, rest = _free_value(card[10:], card)
            _comment_ok(rest, card)
            if isinstance(value, str) and value.endswith('&'):
                continued = name
            if name in header:
                raise FitsError(f'keyword {name} appears twice')
            header[name] = value
            order.append(name)
    _mandatory(cards, header, order, primary)
    return cards, header, pos


def _fixed_logical(card):
    if card[29] not in 'TF' or card[10:29].strip(' '):
        raise FitsError(f'fixed format: logical T / F in column 30: {card!r}')


def _fixed_int(card):
    if not _INT.match(card[10:30].strip(' ')) or card[29] == ' ':
        raise FitsError(f'fixed format: integer right-justified in columns 11-30: {card!r}')


def _mandatory(cards, header, order, primary):
    by_name = {c[:8].rstrip(' '): c for c in cards if c[8:10] == '= '}
    first = 'SIMPLE' if primary else 'XTENSION'
    naxis = header.get('NAXIS')
    if not isinstance(naxis, int) or not 0 <= naxis <= 999:
        raise FitsError('NAXIS must be an integer 0 .. 999')
    want = [first, 'BITPIX', 'NAXIS'] + [f'NAXIS{i}' for i in range(1, naxis + 1)]
    if not primary:
        want += ['PCOUNT', 'GCOUNT']
    if order[:len(want)] != want:
        raise FitsError(f'mandatory keywords must come first, in the order {want}: found {order[:len(want)]}')
    if primary:
        _fixed_logical(by_name['SIMPLE'])
        if header['SIMPLE'] is not True:
            raise FitsError('SIMPLE must be T')
    else:
        card = by_name['XTENSION']
        if card[10] != "'" or card.find("'", 11) < 19:
            raise FitsError(f'XTENSION: a string from column 11 with at least eight characters between the quotes: {card!r}')
    for name in want[1:]:
        _fixed_int(by_name[name])
    if header['BITPIX'] not in (8, 16, 32, 64, -32, -64):
        raise FitsError(f"BITPIX = {header['BITPIX']}")
    for i in range(1, naxis + 1):
        if header[f'NAXIS{i}'] < 0:
            raise FitsError(f'NAXIS{i} is negative')
    if 'EXTEND' in by_name:
        _fixed_logical(by_name['EXTEND'])


def _tform(text, card_name):
    m = re.match(r'^([0-9]*)([LXBIJKAEDCMPQ])(.*)$', text.strip(' '))
    if not m:
        raise FitsError(f'{card_name} = {text!r} is not of the form rTa')
    repeat = int(m.group(1)) if m.group(1) else 1
    return repeat, m.group(2)


def _bintable(header, order, data):
    want = ['XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'PCOUNT', 'GCOUNT', 'TFIELDS']
    if order[:8] != want:
        raise FitsError(f'binary table: the first eight keywords must be {want}, found {order[:8]}')
    if header['XTENSION'] != 'BINTABLE' or header['BITPIX'] != 8 or header['NAXIS'] != 2 or header['GCOUNT'] != 1:
        raise FitsError('binary table: XTENSION = BINTABLE, BITPIX = 8, NAXIS = 2, GCOUNT = 1')
    if header['PCOUNT'] < 0 or not 0 <= header['TFIELDS'] <= 999:
        raise FitsError('binary table: PCOUNT >= 0, TFIELDS 0 .. 999')
    rowlen, nrow = header['NAXIS1'], header['NAXIS2']
    fields, offset = [], 0
    for n in range(1, header['TFIELDS'] + 1):
        if f'TFORM{n}' not in header or not isinstance(header[f'TFORM{n}'], str):
            raise FitsError(f'TFORM{n} is required and must be a string')
        repeat, code = _tform(header[f'TFORM{n}'], f'TFORM{n}')
        size, dtype = _TFORM[code]
        nbytes = (repeat + 7) // 8 if code == 'X' else repeat * size
        name = header.get(f'TTYPE{n}')
        if name is not None and not isinstance(name, str):
            raise FitsError(f'TTYPE{n} must be a string')
        if f'TDIM{n}' in header:
            m = re.match(r'^\(([0-9]+(,[0-9]+)*)\)$', str(header[f'TDIM{n}']).replace(' ', ''))
            if not m or int(np.prod([int(v) for v in m.group(1).split(',')])) > repeat:
                raise FitsError(f"TDIM{n} = {header[f'TDIM{n}']!r} does not describe {repeat} elements")
        fields.append((name or f'col{n}', code, repeat, offset, dtype))
        offset += nbytes
    if offset != rowlen:
        raise FitsError(f'the fields take {offset} bytes per row, NAXIS1 = {rowlen}')
    columns = {}
    table = np.frombuffer(data, dtype='u1', count=rowlen * nrow).reshape(nrow, rowlen)
    for name, code, repeat, off, dtype in fields:
        size = _TFORM[code][0]
        raw = np.ascontiguousarray(table[:, off:off + repeat * size])
        if code == 'A':
            col = np.array([bytes(r).decode('ascii').rstrip(' \x00') for r in raw])
        elif code == 'L':
            flat = raw.reshape(-1)
            if not np.isin(flat, (ord('T'), ord('F'), 0)).all():
                raise FitsError(f'logical column {name}: bytes other than T, F, NUL')
            col = (flat == ord('T')).reshape(nrow, repeat)
        else:
            col = raw.view(dtype).astype(np.dtype(dtype).newbyteorder('=')).reshape(nrow, repeat)
        if code != 'A' and repeat == 1:
            col = col[:, 0]
        columns[name] = col
    return columns


def check_file(path):
    path = str(path)
    with (gzip.open(path, 'rb') if path.endswith('.gz') else open(path, 'rb')) as fh:
        buf = fh.read()
    if len(buf) == 0 or len(buf) % BLOCK:
        raise FitsError(f'{len(buf)} bytes: not a whole number of {BLOCK}-byte blocks')
    hdus, pos = [], 0
    while pos < len(buf):
        primary = pos == 0
        cards, header, pos = _parse_header(buf, pos, primary)
        order = [c[:8].rstrip(' ') for c in cards if c[8:10] == '= ' and c[:8] != 'HIERARCH']
        size = 0
        if header['NAXIS'] > 0:
            size = abs(header['BITPIX']) // 8 * int(np.prod([header[f'NAXIS{i}'] for i in range(1, header['NAXIS'] + 1)]))
        if not primary:
            size = header['GCOUNT'] * (size + header['PCOUNT'])
        padded = (size + BLOCK - 1) // BLOCK * BLOCK
        if pos + padded > len(buf):
            raise FitsError(f'HDU {len(hdus)}: the data run past the end of the file')
        data = buf[pos:pos + size]
        pad = buf[pos + size:pos + padded]
        columns = None
        if not primary and header.get('XTENSION') == 'BINTABLE':
            columns = _bintable(header, order, data)
            if pad.strip(b'\x00'):
                raise FitsError(f'HDU {len(hdus)}: the fill behind a binary table must be zero bytes')
        hdus.append({'header': header, 'cards': cards, 'columns': columns})
        pos += padded
    return hdus
