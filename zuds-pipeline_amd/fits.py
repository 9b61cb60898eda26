"""Dependency-free single-HDU FITS image reader / writer.

The reference reads and writes its products with ``astropy.io.fits``
(``zuds/fitsfile.py:69-94,146-178``); astropy is not available on the GPU box, and
this path only ever touches primary-HDU images (``_DATA_HDU = _HEADER_HDU = 0``,
``zuds/fitsfile.py:37-38``).  FITS standard 4.0: 2880-byte blocks, 80-character
cards, big-endian data, BITPIX 8 / 16 / 32 / 64 / -32 / -64, BSCALE / BZERO.
"""
import numpy as np

BLOCK = 2880
_BITPIX_DTYPE = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}
_MANDATORY = ('SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND')


def _parse_value(s):
    """Value field of a card (columns 11-80) -> (python value, comment)."""
    s = s.rstrip()
    t = s.lstrip()
    if t.startswith("'"):
        # quoted string, '' is an escaped quote
        i = 1
        out = []
        while i < len(t):
            if t[i] == "'":
                if i + 1 < len(t) and t[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(t[i])
            i += 1
        rest = t[i + 1:]
        comment = rest.split('/', 1)[1].strip() if '/' in rest else ''
        return ''.join(out).rstrip(), comment
    if '/' in t:
        v, comment = t.split('/', 1)
        v, comment = v.strip(), comment.strip()
    else:
        v, comment = t.strip(), ''
    if v == 'T':
        return True, comment
    if v == 'F':
        return False, comment
    if v == '':
        return None, comment
    try:
        return int(v), comment
    except ValueError:
        pass
    try:
        return float(v.replace('D', 'E').replace('d', 'e')), comment
    except ValueError:
        return v, comment


def _read_header(read, what):
    header, comments = {}, {}
    nblocks = 0
    done = False
    while not done:
        block = read(BLOCK)
        if len(block) < BLOCK:
            raise ValueError(f'{what}: truncated FITS header')
        nblocks += 1
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode('ascii', 'replace')
            key = card[:8].strip()
            if key == 'END':
                done = True
                break
            if not key or key in ('COMMENT', 'HISTORY') or card[8:10] != '= ':
                continue
            val, com = _parse_value(card[10:])
            if val is None:
                continue
            header[key] = val
            comments[key] = com
    if header.get('SIMPLE') is not True:
        raise ValueError(f'{what}: not a standard FITS file (SIMPLE != T)')
    return header, comments, nblocks * BLOCK


def read_header(path):
    """(header dict, comments dict, data offset in bytes)."""
    with open(path, 'rb') as f:
        return _read_header(f.read, path)


def parse_header(block):
    """(header dict, comments dict) of header bytes as ``header_block`` emits them: what ``read_header``
    returns for a file that starts with them."""
    import io
    h, c, _ = _read_header(io.BytesIO(bytes(block)).read, 'header block')
    return h, c


def read(path, header_only=False):
    """Read the primary HDU: (data or None, header dict, comments dict)."""
    header, comments, off = read_header(path)
    if header_only:
        return None, header, comments
    naxis = int(header.get('NAXIS', 0))
    if naxis == 0:
        return None, header, comments
    shape = tuple(int(header[f'NAXIS{i}']) for i in range(naxis, 0, -1))
    bitpix = int(header['BITPIX'])
    if bitpix not in _BITPIX_DTYPE:
        raise ValueError(f'{path}: unsupported BITPIX {bitpix}')
    dt = np.dtype(_BITPIX_DTYPE[bitpix])
    count = int(np.prod(shape))
    with open(path, 'rb') as f:
        f.seek(off)
        raw = np.fromfile(f, dtype=dt, count=count)
    if raw.size != count:
        raise ValueError(f'{path}: truncated FITS data ({raw.size} of {count} values)')
    data = raw.reshape(shape).astype(dt.newbyteorder('='))
    bscale = header.get('BSCALE', 1)
    bzero = header.get('BZERO', 0)
    if bscale != 1 or bzero != 0:
        if bscale == 1 and float(bzero).is_integer() and bitpix > 0:
            unsigned = {16: (32768, np.uint16), 32: (2147483648, np.uint32)}
            if bitpix in unsigned and int(bzero) == unsigned[bitpix][0]:
                data = (data.astype(np.int64) + int(bzero)).astype(unsigned[bitpix][1])
            else:
                data = data.astype(np.int64) + int(bzero)
        else:
            data = (data * np.float64(bscale) + np.float64(bzero)).astype(np.float32)
    return data, header, comments


def read_raw(path, out=None):
    """The data block of the primary HDU as it lies on disk, undecoded.

    Returns (raw uint8 array of nbytes, header, comments, info) with info =
    dict(bitpix, shape, bscale, bzero, count).  ``out``: optional writable uint8 buffer
    (e.g. pinned memory) of at least nbytes to read into.  Decoding is left to
    ``zm_fits_decode_dev`` on the device (``device.load_fits``)."""
    header, comments, off = read_header(path)
    naxis = int(header.get('NAXIS', 0))
    if naxis == 0:
        raise ValueError(f'{path}: no image data in the primary HDU')
    shape = tuple(int(header[f'NAXIS{i}']) for i in range(naxis, 0, -1))
    bitpix = int(header['BITPIX'])
    if bitpix not in _BITPIX_DTYPE:
        raise ValueError(f'{path}: unsupported BITPIX {bitpix}')
    count = int(np.prod(shape))
    nbytes = count * abs(bitpix) // 8
    if out is None:
        out = np.empty(nbytes, dtype=np.uint8)
    buf = memoryview(out)[:nbytes]
    with open(path, 'rb', buffering=0) as f:
        f.seek(off)
        got = 0
        while got < nbytes:
            k = f.readinto(buf[got:])
            if not k:
                raise ValueError(f'{path}: truncated FITS data ({got} of {nbytes} bytes)')
            got += k
    info = dict(bitpix=bitpix, shape=shape, count=count, nbytes=nbytes,
                bscale=float(header.get('BSCALE', 1)), bzero=float(header.get('BZERO', 0)))
    return out[:nbytes], header, comments, info


def header_block(data_shape, bitpix, header=None, comments=None, extra=()):
    """The padded header bytes ``write`` would emit for an array of this shape / BITPIX."""
    header = dict(header or {})
    comments = comments or {}
    cards = [_card('SIMPLE', True, 'conforms to FITS standard'),
             _card('BITPIX', bitpix, 'array data type'),
             _card('NAXIS', len(data_shape), 'number of array dimensions')]
    for i, n in enumerate(reversed(tuple(data_shape))):
        cards.append(_card(f'NAXIS{i + 1}', int(n)))
    skip = set(_MANDATORY) | {'BSCALE', 'BZERO', 'END'} | {f'NAXIS{i}' for i in range(1, 10)}
    for k, v, c in extra:
        cards.append(_card(k, v, c))
    for k, v in header.items():
        ku = str(k).upper()
        if ku in skip or v is None:
            continue
        if not isinstance(v, (int, float, str, bool, np.integer, np.floating, np.bool_)):
            continue
        cards.append(_card(ku, v, comments.get(k, '') or ''))
    cards.append('END'.ljust(80))
    hdr = ''.join(cards).encode('ascii', 'replace')
    return hdr + b' ' * (-len(hdr) % BLOCK)


def write_raw(path, raw, data_shape, bitpix, header=None, comments=None):
    """Write a primary HDU from an already encoded (big-endian) data block."""
    raw = memoryview(raw)
    with open(path, 'wb') as f:
        f.write(header_block(data_shape, bitpix, header, comments))
        f.write(raw)
        f.write(b'\0' * (-raw.nbytes % BLOCK))


def _fmt_value(v):
    if isinstance(v, (bool, np.bool_)):
        return f"{'T' if v else 'F':>20}"
    if isinstance(v, (int, np.integer)):
        return f'{int(v):>20d}'
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if not np.isfinite(v):
            return f"{repr(str(v)):<20}"
        s = repr(v).upper()
        if 'E' not in s and '.' not in s:
            s += '.0'
        if len(s) > 20:
            s = f'{v:.14E}'
        return f'{s:>20}'
    s = str(v).replace("'", "''")
    return f"'{s:<8}'"


def _card(key, value, comment=''):
    key = str(key).upper()
    if len(key) > 8:
        body = f'HIERARCH {key} = {_fmt_value(value).strip()}'
    else:
        body = f'{key:<8}= {_fmt_value(value)}'
    if comment:
        body += f' / {comment}'
    return body[:80].ljust(80)


def write(path, data, header=None, comments=None):
    """Write a primary-HDU image (overwrites).  ``header`` values that describe
    the array (SIMPLE / BITPIX / NAXIS* / BSCALE / BZERO) are regenerated."""
    header = dict(header or {})
    comments = comments or {}
    data = np.asarray(data)
    extra = []
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    kind = data.dtype
    if kind == np.uint8:
        bitpix = 8
    elif kind == np.int16:
        bitpix = 16
    elif kind == np.uint16:
        bitpix = 16
        data = (data.astype(np.int32) - 32768).astype(np.int16)
        extra = [('BSCALE', 1, ''), ('BZERO', 32768, '')]
    elif kind == np.int32:
        bitpix = 32
    elif kind in (np.int64, np.uint32, np.uint64):
        bitpix = 64
        data = data.astype(np.int64)
    elif kind == np.float32:
        bitpix = -32
    elif kind in (np.float64, np.float16):
        bitpix = -64
        data = data.astype(np.float64)
    elif kind == np.int8:
        bitpix = 16
        data = data.astype(np.int16)
    else:
        raise ValueError(f'cannot write dtype {kind} to FITS')
    hdr = header_block(data.shape, bitpix, header, comments, extra)
    raw = np.ascontiguousarray(data).astype(np.dtype(_BITPIX_DTYPE[bitpix])).tobytes()
    with open(path, 'wb') as f:
        f.write(hdr)
        f.write(raw)
        f.write(b'\0' * (-len(raw) % BLOCK))
