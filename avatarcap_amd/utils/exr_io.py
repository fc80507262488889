"""Minimal OpenEXR scanline reader for the reference's image-normal inputs (`cv.imread(<...>.exr, cv.IMREAD_UNCHANGED)`,
main.py:408-410, and the EXR position / normal maps of gen_data) -- SURVEY.md section 8(f) item 4.  OpenCV is not part
of this build; this module follows the published OpenEXR file layout for single-part scanline images with NONE, RLE,
ZIPS or ZIP compression and HALF / FLOAT / UINT channels (what OpenCV's writer produces).  Tiled, deep, multi-part
files and the PIZ / PXR24 / B44 / DWA codecs are refused with a clear error.  Pinned on a file written by the OpenEXR
library with independently known pixel values (tests/golden/openexr_sample.exr + .ppm, tests/test_real_data_seam.py); tests/test_host.py
also round-trips files produced by an independent writer of the same layout.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL = {0: np.dtype('<u4'), 1: np.dtype('<f2'), 2: np.dtype('<f4')}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16}                      # NONE, RLE, ZIPS, ZIP
_CODEC = {4: 'PIZ', 5: 'PXR24', 6: 'B44', 7: 'B44A', 8: 'DWAA', 9: 'DWAB'}


def _cstr(buf, pos):
    end = buf.index(b'\0', pos)
    return buf[pos:end].decode('latin-1'), end + 1


def _unpredict(raw: bytes) -> bytes:
    """Undo OpenEXR's byte-delta predictor and the even/odd byte split used by the RLE and ZIP codecs."""
    t = np.frombuffer(raw, np.uint8).astype(np.int64)
    if t.size:
        t[1:] -= 128
        t = np.cumsum(t) & 0xff
    t = t.astype(np.uint8)
    half = (t.size + 1) // 2
    out = np.empty(t.size, np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _unrle(data: bytes, expected: int) -> bytes:
    out, i = bytearray(), 0
    while i < len(data) and len(out) < expected:
        c = data[i] - 256 if data[i] > 127 else data[i]
        i += 1
        if c < 0:
            out += data[i:i - c]; i += -c
        else:
            out += bytes([data[i]]) * (c + 1); i += 1
    return bytes(out)


def read_exr(path, order='BGR'):
    """-> float32 array (H, W, C).  Channels named R, G, B (and A) come back in `order` ('BGR' = what cv.imread returns,
    alpha last); any other channel set comes back in the file's (alphabetical) order.  HALF and UINT are converted to float32."""
    buf = open(path, 'rb').read()
    magic, version = struct.unpack_from('<ii', buf, 0)
    if magic != _MAGIC:
        raise ValueError('%s is not an OpenEXR file' % path)
    if version & 0xff != 2 or version & (0x200 | 0x800 | 0x1000):
        raise NotImplementedError('%s: only single-part scanline OpenEXR version 2 files are supported (tiled / deep / multi-part: no)' % path)
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size, = struct.unpack_from('<i', buf, pos); pos += 4
        attrs[name] = (typ, buf[pos:pos + size]); pos += size
    pos += 1
    chans, cb, p = [], attrs['channels'][1], 0
    while cb[p] != 0:
        name, p = _cstr(cb, p)
        ptype, _lin, xs, ys = struct.unpack_from('<iB3xii', cb, p); p += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError('%s: sub-sampled channel %s' % (path, name))
        chans.append((name, _PIXEL[ptype]))
    comp = attrs['compression'][1][0]
    if comp not in _LINES:
        raise NotImplementedError('%s: %s compression is not supported (NONE, RLE, ZIPS, ZIP are)' % (path, _CODEC.get(comp, comp)))
    x0, y0, x1, y1 = struct.unpack('<4i', attrs['dataWindow'][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines = _LINES[comp]
    nblk = (H + lines - 1) // lines
    offsets = struct.unpack_from('<%dQ' % nblk, buf, pos)
    row_bytes = sum(dt.itemsize for _, dt in chans) * W
    planes = {name: np.empty((H, W), np.float32) for name, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from('<ii', buf, off)
        data = buf[off + 8:off + 8 + size]
        n = min(lines, y1 - y + 1)
        want = n * row_bytes
        if comp != 0 and size < want:
            data = _unpredict(zlib.decompress(data) if comp in (2, 3) else _unrle(data, want))
        if len(data) != want:
            raise ValueError('%s: corrupt chunk at scanline %d' % (path, y))
        p = 0
        for r in range(n):
            for name, dt in chans:
                planes[name][y - y0 + r] = np.frombuffer(data, dt, W, p).astype(np.float32)
                p += dt.itemsize * W
    names = [n for n, _ in chans]
    if set('RGB') <= set(names) and set(names) <= set('RGBA'):
        names = [c for c in order if c in names] + (['A'] if 'A' in names else [])
    return np.stack([planes[n] for n in names], -1)
