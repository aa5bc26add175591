"""JFIF container around the entropy-coded row segments of mdct_huffman_rows (host side, numpy only).

The device produces, per component plane, one unstuffed byte-aligned Huffman segment per block row
(restart interval = one block row, ITU-T T.81 E.1.4).  This module does what remains to obtain a
baseline JPEG any decoder opens: byte stuffing (B.1.1.5), RSTm markers between the rows, and the
marker segments (B.2: SOI, APP0/JFIF, DQT, SOF0, DHT, DRI, SOS, EOI).  One non-interleaved scan per
component, so a 4:2:0 picture is three scans (Y at full resolution, Cb / Cr at half).
No reference counterpart (the reference stops at the coefficient reorder, simd_dct.cpp:2221-2230).
"""
import struct

import numpy as np

from . import api


def _seg(marker, payload):
    return b"\xff" + bytes([marker]) + struct.pack(">H", len(payload) + 2) + payload


def _dht(table_class, table_id, bits, vals):
    return bytes([(table_class << 4) | table_id]) + bytes(bits) + bytes(vals)


def stuff(segment):
    """B.1.1.5: a zero byte after every 0xFF of entropy-coded data"""
    a = np.frombuffer(segment, dtype=np.uint8) if not isinstance(segment, np.ndarray) else segment
    ff = np.flatnonzero(a == 0xFF)
    if ff.size == 0:
        return a.tobytes()
    return np.insert(a, ff + 1, 0).tobytes()


def scan_bytes(segments, seg_bytes, seg_stride):
    """rows of one component -> stuffed entropy-coded data with RST0..RST7 between the rows"""
    out = []
    n = len(seg_bytes)
    for r in range(n):
        out.append(stuff(segments[r * seg_stride:r * seg_stride + int(seg_bytes[r])]))
        if r + 1 < n:
            out.append(b"\xff" + bytes([0xD0 + (r & 7)]))
    return b"".join(out)


def write_jpeg(components, width, height, specs=None):
    """components: list of 1 (grey) or 3 (Y, Cb, Cr with Cb/Cr at half resolution) dicts with keys
         'blocks_per_row', 'qtable' (64 integers 1..255, natural order v*8+u) and either
         'scan' (bytes / uint8 array: the stuffed, RST-delimited scan as mdct_jpeg_pack_rows leaves it) or
         'segments' (uint8 array / bytes), 'seg_bytes' (per block row), 'seg_stride' (stuffed and joined here)
       specs: {which: (bits16, vals)} Huffman specifications (default: the library's, api.huffman_spec)
       Returns the file as bytes."""
    specs = specs or {w: api.huffman_spec(w) for w in range(4)}
    zz = api.zigzag_table()
    nc = len(components)
    assert nc in (1, 3)
    f = [b"\xff\xd8", _seg(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")]
    qts = [components[0]["qtable"]] + ([components[1]["qtable"]] if nc == 3 else [])
    for tq, q in enumerate(qts):
        q = np.asarray(q).reshape(64)
        assert np.all(q == np.rint(q)) and q.min() >= 1 and q.max() <= 255, "baseline DQT holds 8-bit integers"
        f.append(_seg(0xDB, bytes([tq]) + bytes(int(q[zz[k]]) for k in range(64))))
    sof = struct.pack(">BHHB", 8, height, width, nc)
    if nc == 1:
        sof += bytes([1, 0x11, 0])
    else:
        sof += bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    f.append(_seg(0xC0, sof))
    f.append(_seg(0xC4, _dht(0, 0, *specs[0]) + _dht(1, 0, *specs[1]) + (_dht(0, 1, *specs[2]) + _dht(1, 1, *specs[3]) if nc == 3 else b"")))
    for ci, c in enumerate(components):
        f.append(_seg(0xDD, struct.pack(">H", c["blocks_per_row"])))
        th = 0 if ci == 0 else 1
        f.append(_seg(0xDA, bytes([1, ci + 1, (th << 4) | th, 0, 63, 0])))
        f.append(bytes(memoryview(np.ascontiguousarray(c["scan"]))) if "scan" in c else scan_bytes(c["segments"], c["seg_bytes"], c["seg_stride"]))
    f.append(b"\xff\xd9")
    return b"".join(f)
