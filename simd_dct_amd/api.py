"""Host-side mirror of the reference's operator interface, on device tensors.

Two levels, both thin wrappers over the C-ABI (include/mdct.h):

* the reference's own three functions, same names / argument order / result codes
  (simd_dct.h:22-31) -- ``simdDCT_EncodeQuantize32ReorderBuffer`` etc.  They accept torch
  uint8 tensors on ``cuda`` (zero-copy) or on the CPU / numpy arrays (staged by the shim).
* the engine's native entry points with half-open block-row ranges and pitches.

torch is used for device memory and streams only.
"""
import ctypes

import numpy as np

from . import _lib

sdr_Success, sdr_InvalidParameter, sdr_NotSupported = 0, 1, 2  # simd_dct.h:22-27

PROFILE_REF_AVX, PROFILE_REF_SSE, PROFILE_REF_SCALAR = 0, 1, 2
LAYOUT_Q32, LAYOUT_STEREO, LAYOUT_BLOCK, LAYOUT_BLOCK_SSE = 0, 1, 2, 3

# main.cpp:179-189, the harness's base quantisation table (data, index v*8+u)
QUANTIZE_BASE = np.array(
    [.17, .11, .10, .16, .24, .40, .51, .61, .12, .12, .14, .19, .26, .58, .60, .55,
     .14, .13, .16, .24, .40, .57, .69, .56, .14, .17, .22, .29, .51, .87, .80, .62,
     .18, .22, .37, .56, .68, 1.09, 1.03, .77, .24, .35, .55, .64, .81, 1.04, 1.13, .92,
     .49, .64, .78, .87, 1.03, 1.21, 1.20, 1.01, .72, .92, .95, .98, 1.12, 1.00, 1.03, .99],
    dtype=np.float32)


class MdctError(RuntimeError):
    pass


def last_error():
    return _lib.load().mdct_last_error().decode()


def _check(rc):
    if rc != 0:
        raise MdctError(f"mdct status {rc}: {last_error()}")


def _lut_ptr(lut):
    if lut is None:
        return None, None
    a = np.ascontiguousarray(np.asarray(lut, dtype=np.float32).reshape(64))
    return a, a.ctypes.data_as(_lib.f32p)


def _ptr(t):
    """address of a torch tensor (any device) or numpy array"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


def _stream(stream=None):
    if stream is not None:  # a raw hipStream_t or a torch.cuda.Stream
        return ctypes.c_void_p(int(getattr(stream, "cuda_stream", stream)))
    import torch

    if torch.cuda.is_available():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def init(device=0):
    """mdct_init: replaces _DetectCPUFeatures() (simd_platform.c:68)."""
    _check(_lib.load().mdct_init(int(device)))


def device_info():
    info = _lib.DeviceInfo()
    _check(_lib.load().mdct_get_device_info(ctypes.byref(info)))
    return {
        "device": info.device, "compute_units": info.compute_units, "wavefront_size": info.wavefront_size,
        "lds_bytes_per_cu": info.lds_bytes_per_cu, "is_gfx950": bool(info.is_gfx950),
        "hbm_bytes": info.hbm_bytes, "name": info.name.decode(),
    }


# ----------------------------------------------------------------------------- reference API
def _ref_call(which, pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY):
    """Device tensors: the kernel runs on torch's CURRENT stream (ordered against the work that
    produced the tensors) and the call returns when it has finished, like the reference's
    blocking functions.  Host arrays: the shim's synchronous staging pipeline."""
    keep, lp = _lut_ptr(pQuantizeLUT)
    return _lib.load().mdct_shim_call_on(which, _ptr(pFrom), _ptr(pTo), lp, sizeX, sizeY, startY, endY, _stream(), 0)


def simdDCT_EncodeQuantize32ReorderBuffer(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY):
    """simd_dct.h:31 / simd_dct.cpp:113-133.  Returns the simdDctResult value."""
    return _ref_call(0, pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY)


def simdDCT_EncodeQuantizeReorderStereoBuffer(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY):
    """simd_dct.h:30 / simd_dct.cpp:71-91."""
    return _ref_call(1, pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY)


def simdDCT_EncodeQuantizeBuffer(pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY):
    """simd_dct.h:29 / simd_dct.cpp:93-111."""
    return _ref_call(2, pFrom, pTo, pQuantizeLUT, sizeX, sizeY, startY, endY)


SIMD_NONE, SIMD_SSE2, SIMD_SSSE3, SIMD_SSE41, SIMD_AVX2 = 0, 1, 2, 3, 4  # include/simd_dct_shim.h


def set_max_simd(level):
    """Counterpart of `--max-simd` (main.cpp:283-438): SIMD_NONE .. SIMD_AVX2 (default), see
    include/simd_dct_shim.h for the tier each level selects per function; negative = unset."""
    _lib.load().mdct_shim_set_max_simd(int(level))


def get_max_simd():
    return _lib.load().mdct_shim_get_max_simd()


# -------------------------------------------------------------------------------- native API
def fwd_quant_u8(src, dst, lut, sizeX, sizeY, by0, by1, layout=LAYOUT_Q32, profile=PROFILE_REF_AVX, pitch_in=None, pitch_out=None, stream=None, check=True):
    """pitch_out (bytes between the output strips of consecutive block rows) selects mdct_fwd_quant_u8_pitched"""
    keep, lp = _lut_ptr(lut)
    if pitch_out is None:
        rc = _lib.load().mdct_fwd_quant_u8(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, lp, sizeX, sizeY, by0, by1, layout, profile, _stream(stream))
    else:
        rc = _lib.load().mdct_fwd_quant_u8_pitched(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, pitch_out, lp, sizeX, sizeY, by0, by1, layout, profile, _stream(stream))
    if check:
        _check(rc)
    return rc


def _plane(fn, src, dst, lut, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check):
    keep, lp = _lut_ptr(lut)
    by1 = sizeY // 8 if by1 is None else by1
    rc = fn(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, sizeX if pitch_out is None else pitch_out, lp, sizeX, sizeY, by0, by1, _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_i16(src, dst, sizeX, sizeY, lut=None, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    return _plane(_lib.load().mdct_fwd_i16, src, dst, lut, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check)


def inv_i16(src, dst, sizeX, sizeY, lut=None, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    return _plane(_lib.load().mdct_inv_i16, src, dst, lut, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check)


def roundtrip_i16(src, dst, sizeX, sizeY, lut=None, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    return _plane(_lib.load().mdct_roundtrip_i16, src, dst, lut, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check)


def fwd_u8_i16(src, dst, sizeX, sizeY, lut=None, level_shift=True, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    """8-bit pixels -> int16 coefficients (mdct_fwd_u8_i16)"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_u8_i16(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, sizeX if pitch_out is None else pitch_out, lp, int(bool(level_shift)),
                                     sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_u8_records(src, sizeX, sizeY, levels, runs, counts, lut=None, level_shift=True, by0=0, by1=None, pitch=None, stream=None, check=True):
    """8-bit pixels -> zig-zag run/level records of the quantised coefficients in one pass (mdct_fwd_u8_records)"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_u8_records(_ptr(src), sizeX if pitch is None else pitch, lp, int(bool(level_shift)), sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1,
                                         _ptr(levels), _ptr(runs), _ptr(counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_i16_records(src, sizeX, sizeY, levels, runs, counts, lut=None, by0=0, by1=None, pitch=None, stream=None, check=True):
    """int16 plane -> zig-zag run/level records of its quantised coefficients in one pass (mdct_fwd_i16_records)"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_i16_records(_ptr(src), sizeX if pitch is None else pitch, lp, sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, _ptr(levels), _ptr(runs), _ptr(counts),
                                          _stream(stream))
    if check:
        _check(rc)
    return rc


def inv_i16_u8(src, dst, sizeX, sizeY, lut=None, level_shift=True, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    """int16 coefficients -> 8-bit pixels (mdct_inv_i16_u8)"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_inv_i16_u8(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, sizeX if pitch_out is None else pitch_out, lp, int(bool(level_shift)),
                                     sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, _stream(stream))
    if check:
        _check(rc)
    return rc


def _plane_f32(fn, src, dst, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check):
    by1 = sizeY // 8 if by1 is None else by1
    rc = fn(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, sizeX if pitch_out is None else pitch_out, sizeX, sizeY, by0, by1, _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_f32(src, dst, sizeX, sizeY, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    return _plane_f32(_lib.load().mdct_fwd_f32, src, dst, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check)


def inv_f32(src, dst, sizeX, sizeY, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    return _plane_f32(_lib.load().mdct_inv_f32, src, dst, sizeX, sizeY, by0, by1, pitch_in, pitch_out, stream, check)


def _plane_array(planes):
    """planes: list of (src, dst, sizeX, sizeY, lut-or-None[, pitch_in, pitch_out]) -> (mdct_plane_i16 array, objects to keep alive)"""
    arr = (_lib.PlaneI16 * max(len(planes), 1))()
    keep = []
    for i, pl in enumerate(planes):
        src, dst, sx, sy, lut = pl[:5]
        pin = pl[5] if len(pl) > 5 and pl[5] is not None else sx
        pout = pl[6] if len(pl) > 6 and pl[6] is not None else sx
        k, lp = _lut_ptr(lut)
        keep.append((k, src, dst))
        arr[i] = _lib.PlaneI16(_ptr(src), _ptr(dst), pin, pout, sx, sy, lp)
    return arr, keep


def roundtrip_i16_planes(planes, stream=None, check=True):
    """planes: list of (src, dst, sizeX, sizeY, lut-or-None); the call BASELINE.json configs[2] is measured on."""
    arr, keep = _plane_array(planes)
    rc = _lib.load().mdct_roundtrip_i16_planes(arr, len(planes), _stream(stream))
    if check:
        _check(rc)
    return rc


MODES = {"fwd": 0, "inv": 1, "roundtrip": 2}  # MDCT_MODE_*


def i16_batch(mode, planes, stream=None, check=True):
    """mdct_{fwd,inv,roundtrip}_i16_batch: any number of separately allocated planes, descriptors in the kernel
    arguments (no allocation; as many planes per launch as fit).  planes as for _plane_array."""
    lib = _lib.load()
    fn = {"fwd": lib.mdct_fwd_i16_batch, "inv": lib.mdct_inv_i16_batch, "roundtrip": lib.mdct_roundtrip_i16_batch}[mode]
    arr, keep = _plane_array(planes)
    rc = fn(arr, len(planes), _stream(stream))
    if check:
        _check(rc)
    return rc


def roundtrip_u8(src, dst, sizeX, sizeY, lut=None, level_shift=True, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None, check=True):
    """8-bit pixels -> forward, quantise, dequantise, inverse -> 8-bit pixels in one pass (mdct_roundtrip_u8); pitches in bytes"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_roundtrip_u8(_ptr(src), _ptr(dst), sizeX if pitch_in is None else pitch_in, sizeX if pitch_out is None else pitch_out, lp, int(bool(level_shift)),
                                       sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, _stream(stream))
    if check:
        _check(rc)
    return rc


def roundtrip_u8_batch(planes, level_shift=True, stream=None, check=True):
    """mdct_roundtrip_u8_batch: planes as for _plane_array (uint8 tensors, pitches in bytes); the call BASELINE.json configs[2]
    (Y + Cb + Cr with per-plane tables, 8-bit planes in and out) is measured on."""
    arr, keep = _plane_array(planes)
    rc = _lib.load().mdct_roundtrip_u8_batch(arr, len(planes), int(bool(level_shift)), _stream(stream))
    if check:
        _check(rc)
    return rc


def _q32_planes(planes):
    """(src, dst, sizeX, sizeY, lut[, pitch_in, strip pitch]): the output strip pitch defaults to the reference's tight 8 * sizeX"""
    return [(p[0], p[1], p[2], p[3], p[4], p[5] if len(p) > 5 else None, p[6] if len(p) > 6 and p[6] is not None else 8 * p[2]) for p in planes]


def fwd_quant32_u8_batch(planes, stream=None, check=True):
    """mdct_fwd_quant32_u8_batch: the reference's q32 product (every block row) of a list of 8-bit planes, one launch;
    planes = list of (src uint8, dst uint8, sizeX, sizeY, lut[, pitch_in bytes, output strip pitch bytes])"""
    arr, keep = _plane_array(_q32_planes(planes))
    rc = _lib.load().mdct_fwd_quant32_u8_batch(arr, len(planes), _stream(stream))
    if check:
        _check(rc)
    return rc


def prepare_fwd_quant32_u8_batch(planes, stream=None):
    arr, keep = _plane_array(_q32_planes(planes))
    return Prepared(_lib.load().mdct_fwd_quant32_u8_batch, (arr, ctypes.c_int(len(planes)), _stream(stream)), keep)


def u8_i16_batch(mode, planes, level_shift=True, stream=None, check=True):
    """mdct_fwd_u8_i16_batch ('fwd') / mdct_inv_i16_u8_batch ('inv'): planes = list of (px uint8, coef int16, sizeX, sizeY, lut-or-None[, pitch_px bytes, pitch_coef elements])"""
    lib = _lib.load()
    arr, keep = _plane_array(planes)
    rc = (lib.mdct_fwd_u8_i16_batch if mode == "fwd" else lib.mdct_inv_i16_u8_batch)(arr, len(planes), int(bool(level_shift)), _stream(stream))
    if check:
        _check(rc)
    return rc


class Batch:
    """mdct_batch: descriptors and tables uploaded once, every run ONE launch (capture-safe).
    mode 'roundtrip_u8': 8-bit planes (mdct_batch_create_u8); 'fwd_u8_i16' / 'inv_i16_u8': (px, coef, ...) planes (mdct_batch_create_u8_i16);
    'q32': 8-bit planes -> the reference's q32 product (mdct_batch_create_q32; pitch_out = output strip pitch, default 8 * sizeX);
    otherwise int16 planes."""

    def __init__(self, mode, planes, level_shift=True):
        lib = _lib.load()
        arr, self._keep = _plane_array(_q32_planes(planes) if mode == "q32" else planes)
        h = ctypes.c_void_p()
        if mode == "q32":
            _check(lib.mdct_batch_create_q32(ctypes.byref(h), arr, len(planes)))
        elif mode in ("fwd_u8_i16", "inv_i16_u8"):
            _check(lib.mdct_batch_create_u8_i16(ctypes.byref(h), MODES["fwd" if mode == "fwd_u8_i16" else "inv"], arr, len(planes), int(bool(level_shift))))
        elif mode == "roundtrip_u8":
            _check(lib.mdct_batch_create_u8(ctypes.byref(h), arr, len(planes), int(bool(level_shift))))
        else:
            _check(lib.mdct_batch_create(ctypes.byref(h), MODES[mode], arr, len(planes)))
        self._h = h
        self.launches = lib.mdct_batch_launches(h)

    def run(self, stream=None, check=True):
        rc = _lib.load().mdct_batch_run(self._h, _stream(stream))
        if check:
            _check(rc)
        return rc

    def prepared(self, stream=None):
        return Prepared(_lib.load().mdct_batch_run, (self._h, _stream(stream)), self)

    def close(self):
        if self._h:
            _lib.load().mdct_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------- stages either side of the transform
def zigzag_table():
    """scan position k -> natural index v*8+u (ITU-T T.81 Figure A.6)"""
    zz = np.zeros(64, dtype=np.uint8)
    _lib.load().mdct_zigzag_table(zz.ctypes.data)
    return zz


def zigzag_rle_i16(coef, sizeX, sizeY, levels, runs=None, counts=None, by0=0, by1=None, pitch=None, stream=None, check=True):
    """zig-zag scan (+ run/level pairs when runs/counts are given) of an int16 coefficient plane"""
    rc = _lib.load().mdct_zigzag_rle_i16(_ptr(coef), sizeX if pitch is None else pitch, sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1,
                                         _ptr(levels), _ptr(runs), _ptr(counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def zigzag_rle_q32(q32, sizeX, sizeY, levels, runs=None, counts=None, by0=0, by1=None, stream=None, check=True):
    """the same from the reference's q32 byte layout (level = byte - 127)"""
    rc = _lib.load().mdct_zigzag_rle_q32(_ptr(q32), sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, _ptr(levels), _ptr(runs), _ptr(counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def zigzag_rle_u8(coef, layout, sizeX, sizeY, levels, runs=None, counts=None, by0=0, by1=None, stream=None, check=True):
    """the same from the q32, stereo (64 coefficient planes) or scalar-encq block layout"""
    if by1 is None:
        by1 = sizeY // (16 if layout == LAYOUT_STEREO else 8)
    rc = _lib.load().mdct_zigzag_rle_u8(_ptr(coef), layout, sizeX, sizeY, by0, by1, _ptr(levels), _ptr(runs), _ptr(counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def huffman_spec(which):
    """(BITS[16], HUFFVAL[...]) of the library's Huffman table `which` (0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma)"""
    bits = np.zeros(16, dtype=np.uint8)
    vals = np.zeros(256, dtype=np.uint8)
    n = ctypes.c_int()
    _check(_lib.load().mdct_huffman_spec(which, bits.ctypes.data, vals.ctypes.data, ctypes.byref(n)))
    return bits.tolist(), vals[:n.value].tolist()


def huffman_seg_stride(sizeX):
    """smallest legal segment stride for mdct_huffman_rows on a plane of this width"""
    return int(_lib.load().mdct_huffman_seg_stride(sizeX))


def huffman_rows(levels, runs, counts, sizeX, sizeY, out, seg_bytes, seg_stride=None, chroma=False, by0=0, by1=None, stream=None, check=True):
    """baseline Huffman coding of the run/level records: one unstuffed segment per block row (mdct_huffman_rows)"""
    rc = _lib.load().mdct_huffman_rows(_ptr(levels), _ptr(runs), _ptr(counts), sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, int(bool(chroma)),
                                       _ptr(out), huffman_seg_stride(sizeX) if seg_stride is None else seg_stride, _ptr(seg_bytes), _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_u8_huffman_rows(src, sizeX, sizeY, out, seg_bytes, lut=None, level_shift=True, seg_stride=None, chroma=False, by0=0, by1=None, pitch=None, ff_counts=None, stream=None, check=True):
    """8-bit pixels -> Huffman row segments in one kernel (mdct_fwd_u8_huffman_rows): the bytes of fwd_u8_records + huffman_rows"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_u8_huffman_rows(_ptr(src), sizeX if pitch is None else pitch, lp, int(bool(level_shift)), sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, int(bool(chroma)),
                                              _ptr(out), huffman_seg_stride(sizeX) if seg_stride is None else seg_stride, _ptr(seg_bytes), _ptr(ff_counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_i16_huffman_rows(src, sizeX, sizeY, out, seg_bytes, lut=None, seg_stride=None, chroma=False, by0=0, by1=None, pitch=None, ff_counts=None, stream=None, check=True):
    """int16 plane -> Huffman row segments in one kernel (mdct_fwd_i16_huffman_rows): the bytes of fwd_i16_records + huffman_rows"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_i16_huffman_rows(_ptr(src), sizeX if pitch is None else pitch, lp, sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, int(bool(chroma)),
                                               _ptr(out), huffman_seg_stride(sizeX) if seg_stride is None else seg_stride, _ptr(seg_bytes), _ptr(ff_counts), _stream(stream))
    if check:
        _check(rc)
    return rc


def _scan_cap(out, out_capacity):
    if out_capacity is not None:
        return out_capacity
    return out.numel() * out.element_size() if hasattr(out, "numel") else out.nbytes


def fwd_u8_jpeg_scan(src, sizeX, sizeY, seg_work, row_work, out, row_offsets, lut=None, level_shift=True, seg_stride=None, chroma=False, by0=0, by1=None, pitch=None, first_rst=0,
                     out_capacity=None, stream=None, check=True):
    """8-bit pixels -> finished, stuffed scan with RSTm between the block rows in ONE launch (mdct_fwd_u8_jpeg_scan).
    seg_work: by1 * seg_stride bytes of scratch; row_work: by1 - by0 + 2 int64 zeroed once by the caller; row_offsets: by1 - by0 + 1 int64"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_u8_jpeg_scan(_ptr(src), sizeX if pitch is None else pitch, lp, int(bool(level_shift)), sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, int(bool(chroma)),
                                           _ptr(seg_work), huffman_seg_stride(sizeX) if seg_stride is None else seg_stride, _ptr(row_work), first_rst, _ptr(out), _scan_cap(out, out_capacity),
                                           _ptr(row_offsets), _stream(stream))
    if check:
        _check(rc)
    return rc


def fwd_i16_jpeg_scan(src, sizeX, sizeY, seg_work, row_work, out, row_offsets, lut=None, seg_stride=None, chroma=False, by0=0, by1=None, pitch=None, first_rst=0, out_capacity=None,
                      stream=None, check=True):
    """int16 plane -> finished scan in ONE launch (mdct_fwd_i16_jpeg_scan); arguments as fwd_u8_jpeg_scan"""
    keep, lp = _lut_ptr(lut)
    rc = _lib.load().mdct_fwd_i16_jpeg_scan(_ptr(src), sizeX if pitch is None else pitch, lp, sizeX, sizeY, by0, sizeY // 8 if by1 is None else by1, int(bool(chroma)),
                                            _ptr(seg_work), huffman_seg_stride(sizeX) if seg_stride is None else seg_stride, _ptr(row_work), first_rst, _ptr(out), _scan_cap(out, out_capacity),
                                            _ptr(row_offsets), _stream(stream))
    if check:
        _check(rc)
    return rc


def jpeg_pack_rows(segments, seg_bytes, seg_stride, n_rows, out, row_offsets, first_rst=0, out_capacity=None, ff_counts=None, stream=None, check=True):
    """row segments of huffman_rows -> one stuffed scan with RSTm between the rows (mdct_jpeg_pack_rows); row_offsets: n_rows + 1 int64.
    ff_counts (from fwd_*_huffman_rows): mdct_jpeg_pack_rows_counted, no counting pass"""
    cap = out.numel() * out.element_size() if out_capacity is None and hasattr(out, "numel") else (out.nbytes if out_capacity is None else out_capacity)
    if ff_counts is not None:
        rc = _lib.load().mdct_jpeg_pack_rows_counted(_ptr(segments), _ptr(seg_bytes), _ptr(ff_counts), seg_stride, n_rows, first_rst, _ptr(out), cap, _ptr(row_offsets), _stream(stream))
    else:
        rc = _lib.load().mdct_jpeg_pack_rows(_ptr(segments), _ptr(seg_bytes), seg_stride, n_rows, first_rst, _ptr(out), cap, _ptr(row_offsets), _stream(stream))
    if check:
        _check(rc)
    return rc


def split420_u8(ycc, sizeX, sizeY, y, cb, cr, pitch=None, pitch_y=None, pitch_c=None, stream=None, check=True):
    """interleaved 8-bit Y Cb Cr -> level-shifted int16 Y (full) and Cb / Cr (2x2 box average) planes"""
    rc = _lib.load().mdct_split420_u8(_ptr(ycc), 3 * sizeX if pitch is None else pitch, sizeX, sizeY, _ptr(y), _ptr(cb), _ptr(cr),
                                      sizeX if pitch_y is None else pitch_y, sizeX // 2 if pitch_c is None else pitch_c, _stream(stream))
    if check:
        _check(rc)
    return rc


def split420_u8_planes(ycc, sizeX, sizeY, y, cb, cr, pitch=None, pitch_y=None, pitch_c=None, stream=None, check=True):
    """interleaved 8-bit Y Cb Cr -> 8-bit Y (full) and Cb / Cr (2x2 box average) planes, not level-shifted: the input of the 8-bit plane batches"""
    rc = _lib.load().mdct_split420_u8_planes(_ptr(ycc), 3 * sizeX if pitch is None else pitch, sizeX, sizeY, _ptr(y), _ptr(cb), _ptr(cr),
                                             sizeX if pitch_y is None else pitch_y, sizeX // 2 if pitch_c is None else pitch_c, _stream(stream))
    if check:
        _check(rc)
    return rc


# ------------------------------------------------------------------------- multi-GPU (RCCL via the C-ABI)
UNIQUE_ID_BYTES = 128


def comm_unique_id():
    """rank 0: 128 opaque bytes to hand to the other ranks (mdct_comm_get_unique_id)"""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _check(_lib.load().mdct_comm_get_unique_id(buf))
    return buf.raw


class Comm:
    """mdct_comm: one per process / GPU; collective construction over all ranks"""

    def __init__(self, rank, world, unique_id):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        ident = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _check(self._lib.mdct_comm_init(ctypes.byref(self._h), int(rank), int(world), ident))
        self.rank, self.world = rank, world

    def allgather_rows(self, buf, row_bytes, n_rows, stream=None):
        _check(self._lib.mdct_allgather_rows(self._h, _ptr(buf), row_bytes, n_rows, _stream(stream)))

    def allgather_stereo(self, buf, sizeX, sizeY, stream=None):
        _check(self._lib.mdct_allgather_stereo(self._h, _ptr(buf), sizeX, sizeY, _stream(stream)))

    def close(self):
        if self._h:
            self._lib.mdct_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_rows_c(n_rows, world, rank):
    """mdct_shard_rows (the C-ABI's own arithmetic; equals sharding.shard_rows)"""
    b0, b1 = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.load().mdct_shard_rows(n_rows, world, rank, ctypes.byref(b0), ctypes.byref(b1))
    return b0.value, b1.value


def stereo_shard_piece(sizeX, sizeY, world, rank):
    """(first_offset, plane_stride, piece_bytes) of rank's shard in the stereo layout"""
    a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    _check(_lib.load().mdct_stereo_shard_piece(sizeX, sizeY, world, rank, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return a.value, b.value, c.value


def stream_copy(src, dst, nbytes, stream=None):
    _check(_lib.load().mdct_stream_copy(_ptr(src), _ptr(dst), nbytes, _stream(stream)))


class Prepared:
    """A launch with its ctypes arguments marshalled once: calling it costs one foreign call
    (~1 us) instead of re-deriving pointers, table and stream (~10 us), which matters when
    the kernel itself runs for ~45 us.  Holds references to the tensors it was built from."""

    def __init__(self, fn, args, keep):
        self._fn, self._args, self._keep = fn, args, keep

    def __call__(self):
        rc = self._fn(*self._args)
        if rc != 0:
            _check(rc)


def _c(v, t):
    return v if v is None else t(v)


def prepare_plane_i16(mode, src, dst, sizeX, sizeY, lut=None, by0=0, by1=None, pitch_in=None, pitch_out=None, stream=None):
    """mode: 'fwd' | 'inv' | 'roundtrip' -> Prepared launch of mdct_{mode}_i16"""
    lib = _lib.load()
    fn = {"fwd": lib.mdct_fwd_i16, "inv": lib.mdct_inv_i16, "roundtrip": lib.mdct_roundtrip_i16}[mode]
    keep, lp = _lut_ptr(lut)
    sz = ctypes.c_size_t
    args = (ctypes.c_void_p(_ptr(src)), ctypes.c_void_p(_ptr(dst)), sz(sizeX if pitch_in is None else pitch_in), sz(sizeX if pitch_out is None else pitch_out),
            lp, sz(sizeX), sz(sizeY), sz(by0), sz(sizeY // 8 if by1 is None else by1), _stream(stream))
    return Prepared(fn, args, (keep, src, dst))


def prepare_u8_i16(mode, src, dst, sizeX, sizeY, lut=None, level_shift=True, stream=None):
    """mode 'fwd' (uint8 -> int16) or 'inv' (int16 -> uint8): Prepared launch"""
    lib = _lib.load()
    fn = lib.mdct_fwd_u8_i16 if mode == "fwd" else lib.mdct_inv_i16_u8
    keep, lp = _lut_ptr(lut)
    sz = ctypes.c_size_t
    args = (ctypes.c_void_p(_ptr(src)), ctypes.c_void_p(_ptr(dst)), sz(sizeX), sz(sizeX), lp, ctypes.c_int(int(bool(level_shift))), sz(sizeX), sz(sizeY), sz(0), sz(sizeY // 8), _stream(stream))
    return Prepared(fn, args, (keep, src, dst))


def prepare_fwd_quant_u8(src, dst, lut, sizeX, sizeY, by0, by1, layout=LAYOUT_Q32, profile=PROFILE_REF_AVX, pitch_in=None, stream=None):
    lib = _lib.load()
    keep, lp = _lut_ptr(lut)
    sz = ctypes.c_size_t
    args = (ctypes.c_void_p(_ptr(src)), ctypes.c_void_p(_ptr(dst)), sz(sizeX if pitch_in is None else pitch_in), lp, sz(sizeX), sz(sizeY), sz(by0), sz(by1),
            ctypes.c_int(layout), ctypes.c_int(profile), _stream(stream))
    return Prepared(lib.mdct_fwd_quant_u8, args, (keep, src, dst))


def prepare_roundtrip_i16_planes(planes, stream=None):
    """planes as for roundtrip_i16_planes; descriptors marshalled once"""
    arr, keep = _plane_array(planes)
    return Prepared(_lib.load().mdct_roundtrip_i16_planes, (arr, ctypes.c_int(len(planes)), _stream(stream)), keep)


def prepare_i16_batch(mode, planes, stream=None):
    lib = _lib.load()
    fn = {"fwd": lib.mdct_fwd_i16_batch, "inv": lib.mdct_inv_i16_batch, "roundtrip": lib.mdct_roundtrip_i16_batch}[mode]
    arr, keep = _plane_array(planes)
    return Prepared(fn, (arr, ctypes.c_int(len(planes)), _stream(stream)), keep)


def prepare_u8_batch(planes, level_shift=True, stream=None):
    """Prepared launch of mdct_roundtrip_u8_batch (planes as for roundtrip_u8_batch)"""
    arr, keep = _plane_array(planes)
    return Prepared(_lib.load().mdct_roundtrip_u8_batch, (arr, ctypes.c_int(len(planes)), ctypes.c_int(int(bool(level_shift))), _stream(stream)), keep)


def prepare_u8_i16_batch(mode, planes, level_shift=True, stream=None):
    lib = _lib.load()
    arr, keep = _plane_array(planes)
    return Prepared(lib.mdct_fwd_u8_i16_batch if mode == "fwd" else lib.mdct_inv_i16_u8_batch, (arr, ctypes.c_int(len(planes)), ctypes.c_int(int(bool(level_shift))), _stream(stream)), keep)


def prepare_roundtrip_u8(src, dst, sizeX, sizeY, lut=None, level_shift=True, stream=None):
    keep, lp = _lut_ptr(lut)
    sz = ctypes.c_size_t
    args = (ctypes.c_void_p(_ptr(src)), ctypes.c_void_p(_ptr(dst)), sz(sizeX), sz(sizeX), lp, ctypes.c_int(int(bool(level_shift))), sz(sizeX), sz(sizeY), sz(0), sz(sizeY // 8), _stream(stream))
    return Prepared(_lib.load().mdct_roundtrip_u8, args, (keep, src, dst))


def table_cache_stats():
    """mdct_table_cache_stats of the current device: dict of the MDCT_TABLE_STAT_* counters"""
    a = (ctypes.c_uint64 * 6)()
    _check(_lib.load().mdct_table_cache_stats(a, 6))
    return dict(zip(("hits", "uploads", "evictions", "from_arguments", "stream_waits", "unfenceable"), [int(v) for v in a]))


def clock_probe(out, ticks_100MHz, waves=8, stream=None):
    """mdct_clock_probe: `out` = int64 device tensor of 2 * waves entries (shader cycles, 100 MHz ticks) per wave"""
    _check(_lib.load().mdct_clock_probe(_ptr(out), int(ticks_100MHz), int(waves), _stream(stream)))


def prepare_stream_copy(src, dst, nbytes, stream=None):
    lib = _lib.load()
    return Prepared(lib.mdct_stream_copy, (ctypes.c_void_p(_ptr(src)), ctypes.c_void_p(_ptr(dst)), ctypes.c_size_t(nbytes), _stream(stream)), (src, dst))


class Timer:
    """HIP-event timer on the stream the kernels are launched on (mdct_timer_*)."""

    def __init__(self):
        self._lib = _lib.load()
        self._t = self._lib.mdct_timer_create()
        if not self._t:
            raise MdctError(last_error())

    def start(self, stream=None):
        _check(self._lib.mdct_timer_start(self._t, _stream(stream)))

    def stop(self, stream=None):
        _check(self._lib.mdct_timer_stop(self._t, _stream(stream)))

    def wait_spin(self):
        """poll the stop event until it has completed (no interrupt wake-up latency)"""
        _check(self._lib.mdct_timer_wait_spin(self._t))

    def elapsed_ms(self):
        ms = self._lib.mdct_timer_elapsed_ms(self._t)
        if ms < 0:
            raise MdctError(last_error())
        return ms

    def __del__(self):
        try:
            self._lib.mdct_timer_destroy(self._t)
        except Exception:
            pass
