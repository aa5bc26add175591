"""ctypes access to the CHECKER: oracle/liboracle.so (the repo's plain-C restatement) and,
when it has been built in the container that holds /root/reference, oracle/_ref/ (the real
reference).  Test infrastructure -- never imported by the package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsimd_dct_ref.so")

sz = ctypes.c_size_t
vp = ctypes.c_void_p
f32p = ctypes.POINTER(ctypes.c_float)

_orc = None
_ref = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle():
    global _orc
    if _orc is None:
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("dct_oracle.c", "dct_oracle.h", "time_mt.c"))
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < src_m:
            build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        for n in ("orc_q32_avx", "orc_stereo_sse", "orc_encq_sse", "orc_stereo_scalar", "orc_encq_scalar"):
            getattr(lib, n).argtypes = [vp, vp, f32p, sz, sz, sz, sz]
        lib.orc_q32_native.argtypes = [vp, vp, sz, f32p, sz, sz, sz, sz]
        for n in ("orc_fwd_i16", "orc_inv_i16", "orc_roundtrip_i16"):
            getattr(lib, n).argtypes = [vp, vp, sz, sz, f32p, sz, sz, sz, sz]
        for n in ("orc_fwd_u8_i16", "orc_inv_i16_u8"):
            getattr(lib, n).argtypes = [vp, vp, sz, sz, f32p, ctypes.c_int, sz, sz, sz, sz]
        lib.orc_roundtrip_u8.argtypes = [vp, vp, sz, sz, f32p, ctypes.c_int, sz, sz, sz, sz]
        for n in ("orc_fwd_f32", "orc_inv_f32", "orc_fwd_f64ref"):
            getattr(lib, n).argtypes = [vp, vp, sz, sz, sz, sz, sz, sz]
        lib.orc_zigzag_table.argtypes = [vp]
        lib.orc_zigzag_rle_i16.argtypes = [vp, sz, sz, sz, sz, sz, vp, vp, vp]
        lib.orc_zigzag_rle_q32.argtypes = [vp, sz, sz, sz, sz, vp, vp, vp]
        lib.orc_zigzag_rle_u8.argtypes = [vp, ctypes.c_int, sz, sz, sz, sz, vp, vp, vp]
        lib.orc_huffman_rows.argtypes = [vp, vp, vp, sz, sz, sz, sz, ctypes.c_int, vp, sz, vp]
        lib.orc_jpeg_pack_rows.argtypes = [vp, vp, sz, sz, ctypes.c_int, vp, sz, vp]
        lib.orc_huffman_spec.argtypes = [ctypes.c_int, vp, vp, vp]
        lib.orc_split420_u8.argtypes = [vp, sz, sz, sz, vp, vp, vp, sz, sz]
        lib.orc_time_q32_mt.argtypes = [vp, ctypes.c_int, vp, vp, f32p, sz, sz, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        lib.orc_dct8.argtypes = [vp, ctypes.c_ssize_t, ctypes.c_int]
        lib.orc_idct8_own.argtypes = [vp, ctypes.c_ssize_t]
        _orc = lib
    return _orc


def reference():
    """the real reference build, or None (it exists only where oracle/Makefile `ref` ran)"""
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        lib = ctypes.CDLL(REF_SO)
        lib.ref_call_tier.argtypes = [ctypes.c_int, vp, vp, f32p, sz, sz, sz, sz]
        lib.ref_call_public.argtypes = [ctypes.c_int, vp, vp, f32p, sz, sz, sz, sz]
        lib.ref_detect_cpu.argtypes = []
        lib.ref_detect_cpu.restype = None
        _ref = lib
    return _ref


# every tier function the reference exports (oracle/ref_driver.cpp ids) -> the x86 feature it needs
REF_TIERS = {
    "q32_avx2": (0, "avx2"), "q32_avx512vl": (1, "avx512vl"),
    "stereo_sse41": (2, "sse4_1"), "stereo_ssse3": (3, "ssse3"), "stereo_sse2": (4, "sse2"), "stereo_scalar": (5, None),
    "encq_sse41": (6, "sse4_1"), "encq_ssse3": (7, "ssse3"), "encq_scalar": (8, None),
}


def host_cpu_flags():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def run_tier(tier, img, lut, W, H, y0, y1, out=None):
    """the real reference's tier function `tier` (a REF_TIERS key), called directly"""
    img = np.ascontiguousarray(img, dtype=np.uint8).reshape(-1)
    out = np.zeros(W * H, dtype=np.uint8) if out is None else out
    keep, lp = _lut(lut)
    rc = reference().ref_call_tier(REF_TIERS[tier][0], img.ctypes.data, out.ctypes.data, lp, W, H, y0, y1)
    assert rc == 0, rc
    return out


def run_public(which, img, lut, W, H, y0, y1, out=None, detect=True):
    """the real reference's PUBLIC dispatcher (0 q32, 1 stereo, 2 encq; simd_dct.cpp:71-133), after the
    caller-side _DetectCPUFeatures() main.cpp:449 performs.  Returns (simdDctResult, out)."""
    img = np.ascontiguousarray(img, dtype=np.uint8).reshape(-1)
    out = np.zeros(W * H, dtype=np.uint8) if out is None else out
    keep, lp = _lut(lut)
    if detect:
        reference().ref_detect_cpu()
    rc = reference().ref_call_public(which, img.ctypes.data, out.ctypes.data, lp, W, H, y0, y1)
    return rc, out


def _lut(lut):
    a = np.ascontiguousarray(np.asarray(lut, dtype=np.float32).reshape(64))
    return a, a.ctypes.data_as(f32p)


REF_FUNCS = {  # behaviour -> (oracle symbol, reference tier id in oracle/ref_driver.cpp)
    "q32_avx": ("orc_q32_avx", 0),
    "stereo_sse": ("orc_stereo_sse", 2),
    "encq_sse": ("orc_encq_sse", 6),
    "stereo_scalar": ("orc_stereo_scalar", 5),
    "encq_scalar": ("orc_encq_scalar", 8),
}


def run_behaviour(name, img, lut, W, H, y0, y1, out=None, use_reference=False):
    """img: uint8 array of W*H bytes.  out: pre-filled W*H buffer (canary) or None (zeros)."""
    img = np.ascontiguousarray(img, dtype=np.uint8).reshape(-1)
    out = np.zeros(W * H, dtype=np.uint8) if out is None else out
    keep, lp = _lut(lut)
    sym, tier = REF_FUNCS[name]
    if use_reference:
        rc = reference().ref_call_tier(tier, img.ctypes.data, out.ctypes.data, lp, W, H, y0, y1)
    else:
        rc = getattr(oracle(), sym)(img.ctypes.data, out.ctypes.data, lp, W, H, y0, y1)
    return rc, out


def q32_native(img, lut, W, H, by0, by1, pitch=None, out=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros(W * H, dtype=np.uint8) if out is None else out
    keep, lp = _lut(lut)
    rc = oracle().orc_q32_native(img.ctypes.data, out.ctypes.data, W if pitch is None else pitch, lp, W, H, by0, by1)
    return rc, out


def i16(mode, src, W, H, lut=None, by0=0, by1=None, out=None):
    src = np.ascontiguousarray(src, dtype=np.int16)
    out = np.zeros((H, W), dtype=np.int16) if out is None else out
    lp = None
    if lut is not None:
        keep, lp = _lut(lut)
    fn = getattr(oracle(), {"fwd": "orc_fwd_i16", "inv": "orc_inv_i16", "roundtrip": "orc_roundtrip_i16"}[mode])
    rc = fn(src.ctypes.data, out.ctypes.data, W, W, lp, W, H, by0, H // 8 if by1 is None else by1)
    assert rc == 0, rc
    return out


def u8_i16(mode, src, W, H, lut=None, level_shift=True):
    """mode 'fwd': uint8 [H,W] -> int16; 'inv': int16 [H,W] -> uint8"""
    lp = None
    if lut is not None:
        keep, lp = _lut(lut)
    if mode == "fwd":
        src = np.ascontiguousarray(src, dtype=np.uint8)
        out = np.zeros((H, W), dtype=np.int16)
        rc = oracle().orc_fwd_u8_i16(src.ctypes.data, out.ctypes.data, W, W, lp, int(level_shift), W, H, 0, H // 8)
    else:
        src = np.ascontiguousarray(src, dtype=np.int16)
        out = np.zeros((H, W), dtype=np.uint8)
        rc = oracle().orc_inv_i16_u8(src.ctypes.data, out.ctypes.data, W, W, lp, int(level_shift), W, H, 0, H // 8)
    assert rc == 0, rc
    return out


def roundtrip_u8(src, W, H, lut=None, level_shift=True, by0=0, by1=None, pitch_in=None, pitch_out=None, out=None, threads=1):
    """uint8 [H, pitch_in] -> uint8 [H, pitch_out]: the fused 8-bit round trip (== u8_i16('inv', u8_i16('fwd', ..)), tested);
    threads > 1: block-row stripes on host threads (whole large planes)"""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    pin = W if pitch_in is None else pitch_in
    pout = W if pitch_out is None else pitch_out
    out = np.zeros((H, pout), dtype=np.uint8) if out is None else out
    lp = None
    if lut is not None:
        keep, lp = _lut(lut)
    by1 = H // 8 if by1 is None else by1
    if threads > 1:
        _par_rows(lambda a, b: oracle().orc_roundtrip_u8(src.ctypes.data, out.ctypes.data, pin, pout, lp, int(level_shift), W, H, by0 + a, by0 + b), by1 - by0, threads)
    else:
        rc = oracle().orc_roundtrip_u8(src.ctypes.data, out.ctypes.data, pin, pout, lp, int(level_shift), W, H, by0, by1)
        assert rc == 0, rc
    return out


def f32(mode, src, W, H, by0=0, by1=None):
    src = np.ascontiguousarray(src, dtype=np.float32)
    out = np.zeros((H, W), dtype=np.float64 if mode == "f64ref" else np.float32)
    fn = getattr(oracle(), {"fwd": "orc_fwd_f32", "inv": "orc_inv_f32", "f64ref": "orc_fwd_f64ref"}[mode])
    rc = fn(src.ctypes.data, out.ctypes.data, W, W, W, H, by0, H // 8 if by1 is None else by1)
    assert rc == 0, rc
    return out


# ------------------------------------------------------------------ whole-plane checks, threaded
def host_threads(cap=32):
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def _par_rows(call, rows, threads=None):
    """run call(b0, b1) over disjoint block-row stripes on host threads (ctypes releases the GIL)"""
    import threading

    threads = host_threads() if threads is None else threads
    threads = max(1, min(threads, rows))
    cuts = [rows * i // threads for i in range(threads + 1)]
    errs = []

    def work(a, b):
        try:
            rc = call(a, b)
            if rc != 0:
                errs.append(rc)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(a, b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def i16_par(mode, src, W, H, lut=None, out=None, threads=None):
    """i16() over a whole (large) plane on all host threads"""
    src = np.ascontiguousarray(src, dtype=np.int16)
    out = np.empty((H, W), dtype=np.int16) if out is None else out
    lp = None
    if lut is not None:
        keep, lp = _lut(lut)
    fn = getattr(oracle(), {"fwd": "orc_fwd_i16", "inv": "orc_inv_i16", "roundtrip": "orc_roundtrip_i16"}[mode])
    _par_rows(lambda a, b: fn(src.ctypes.data, out.ctypes.data, W, W, lp, W, H, a, b), H // 8, threads)
    return out


def q32_native_par(img, lut, W, H, out=None, threads=None):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(W * H, dtype=np.uint8) if out is None else out
    keep, lp = _lut(lut)
    _par_rows(lambda a, b: oracle().orc_q32_native(img.ctypes.data, out.ctypes.data, W, lp, W, H, a, b), H // 8, threads)
    return out


def f32_par(mode, src, W, H, threads=None):
    src = np.ascontiguousarray(src, dtype=np.float32)
    out = np.empty((H, W), dtype=np.float64 if mode == "f64ref" else np.float32)
    fn = getattr(oracle(), {"fwd": "orc_fwd_f32", "inv": "orc_inv_f32", "f64ref": "orc_fwd_f64ref"}[mode])
    _par_rows(lambda a, b: fn(src.ctypes.data, out.ctypes.data, W, W, W, H, a, b), H // 8, threads)
    return out


# ------------------------------------------------------------------ stages either side of the transform
def zigzag_table():
    zz = np.zeros(64, dtype=np.uint8)
    oracle().orc_zigzag_table(zz.ctypes.data)
    return zz


def zigzag_rle(kind, src, W, H, rle=True, by0=0, by1=None, pitch=None, fill=0):
    """kind 'i16' (int16 plane [H, pitch]), 'q32', 'stereo' or 'block' (bytes).  Returns (levels [nblk, 64], runs, counts); arrays pre-filled with `fill`."""
    nblk = (W // 8) * (H // 8)
    levels = np.full((nblk, 64), fill, dtype=np.int16)
    runs = np.full((nblk, 64), fill & 0xFF, dtype=np.uint8) if rle else None
    counts = np.full(nblk, fill & 0xFF, dtype=np.uint8) if rle else None
    by1 = H // (16 if kind == "stereo" else 8) if by1 is None else by1
    rp = runs.ctypes.data if rle else None
    cp = counts.ctypes.data if rle else None
    if kind == "i16":
        src = np.ascontiguousarray(src, dtype=np.int16)
        rc = oracle().orc_zigzag_rle_i16(src.ctypes.data, W if pitch is None else pitch, W, H, by0, by1, levels.ctypes.data, rp, cp)
    elif kind == "q32":
        src = np.ascontiguousarray(src, dtype=np.uint8)
        rc = oracle().orc_zigzag_rle_q32(src.ctypes.data, W, H, by0, by1, levels.ctypes.data, rp, cp)
    else:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        rc = oracle().orc_zigzag_rle_u8(src.ctypes.data, {"stereo": 1, "block": 2}[kind], W, H, by0, by1, levels.ctypes.data, rp, cp)
    assert rc == 0, rc
    return levels, runs, counts


def u8_records(img, W, H, lut=None, level_shift=True):
    """the checker's composition mdct_fwd_u8_records must equal: pixels -> int16 coefficients -> run/level records"""
    return zigzag_rle("i16", u8_i16("fwd", img, W, H, lut=lut, level_shift=level_shift), W, H)


def split420(ycc, W, H):
    ycc = np.ascontiguousarray(ycc, dtype=np.uint8)
    y = np.zeros((H, W), dtype=np.int16)
    cb = np.zeros((H // 2, W // 2), dtype=np.int16)
    cr = np.zeros((H // 2, W // 2), dtype=np.int16)
    rc = oracle().orc_split420_u8(ycc.ctypes.data, 3 * W, W, H, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, W, W // 2)
    assert rc == 0, rc
    return y, cb, cr


def huffman_spec(which):
    bits = np.zeros(16, dtype=np.uint8)
    vals = np.zeros(256, dtype=np.uint8)
    n = ctypes.c_int()
    assert oracle().orc_huffman_spec(which, bits.ctypes.data, vals.ctypes.data, ctypes.byref(n)) == 0
    return bits.tolist(), vals[:n.value].tolist()


def huffman_rows(levels, runs, counts, W, H, chroma=False, by0=0, by1=None, fill=0):
    """-> (segments uint8 [(H/8) * stride], seg_bytes uint32 [H/8], stride)"""
    stride = (W // 8) * 208 + 8
    seg = np.full((H // 8) * stride, fill, dtype=np.uint8)
    nb = np.full(H // 8, fill * 0x01010101, dtype=np.uint32)
    rc = oracle().orc_huffman_rows(levels.ctypes.data, runs.ctypes.data, counts.ctypes.data, W, H, by0, H // 8 if by1 is None else by1, int(chroma),
                                   seg.ctypes.data, stride, nb.ctypes.data)
    assert rc == 0, rc
    return seg, nb, stride


def jpeg_pack_rows(seg, seg_bytes, stride, first_rst=0, capacity=None, fill=0):
    """-> (out uint8 [capacity], row_off uint64 [n_rows + 1])"""
    seg = np.ascontiguousarray(seg, dtype=np.uint8)
    nb = np.ascontiguousarray(seg_bytes, dtype=np.uint32)
    n = len(nb)
    capacity = 2 * int(nb.sum()) + 2 * n if capacity is None else capacity
    out = np.full(capacity, fill, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    rc = oracle().orc_jpeg_pack_rows(seg.ctypes.data, nb.ctypes.data, stride, n, first_rst, out.ctypes.data, capacity, off.ctypes.data)
    assert rc == 0, rc
    return out, off
