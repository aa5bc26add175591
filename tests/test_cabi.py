"""CPU tests of the boundary: the library loads, exports every symbol the headers declare
(plus the reference's three mangled C++ names), and rejects bad arguments with the
reference's status codes before touching any device."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as G
from simd_dct_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    G.build_hip()
    return _lib.load()


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mdct_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared("mdct.h") + _declared("simd_dct_shim.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound but not declared in include/"


def test_reference_cxx_symbols_exported(lib):
    # simd_dct.h:29-31 has C++ linkage; a relinked caller resolves exactly these names
    for m in _lib.MANGLED:
        assert hasattr(lib, m), m


def test_library_carries_gfx950_code_object():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"k_fwd_quant_u8" in data and b"k_i16" in data


def test_status_codes_without_device(lib):
    a = np.zeros(64 * 16, dtype=np.uint8)
    lut = api.QUANTIZE_BASE
    # reference dispatcher order: null -> 1, shape -> 2 (simd_dct.cpp:75-76, :117-118)
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(None, a, lut, 64, 16, 0, 16) == api.sdr_InvalidParameter
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, None, lut, 64, 16, 0, 16) == api.sdr_InvalidParameter
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, a, lut, 56, 16, 0, 16) == api.sdr_NotSupported
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, a, lut, 64, 12, 0, 16) == api.sdr_NotSupported
    assert api.simdDCT_EncodeQuantizeBuffer(a, a, lut, 60, 16, 0, 16) == api.sdr_NotSupported
    assert api.simdDCT_EncodeQuantizeReorderStereoBuffer(None, a, lut, 64, 16, 0, 16) == api.sdr_InvalidParameter
    # empty row range: nothing to do, success without a device (simd_dct.cpp:2252 breaks at once)
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, a, lut, 64, 16, 40, 16) == api.sdr_Success
    # no scalar q32 tier exists: --max-simd none -> sdr_NotSupported (simd_dct.cpp:127)
    api.set_max_simd(0)
    try:
        assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, a, lut, 64, 16, 0, 16) == api.sdr_NotSupported
    finally:
        api.set_max_simd(2)
    # native entry points
    assert api.fwd_quant_u8(None, a, lut, 64, 16, 0, 2, check=False) == 1
    assert api.fwd_quant_u8(a, a, lut, 56, 16, 0, 2, check=False) == 2
    assert api.fwd_quant_u8(a, a, lut, 64, 16, 0, 3, check=False) == 1  # range beyond the plane
    assert "range" in api.last_error()
    assert api.fwd_quant_u8(a, a, lut, 64, 16, 0, 2, layout=api.LAYOUT_STEREO, profile=api.PROFILE_REF_AVX, check=False) == 2
    b = np.zeros(64 * 16, dtype=np.int16)
    assert api.roundtrip_i16(b, b, 60, 16, check=False) == 2
    assert api.fwd_i16(b, None, 64, 16, check=False) == 1
    assert api.fwd_i16(b[1:], b, 64, 8, check=False) == 1  # rows not 16-byte aligned


def test_product_does_not_import_the_oracle():
    """the oracle is the checker; nothing under simd_dct_amd/ or include/ may reach it"""
    for base in ("simd_dct_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    assert "liboracle" not in txt and "dct_oracle" not in txt and "import oracle" not in txt, os.path.join(dp, fn)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmdct_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_synthetic_generators_agree_host_and_torch():
    """bench and tests feed the CPU checker and the GPU path from the same counter hash"""
    import torch

    from simd_dct_amd import synth

    for kind in ("noise", "photo"):
        a = synth.plane_u8_np(96, 40, kind, seed=123)
        b = synth.plane_u8_torch(96, 40, kind, seed=123, device="cpu").numpy()
        assert np.array_equal(a, b), kind
    for bits in (8, 12):
        a = synth.plane_i16_np(64, 24, "photo", seed=7, bits=bits)
        b = synth.plane_i16_torch(64, 24, "photo", seed=7, bits=bits, device="cpu").numpy()
        assert np.array_equal(a, b), bits
        assert a.min() >= -(1 << (bits - 1)) and a.max() < (1 << (bits - 1))
