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
        api.set_max_simd(api.SIMD_AVX2)
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


def _cxx(tmp_path, name, source, link=True):
    """compile (and link against libmdct_hip.so) a small C++ caller, the way a user of the reference would"""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = tmp_path / (name + ".cpp")
    src.write_text(source)
    exe = tmp_path / name
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src)]
    cmd += (["-L" + os.path.dirname(_lib.LIB_PATH), "-lmdct_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)] if link else ["-fsyntax-only"])
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(exe)


def test_header_carries_the_reference_macros(tmp_path, lib):
    """simd_dct.h:7-20: IN / OUT / IN_OUT and _SUCCEEDED / _FAILED survive a header swap"""
    _cxx(tmp_path, "macros", '''
#include "simd_dct_shim.h"
static_assert(_SUCCEEDED(sdr_Success) && _FAILED(sdr_NotSupported) && _FAILED(sdr_InvalidParameter), "status tests");
static simdDctResult call(IN const uint8_t *a, OUT uint8_t *b, IN_OUT float *t) { return simdDCT_EncodeQuantizeBuffer(a, b, t, 8, 8, 0, 8); }
int main() { (void)&call; return 0; }
''', link=False)


def test_shim_follows_the_reference_cpu_flags_when_linked(tmp_path, lib):
    """A program that still links the reference's simd_platform.c defines its flag globals
    (simd_platform.h:21-46); the shim then picks tiers exactly as simd_dct.cpp:78-85, :100-105, :120-127
    do -- all false until the caller runs _DetectCPUFeatures() -- unless mdct_shim_set_max_simd() overrides."""
    import subprocess

    exe = _cxx(tmp_path, "flags", '''
#include <cstdio>
#include "simd_dct_shim.h"
extern "C" { bool sse2Supported = false, ssse3Supported = false, sse41Supported = false, avx2Supported = false, avx512VLSupported = false; }
int main() {
  static uint8_t a[64 * 16], b[64 * 16];
  static float lut[64];
  for (float &f : lut) f = 1.f;
  printf("%d", mdct_shim_get_max_simd());                      // nothing detected: scalar tiers
  printf(" %d", (int)simdDCT_EncodeQuantize32ReorderBuffer(a, b, lut, 64, 16, 0, 16)); // -> sdr_NotSupported (:127), no device touched
  sse2Supported = true;            printf(" %d", mdct_shim_get_max_simd());
  ssse3Supported = true;           printf(" %d", mdct_shim_get_max_simd());
  sse41Supported = true;           printf(" %d", mdct_shim_get_max_simd());
  avx2Supported = true;            printf(" %d", mdct_shim_get_max_simd());
  avx2Supported = false; avx512VLSupported = true; printf(" %d", mdct_shim_get_max_simd());
  mdct_shim_set_max_simd(MDCT_SIMD_SSSE3); printf(" %d", mdct_shim_get_max_simd());   // explicit cap wins
  mdct_shim_set_max_simd(-1);      printf(" %d", mdct_shim_get_max_simd());            // unset: flags again
  printf(" %d\\n", (int)simdDCT_EncodeQuantizeBuffer(a, b, lut, 0, 16, 0, 16));        // sizeX == 0: nothing to do, success
  return 0;
}
''')
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["0", "2", "1", "2", "3", "4", "4", "2", "4", "0"], r.stdout
    # without those globals in the program the default is the AVX2 tier
    assert api.get_max_simd() == api.SIMD_AVX2


def test_cli_max_simd_spellings():
    """main.cpp:87-97 spells the tiers sse4.1 / sse4.2 / avx512f / avx512bw; unknown ones abort (:438)"""
    import subprocess

    cli = G.build_cli()
    r = subprocess.run([cli, "synthetic:noise", "64", "16", "--max-simd", "sse41"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Invalid SIMD Variant 'sse41'" in r.stdout
    for ok in ("sse4.1", "sse4.2", "avx512f", "avx512bw", "avx", "ssse3", "sse3", "sse2", "none", "avx2"):
        r = subprocess.run([cli, "synthetic:noise", "64", "16", "--max-simd", ok, "--cpu-core", "0", "--runs", "1"], capture_output=True, text=True, timeout=60)
        assert "Invalid" not in r.stdout, (ok, r.stdout)
        assert r.returncode in (0, 2, 3), (ok, r.returncode, r.stdout)  # 2/3: no HIP device in this container


def test_pitched_output_argument_checks(lib):
    a = np.zeros(64 * 16, dtype=np.uint8)
    lut = api.QUANTIZE_BASE
    assert api.fwd_quant_u8(a, a, lut, 64, 16, 0, 2, pitch_out=256, check=False) == 1  # < 8*sizeX
    assert api.fwd_quant_u8(a, a, lut, 64, 16, 0, 2, pitch_out=520, check=False) == 1  # not a multiple of 16
    assert api.fwd_quant_u8(a, a, lut, 64, 16, 0, 1, layout=api.LAYOUT_STEREO, profile=api.PROFILE_REF_SSE, pitch_out=1024, check=False) == 2
    assert "Q32 and BLOCK" in api.last_error()
    # sizeX == 0 passes the reference's shape test and does nothing (simd_dct.cpp:118, :2103)
    assert api.simdDCT_EncodeQuantize32ReorderBuffer(a, a, lut, 0, 16, 0, 16) == api.sdr_Success
    assert api.simdDCT_EncodeQuantizeBuffer(a, a, lut, 0, 16, 0, 16) == api.sdr_Success


def test_plain_c_caller_of_the_batch_api_compiles_links_and_gets_status_codes(tmp_path, lib):
    """include/mdct.h is a C header: a gcc -std=c99 program lists planes in mdct_plane_i16, calls the batch entry points and the
    device-table form, and -- without a device in this container -- gets the reference's status codes back: argument errors (1, 2)
    before any device is touched, MDCT_NOT_SUPPORTED (2) with a message where a device would be needed"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "batch_c.c"
    src.write_text(r'''
#include <stdio.h>
#include "mdct.h"
int main(void) {
  static int16_t a[64 * 16], b[64 * 16];
  float lut[64];
  for (int i = 0; i < 64; i++) lut[i] = 16.0f;
  mdct_plane_i16 pl[2] = {{a, b, 64, 64, 64, 16, lut}, {a, b, 64, 64, 64, 16, NULL}};
  mdct_batch *h = (mdct_batch *)1;
  pl[1].sizeX = 60;
  printf("%d", mdct_roundtrip_i16_batch(pl, 2, NULL));      /* 2: not a multiple of 8x8, nothing launched */
  pl[1].sizeX = 64; pl[1].to = NULL;
  printf(" %d", mdct_fwd_i16_batch(pl, 2, NULL));           /* 1: null plane pointer */
  pl[1].to = b;
  printf(" %d", mdct_batch_create(&h, 9, pl, 2));           /* 1: unknown mode; the handle is cleared */
  printf(" %d", h == NULL);
  printf(" %d", mdct_batch_create(&h, MDCT_MODE_INV, pl, 2) != MDCT_INVALID_PARAMETER); /* valid arguments: success on a GPU box, 2 here */
  printf(" %d %d\n", mdct_batch_launches(NULL), mdct_batch_destroy(h == (mdct_batch *)1 ? NULL : h));
  return 0;
}
''')
    exe = tmp_path / "batch_c"
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-L" + os.path.dirname(_lib.LIB_PATH), "-lmdct_hip",
                        "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["2", "1", "1", "1", "1", "0", "0"], r.stdout
