"""CPU tests: the oracle is pinned against (i) fixtures produced by the real reference,
(ii) the survey's SHA-256 known answers, (iii) the real reference itself when oracle/_ref
exists (build container only), and its engine-own parts against a double-precision DCT."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as O
from simd_dct_amd import synth
from simd_dct_amd.api import QUANTIZE_BASE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_matches_reference_fixtures(golden):
    meta, vec = golden
    W, H = meta["W"], meta["H"]
    for c in meta["cases"]:
        lut = (QUANTIZE_BASE * np.float32(c["scale"])).astype(np.float32)
        out = np.full(W * H, meta["canary"], dtype=np.uint8)
        rc, out = O.run_behaviour(c["behaviour"], vec["in_" + c["input"]], lut, W, H, c["startY"], c["endY"], out=out)
        assert rc == 0
        assert np.array_equal(out, vec[c["key"]]), c["key"]


def test_oracle_full_plane_trick(golden):
    meta, vec = golden
    W, H = meta["W"], meta["H"]
    lut = (QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
    for kind in ("noise", "photo"):
        out = np.full(W * H, meta["canary"], dtype=np.uint8)
        O.run_behaviour("q32_avx", vec["in_" + kind], lut, W, 2 * H, 0, 2 * H, out=out)
        assert np.array_equal(out, vec[f"q32_full__{kind}"])
        # and the engine's native full-plane range is the same thing
        rc, nat = O.q32_native(vec["in_" + kind], lut, W, H, 0, H // 8)
        assert rc == 0 and np.array_equal(nat, out)


def test_oracle_large_plane_hashes(golden):
    meta, _ = golden
    for key, sha in meta["sha256_zero_prefilled"].items():
        beh, kind, dims, sc = key.split("__")
        w, h = map(int, dims.split("x"))
        lut = (QUANTIZE_BASE * np.float32(float(sc[1:]))).astype(np.float32)
        rc, out = O.run_behaviour(beh, synth.plane_u8_np(w, h, kind), lut, w, h, 0, h)
        assert hashlib.sha256(out.tobytes()).hexdigest() == sha, key


def test_oracle_survey_known_answers():
    with open(os.path.join(ROOT, "tests", "golden", "survey_known_answers.json")) as f:
        ka = json.load(f)
    W, H = ka["W"], ka["H"]
    i = np.arange(W * H, dtype=np.uint64)
    g0 = ((((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(24))).astype(np.uint8)
    assert hashlib.sha256(g0.tobytes()).hexdigest() == ka["input_g0"]
    for c in ka["cases"]:
        lut = (QUANTIZE_BASE * np.float32(c["scale"])).astype(np.float32)
        out = np.zeros(W * H, dtype=np.uint8)  # buffers stay W*H even for the sizeY = 2H call trick
        rc, out = O.run_behaviour(c["behaviour"], g0, lut, W, c["sizeY"], c["startY"], c["endY"], out=out)
        assert hashlib.sha256(out.tobytes()).hexdigest() == c["sha256"], c


@pytest.mark.skipif(O.reference() is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("beh", list(O.REF_FUNCS))
def test_oracle_equals_real_reference_random(beh):
    rng = np.random.default_rng(7)
    for (W, H) in ((64, 16), (192, 48), (256, 64)):
        for scale in (0.05, 1.0, 8.0, 100.0, 2000.0):
            img = rng.integers(0, 256, W * H, dtype=np.uint8)
            lut = (QUANTIZE_BASE * np.float32(scale) * rng.uniform(0.5, 2.0, 64).astype(np.float32)).astype(np.float32)
            for (y0, y1) in ((0, H), (16, 16), (8, H // 2)):
                a = np.full(W * H, 0x5A, dtype=np.uint8)
                b = a.copy()
                O.run_behaviour(beh, img, lut, W, H, y0, y1, out=a)
                O.run_behaviour(beh, img, lut, W, H, y0, y1, out=b, use_reference=True)
                assert np.array_equal(a, b), (beh, W, H, scale, y0, y1)


@pytest.mark.skipif(O.reference() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_real_reference_extreme_tables():
    """cvtps_epi32 'integer indefinite' corner: tiny, zero, negative, inf, NaN table entries."""
    rng = np.random.default_rng(11)
    W, H = 64, 16
    img = rng.integers(0, 256, W * H, dtype=np.uint8)
    img[:64] = 0  # exact-zero coefficients make 0*inf = NaN reachable
    for special in (1e-4, 1e-7, 0.0, -0.3, np.inf, np.nan, 1e-30, 3e38):
        lut = QUANTIZE_BASE.copy()
        lut[::3] = special
        for beh in O.REF_FUNCS:
            a = np.zeros(W * H, dtype=np.uint8)
            b = a.copy()
            O.run_behaviour(beh, img, lut, W, H, 0, H, out=a)
            O.run_behaviour(beh, img, lut, W, H, 0, H, out=b, use_reference=True)
            assert np.array_equal(a, b), (beh, special)


@pytest.mark.skipif(O.reference() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_every_reference_tier_and_public_dispatcher_is_pinned():
    """The tiers nobody calls by default -- q32 AVX-512VL (simd_dct.cpp:1869-2059), stereo SSSE3 / SSE2
    (:1330-1536, :1106-1327), encq SSSE3 (:1707-1864) -- equal the pinned tier of their family byte for
    byte, and the three PUBLIC dispatchers (:71-133) after _DetectCPUFeatures() pick exactly those."""
    flags = O.host_cpu_flags()
    rng = np.random.default_rng(20261003)
    same = {"q32_avx512vl": ("q32_avx2", "q32_avx"), "stereo_ssse3": ("stereo_sse41", "stereo_sse"), "stereo_sse2": ("stereo_sse41", "stereo_sse"),
            "encq_ssse3": ("encq_sse41", "encq_sse")}
    ran = 0
    for (W, H) in ((64, 16), (192, 48), (256, 64)):
        for scale in (1.0, 8.0, 2000.0):
            img = rng.integers(0, 256, W * H, dtype=np.uint8)
            lut = (QUANTIZE_BASE * np.float32(scale) * rng.uniform(0.5, 2.0, 64).astype(np.float32)).astype(np.float32)
            for (y0, y1) in ((0, H), (16, 32), (8, H // 2)):
                for tier, (pinned, beh) in same.items():
                    if O.REF_TIERS[tier][1] not in flags:
                        continue
                    a = O.run_tier(tier, img, lut, W, H, y0, y1, out=np.full(W * H, 0x5A, dtype=np.uint8))
                    b = O.run_tier(pinned, img, lut, W, H, y0, y1, out=np.full(W * H, 0x5A, dtype=np.uint8))
                    c = np.full(W * H, 0x5A, dtype=np.uint8)
                    O.run_behaviour(beh, img, lut, W, H, y0, y1, out=c)  # the restatement
                    assert np.array_equal(a, b) and np.array_equal(a, c), (tier, W, H, scale, y0, y1)
                    ran += 1
                # public entry points after feature detection: q32 -> AVX-512VL/AVX2, stereo -> SSE4.1, encq -> SSE4.1
                if "avx2" in flags and "sse4_1" in flags:
                    for which, beh in ((0, "q32_avx"), (1, "stereo_sse"), (2, "encq_sse")):
                        rc, a = O.run_public(which, img, lut, W, H, y0, y1, out=np.full(W * H, 0x5A, dtype=np.uint8))
                        c = np.full(W * H, 0x5A, dtype=np.uint8)
                        O.run_behaviour(beh, img, lut, W, H, y0, y1, out=c)
                        assert rc == 0 and np.array_equal(a, c), ("public", which, W, H, scale, y0, y1)
    assert ran > 0
    # dispatcher status codes (simd_dct.cpp:75-76, :117-118): null -> 1, bad shape -> 2
    img = np.zeros(64 * 16, dtype=np.uint8)
    assert O.run_public(0, img, QUANTIZE_BASE, 56, 16, 0, 16)[0] == 2
    assert O.run_public(1, img, QUANTIZE_BASE, 60, 16, 0, 16)[0] == 2
    assert O.reference().ref_call_public(0, None, img.ctypes.data, O._lut(QUANTIZE_BASE)[1], 64, 16, 0, 16) == 1


def test_argument_errors_match_reference_codes():
    img = np.zeros(64 * 16, dtype=np.uint8)
    lut = QUANTIZE_BASE
    keep, lp = O._lut(lut)
    o = O.oracle()
    assert o.orc_q32_avx(None, img.ctypes.data, lp, 64, 16, 0, 16) == 1  # simd_dct.cpp:117
    assert o.orc_q32_avx(img.ctypes.data, None, lp, 64, 16, 0, 16) == 1
    assert o.orc_q32_avx(img.ctypes.data, img.ctypes.data, lp, 56, 16, 0, 16) == 2  # :118, sizeX % 64
    assert o.orc_q32_avx(img.ctypes.data, img.ctypes.data, lp, 64, 12, 0, 16) == 2
    assert o.orc_encq_scalar(img.ctypes.data, img.ctypes.data, lp, 60, 16, 0, 16) == 2  # :98


def test_true_dct_kernels_against_double():
    """K_TRUE (the reference's scalar kernel) and K_OWN are the orthonormal DCT-II; K_AVX / K_SSE
    carry the documented sign quirks on k=3 / k=1 only."""
    rng = np.random.default_rng(3)
    n = np.arange(8)
    C = np.array([[(np.sqrt(1 / 8) if k == 0 else 0.5) * np.cos((2 * nn + 1) * k * np.pi / 16) for nn in n] for k in range(8)])
    o = O.oracle()
    for _ in range(50):
        x = rng.integers(0, 256, 8).astype(np.float32)
        want = C @ x.astype(np.float64)
        for which, bad in ((2, None), (3, None), (0, 3), (1, 1)):
            y = x.copy()
            o.orc_dct8(y.ctypes.data, 1, which)
            err = np.abs(y - want)
            ok = np.ones(8, bool)
            if bad is not None:
                ok[bad] = False
            assert err[ok].max() < 2e-4, (which, err)
    # survey probe row: scalar k1 = -75.0306 (true), SSE k1 = -64.8859
    row = np.array([10, 200, 33, 47, 99, 150, 7, 250], dtype=np.float32)
    a, b = row.copy(), row.copy()
    o.orc_dct8(a.ctypes.data, 1, 2)
    o.orc_dct8(b.ctypes.data, 1, 1)
    assert abs(a[1] + 75.0306) < 1e-3 and abs(b[1] + 64.8859) < 1e-3


def test_own_inverse_inverts_forward():
    rng = np.random.default_rng(5)
    o = O.oracle()
    for _ in range(100):
        x = rng.integers(-2048, 2048, 8).astype(np.float32)
        y = x.copy()
        o.orc_dct8(y.ctypes.data, 1, 3)
        o.orc_idct8_own(y.ctypes.data, 1)
        assert np.abs(y - x).max() < 2e-3


def test_own_planes_against_double_and_roundtrip():
    W, H = 1024, 512  # 8192 blocks per case: the only independent evidence for the engine-own arithmetic
    for bits in (8, 12):
        src = synth.plane_i16_np(W, H, "photo", bits=bits)
        # fused fwd->inv returns the input bit-exactly (config 2's "bit-exact round-trip")
        assert np.array_equal(O.i16("roundtrip", src, W, H), src)
        coef = O.i16("fwd", src, W, H)
        ref = np.rint(O.f32("f64ref", src.astype(np.float32), W, H))
        assert np.abs(coef - ref).max() <= 1  # rounding of a value within 1e-3 of a tie
        assert (coef != ref).mean() < 1e-2  # exact .5 ties (DC = sum/8) fall either side in double
        # unfused fwd -> inv: integer coefficients cost at most a couple of grey levels
        back = O.i16("inv", coef, W, H)
        assert np.abs(back.astype(np.int32) - src).max() <= 2
    # quantised round trip equals the composition of its unfused parts
    lut = (QUANTIZE_BASE * np.float32(40)).astype(np.float32)
    src = synth.plane_i16_np(W, H, "photo")
    rt = O.i16("roundtrip", src, W, H, lut=lut)
    assert np.array_equal(rt, O.i16("inv", O.i16("fwd", src, W, H, lut=lut), W, H, lut=lut))
    assert 0 < np.abs(rt.astype(np.int32) - src).mean() < 12


def test_f32_against_double_tolerance():
    """config 5: 1e-5 relative to the block's max-abs coefficient (element-wise
    relative error is unattainable near zero for ANY float32 DCT)."""
    W, H = 256, 64
    src = synth.plane_u8_np(W, H, "photo").astype(np.float32)
    got = O.f32("fwd", src, W, H).astype(np.float64)
    want = O.f32("f64ref", src, W, H)
    blk = lambda a: a.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    rel = np.abs(blk(got) - blk(want)).max(1) / np.abs(blk(want)).max(1)
    assert rel.max() < 1e-5
    back = O.f32("inv", got.astype(np.float32), W, H)
    assert np.abs(back - src).max() < 1e-3


def test_row_range_is_additive():
    """fake ranks: disjoint block-row ranges into one buffer == one full range"""
    W, H = 128, 64
    img = synth.plane_u8_np(W, H, "noise")
    lut = (QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
    rc, full = O.q32_native(img, lut, W, H, 0, H // 8)
    parts = np.zeros(W * H, dtype=np.uint8)
    for (a, b) in ((0, 3), (3, 4), (4, 8)):
        O.q32_native(img, lut, W, H, a, b, out=parts)
    assert np.array_equal(parts, full)


def test_config0_full_size_reference_cpu_path(golden):
    """BASELINE.json configs[0]: single 8192x8192 uint8 plane through the reference's CPU path
    (no GPU).  The oracle reproduces the hash the real reference gave for the whole plane."""
    meta, _ = golden
    W = H = 8192
    img = synth.plane_u8_np(W, H, "photo")
    lut = (QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
    out = np.zeros(W * H, dtype=np.uint8)
    O.run_behaviour("q32_avx", img, lut, W, 2 * H, 0, 2 * H, out=out)
    assert hashlib.sha256(out.tobytes()).hexdigest() == meta["config0_sha256"]["q32_avx__photo__8192x8192__x2000__full"]
    if O.reference() is not None:  # and the real thing, where it exists
        ref = np.zeros(W * H, dtype=np.uint8)
        O.run_behaviour("q32_avx", img, lut, W, 2 * H, 0, 2 * H, out=ref, use_reference=True)
        assert np.array_equal(ref, out)


def test_config0_full_size_every_behaviour_hashes_to_the_real_reference(golden):
    """the other four behaviours at 8192x8192: the restatement (no reference needed) reproduces the SHA-256 the REAL
    reference's bytes were recorded under (tests/golden/make_golden.py) -- the hashes bench.py checks the GPU against"""
    meta, _ = golden
    sha = meta["config0_sha256"]
    W = H = 8192
    img = synth.plane_u8_np(W, H, "photo")
    lut8 = (QUANTIZE_BASE * np.float32(8)).astype(np.float32)
    for beh in ("stereo_sse", "stereo_scalar"):
        rc, out = O.run_behaviour(beh, img, lut8, W, H, 0, H)
        assert rc == 0 and hashlib.sha256(out.tobytes()).hexdigest() == sha[f"{beh}__photo__8192x8192__x8"], beh
    for beh in ("encq_sse", "encq_scalar"):
        rc, out = O.run_behaviour(beh, img, lut8, W, H, 0, H)  # main.cpp's call: top half
        assert rc == 0 and hashlib.sha256(out.tobytes()).hexdigest() == sha[f"{beh}__photo__8192x8192__x8__half"], beh
        big = np.zeros(W * H + 64, dtype=np.uint8)  # sizeY = 2H form; 64 spare bytes take the SSE tier's surviving spill (:1676)
        rc, _ = O.run_behaviour(beh, img, lut8, W, 2 * H, 0, 2 * H, out=big)
        assert rc == 0 and hashlib.sha256(big[:W * H].tobytes()).hexdigest() == sha[f"{beh}__photo__8192x8192__x8__full"], beh


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """the checker itself is memory-safe: every entry point on exact-size heap buffers under
    ASan + UBSan (CPU only; GPU sanitizers are not available on the pool)"""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "oracle_asan"
    src = os.path.join(ROOT, "oracle")
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + src,
                        os.path.join(src, "dct_oracle.c"), os.path.join(src, "sanitize_driver.c"), "-lm", "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)  # ASan must come first in the library list
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "sanitize ok" in r.stdout, r.stdout + r.stderr


def test_u8_i16_pair_against_double():
    """8-bit pixels <-> int16 coefficients: forward equals rounding the double-precision DCT of the
    level-shifted pixels (up to exact .5 ties), inverse of the forward returns the pixels"""
    W, H = 128, 64
    img = synth.plane_u8_np(W, H, "photo")
    for shift in (True, False):
        coef = O.u8_i16("fwd", img, W, H, level_shift=shift)
        ref = np.rint(O.f32("f64ref", img.astype(np.float32) - (128.0 if shift else 0.0), W, H))
        assert np.abs(coef - ref).max() <= 1 and (coef != ref).mean() < 1e-2
        back = O.u8_i16("inv", coef, W, H, level_shift=shift)
        assert np.abs(back.astype(np.int32) - img).max() <= 2
    # the level shift only moves DC, by exactly 8 * 128 = 1024
    a = O.u8_i16("fwd", img, W, H, level_shift=True).astype(np.int32)
    b = O.u8_i16("fwd", img, W, H, level_shift=False).astype(np.int32)
    d = b - a
    assert (d[::8, ::8] == 1024).all()
    d[::8, ::8] = 0
    assert not d.any()
    # saturation of the decoder: absurd coefficients clamp to [0, 255]
    wild = np.zeros((8, 8), dtype=np.int16)
    wild[0, 0] = 32767
    assert (O.u8_i16("inv", wild, 8, 8) == 255).all()
    wild[0, 0] = -32768
    assert (O.u8_i16("inv", wild, 8, 8) == 0).all()


@pytest.mark.skipif(O.reference() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_real_reference_property_based():
    """hypothesis-driven differential test of the restatement against the real reference: arbitrary geometry
    (multiples of the tier's block size), arbitrary row ranges incl. inverted and out-of-range ones, tables over
    eight decades, constant / random / extreme planes; canary-filled outputs compared in full"""
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")

    @hyp.settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(beh=st.sampled_from(sorted(O.REF_FUNCS)), gw=st.integers(1, 6), gh=st.integers(1, 6), y0=st.integers(0, 200), y1=st.integers(0, 300),
               logscale=st.floats(-3.0, 5.0), kind=st.integers(0, 3), seed=st.integers(0, 2**31 - 1))
    def check(beh, gw, gh, y0, y1, logscale, kind, seed):
        W, H = gw * 64, gh * 16
        rng = np.random.default_rng(seed)
        if kind == 0:
            img = rng.integers(0, 256, W * H, dtype=np.uint8)
        elif kind == 1:
            img = np.full(W * H, rng.integers(0, 256), dtype=np.uint8)
        elif kind == 2:
            img = rng.choice(np.array([0, 255], dtype=np.uint8), W * H)
        else:
            img = synth.plane_u8_np(W, H, "photo", seed=seed).reshape(-1)
        lut = (QUANTIZE_BASE * np.float32(10.0 ** logscale) * rng.uniform(0.25, 4.0, 64).astype(np.float32)).astype(np.float32)
        a = np.full(W * H, 0xC3, dtype=np.uint8)
        b = a.copy()
        ra, _ = O.run_behaviour(beh, img, lut, W, H, y0, y1, out=a)
        rb, _ = O.run_behaviour(beh, img, lut, W, H, y0, y1, out=b, use_reference=True)
        assert np.array_equal(a, b), (beh, W, H, y0, y1, logscale, kind, seed)

    check()


def test_fma_clone_and_libm_fmaf_build_agree(tmp_path):
    """The engine-own butterflies fuse 4 multiplies per pass (fmaf).  oracle/dct_oracle.c compiles the functions that reach them twice
    (an FMA3 clone where fmaf is one instruction, a baseline clone calling libm's exactly rounded fmaf); a build without clones must
    produce the same bits as whichever clone this host resolves to, on full-range inputs."""
    import ctypes
    import subprocess

    so = str(tmp_path / "liboracle_noclones.so")
    subprocess.run(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-fPIC", "-msse4.1", "-DORC_NO_CLONES", "-shared", "-pthread", "-o", so,
                    os.path.join(ROOT, "oracle", "dct_oracle.c"), os.path.join(ROOT, "oracle", "time_mt.c"), "-lm"], check=True)
    plain = ctypes.CDLL(so)
    sz, vp, f32p = ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)
    W, H = 256, 64
    rng = np.random.default_rng(6)
    src = rng.integers(-32768, 32767, size=(H, W), dtype=np.int16)
    for name in ("orc_fwd_i16", "orc_inv_i16", "orc_roundtrip_i16"):
        for lib_lut in (None, (QUANTIZE_BASE * np.float32(7)).astype(np.float32)):
            outs = []
            for lib in (plain, O.oracle()):
                fn = getattr(lib, name)
                fn.argtypes = [vp, vp, sz, sz, f32p, sz, sz, sz, sz]
                out = np.zeros_like(src)
                lp = None if lib_lut is None else lib_lut.ctypes.data_as(f32p)
                assert fn(src.ctypes.data, out.ctypes.data, W, W, lp, W, H, 0, H // 8) == 0
                outs.append(out)
            assert np.array_equal(outs[0], outs[1]), name


def test_engine_own_quantiser_is_one_rounding_of_the_exact_product():
    """c = sat_i16(rne(y * qf)) (round 6): the product of two floats is exact in double, so rint of it is THE correctly rounded quantised
    value -- the checker's fused form must equal it everywhere, ties (y * qf = k + 1/2 exactly) and the saturating range included, where
    rounds 1-5's mul-then-round form differs on the products that round up to a tie.  Same for the 8-bit output stage."""
    import ctypes

    lib = O.oracle()
    f32p = ctypes.POINTER(ctypes.c_float)
    lib.orc_quant_i16.argtypes = [f32p, f32p, ctypes.c_void_p, ctypes.c_size_t]
    lib.orc_sat_u8_rne.argtypes = [f32p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(66)
    n = 1 << 20
    y = (rng.integers(-200000, 200000, n) / 8.0).astype(np.float32)  # raw AAN outputs are multiples of small powers of two
    qf = rng.uniform(1e-3, 0.9, n).astype(np.float32)
    # exact ties: qf = 2^-k, y = (2 m + 1) 2^(k-1)
    k = rng.integers(1, 6, 4096)
    y[:4096] = ((2 * rng.integers(-3000, 3000, 4096) + 1) * 2.0 ** (k - 1)).astype(np.float32)
    qf[:4096] = (2.0 ** -k.astype(np.float64)).astype(np.float32)
    # far outside int16, both signs, and products near 2^22 where the magic add runs out of unit resolution
    y[4096:8192] = rng.uniform(-3e6, 3e6, 4096).astype(np.float32)
    qf[4096:8192] = rng.uniform(0.5, 4.0, 4096).astype(np.float32)
    got = np.zeros(n, dtype=np.int16)
    lib.orc_quant_i16(y.ctypes.data_as(f32p), qf.ctypes.data_as(f32p), got.ctypes.data, n)
    want = np.clip(np.rint(y.astype(np.float64) * qf.astype(np.float64)), -32768, 32767).astype(np.int16)
    assert np.array_equal(got, want)
    old = np.clip(np.rint((y * qf).astype(np.float32)), -32768, 32767).astype(np.int16)  # two roundings
    assert (old != want).sum() > 0  # the distinction is real on this sample
    x = rng.uniform(-40, 300, n).astype(np.float32)
    x[:512] = (np.arange(512) - 128) + 0.5  # ties
    x[512:516] = [np.nan, np.inf, -np.inf, -0.0]
    gu = np.zeros(n, dtype=np.uint8)
    lib.orc_sat_u8_rne(x.ctypes.data_as(f32p), gu.ctypes.data, n)
    wu = np.clip(np.rint(np.nan_to_num(x.astype(np.float64), nan=0.0, posinf=255.0, neginf=0.0)), 0, 255).astype(np.uint8)
    assert np.array_equal(gu, wu)
