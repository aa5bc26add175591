"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the oracle on
the same seeded inputs, against the fixtures the real reference produced, and -- at
BASELINE.json's full sizes -- through size-independent properties.  Bit-exact everywhere
(u8 / int16 / float32 are all compared with array_equal); the only tolerance is config 5's
float32-vs-double bound, stated where it is used."""
import os

import numpy as np
import pytest

import oracle as O
import simd_dct_amd as M
from simd_dct_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CANARY = 0xA5
BEHAVIOURS = {  # behaviour -> (reference-API function, --max-simd level, native layout, native profile)
    "q32_avx": (M.simdDCT_EncodeQuantize32ReorderBuffer, M.SIMD_AVX2, M.LAYOUT_Q32, M.PROFILE_REF_AVX),
    "stereo_sse": (M.simdDCT_EncodeQuantizeReorderStereoBuffer, M.SIMD_AVX2, M.LAYOUT_STEREO, M.PROFILE_REF_SSE),
    "encq_sse": (M.simdDCT_EncodeQuantizeBuffer, M.SIMD_AVX2, M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE),
    "stereo_scalar": (M.simdDCT_EncodeQuantizeReorderStereoBuffer, 0, M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR),
    "encq_scalar": (M.simdDCT_EncodeQuantizeBuffer, 0, M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR),
}


@pytest.fixture(scope="module", autouse=True)
def device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    M.init(0)
    info = M.device_info()
    assert info["is_gfx950"] and info["wavefront_size"] == 64, info
    yield info
    M.set_max_simd(M.SIMD_AVX2)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def lut_x(scale):
    return (M.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)


def run_ref_api(beh, img_np, lut, W, H, y0, y1, host=False):
    fn, level, _, _ = BEHAVIOURS[beh]
    M.set_max_simd(level)
    try:
        if host:
            out = np.full(W * H, CANARY, dtype=np.uint8)
            rc = fn(np.ascontiguousarray(img_np).reshape(-1), out, lut, W, H, y0, y1)
            return rc, out
        src = dev(img_np.reshape(-1))
        out = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
        rc = fn(src, out, lut, W, H, y0, y1)
        torch.cuda.synchronize()
        return rc, out.cpu().numpy()
    finally:
        M.set_max_simd(M.SIMD_AVX2)


# ------------------------------------------------------------------ reference fixtures
def test_reference_fixtures_through_the_drop_in_api(golden):
    """the bytes the REAL reference wrote (tests/golden, incl. untouched canary bytes)"""
    meta, vec = golden
    W, H = meta["W"], meta["H"]
    for c in meta["cases"]:
        rc, out = run_ref_api(c["behaviour"], vec["in_" + c["input"]], lut_x(c["scale"]), W, H, c["startY"], c["endY"])
        assert rc == 0, (c, M.last_error())
        assert np.array_equal(out, vec[c["key"]]), c["key"]
    for kind in ("noise", "photo"):  # sizeY = 2H call trick: full plane through the reference API
        src = dev(vec["in_" + kind].reshape(-1))
        out = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(src, out, lut_x(2000), W, 2 * H, 0, 2 * H) == 0
        assert np.array_equal(out.cpu().numpy(), vec[f"q32_full__{kind}"])


def test_reference_fixtures_host_pointers(golden):
    """same, with plain host memory in and out (the reference's only calling mode)"""
    meta, vec = golden
    W, H = meta["W"], meta["H"]
    for c in meta["cases"]:
        if c["scale"] not in (2000.0, 8.0):
            continue
        rc, out = run_ref_api(c["behaviour"], vec["in_" + c["input"]], lut_x(c["scale"]), W, H, c["startY"], c["endY"], host=True)
        assert rc == 0, (c, M.last_error())
        assert np.array_equal(out, vec[c["key"]]), c["key"]


# ------------------------------------------------------------------ oracle, seeded inputs
@pytest.mark.parametrize("beh", list(BEHAVIOURS))
def test_reference_api_matches_oracle(beh):
    rng = np.random.default_rng(2026)
    for (W, H) in ((64, 16), (192, 48), (448, 80), (1024, 256)):
        if "stereo" in beh and H % 16:
            continue
        for kind, scale in (("noise", 2000.0 if beh == "q32_avx" else 8.0), ("photo", 100.0 if beh == "q32_avx" else 1.0)):
            img = synth.plane_u8_np(W, H, kind, seed=synth.SEED + W)
            lut = (lut_x(scale) * rng.uniform(0.5, 2.0, 64).astype(np.float32)).astype(np.float32)
            for (y0, y1) in ((0, H), (16, 32), (0, 0), (8, H // 2), (H, H)):
                rc, got = run_ref_api(beh, img, lut, W, H, y0, y1)
                want = np.full(W * H, CANARY, dtype=np.uint8)
                rc2, want = O.run_behaviour(beh, img, lut, W, H, y0, y1, out=want)
                assert rc == rc2 == 0
                assert np.array_equal(got, want), (beh, W, H, kind, y0, y1, int((got != want).sum()))


def test_q32_native_ranges_pitch_and_tails():
    """native C-ABI: half-open full-plane ranges, input pitch > width, partial last wave,
    untouched rows keep the canary, and disjoint ranges add up (the multi-GPU shard property)"""
    for (W, H) in ((64, 8), (64, 24), (320, 40), (1984, 72), (4096, 64)):
        pitch = W + 64
        img = synth.plane_u8_np(pitch, H, "photo")  # a wider plane; the engine sees a W-wide window
        lut = lut_x(2000)
        src = dev(img)
        rows = H // 8
        for (b0, b1) in ((0, rows), (rows // 2, rows), (0, 1), (rows - 1, rows), (1, 1)):
            out = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
            M.fwd_quant_u8(src, out, lut, W, H, b0, b1, pitch_in=pitch)
            want = np.full(W * H, CANARY, dtype=np.uint8)
            O.q32_native(img, lut, W, H, b0, b1, pitch=pitch, out=want)
            assert np.array_equal(out.cpu().numpy(), want), (W, H, b0, b1)
        parts = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
        cuts = sorted({0, rows // 3, rows // 2, rows})
        for a, b in zip(cuts[:-1], cuts[1:]):
            M.fwd_quant_u8(src, parts, lut, W, H, a, b, pitch_in=pitch)
        full = np.zeros(W * H, dtype=np.uint8)
        O.q32_native(img, lut, W, H, 0, rows, pitch=pitch, out=full)
        assert np.array_equal(parts.cpu().numpy(), full)


def test_unaligned_input_pointer():
    """the reference takes any alignment (unaligned loads, simd_dct.cpp:2109)"""
    W, H = 128, 32
    img = synth.plane_u8_np(W, H, "noise")
    buf = torch.zeros(W * H + 16, dtype=torch.uint8, device="cuda")
    for off in (1, 3, 4):
        buf[off:off + W * H] = dev(img.reshape(-1))
        out = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(buf[off:], out, lut_x(2000), W, 2 * H, 0, 2 * H) == 0
        rc, want = O.q32_native(img, lut_x(2000), W, H, 0, H // 8)
        assert np.array_equal(out.cpu().numpy(), want), off


def test_extreme_tables_integer_indefinite_corner():
    """cvtps_epi32 returns 0x80000000 for NaN / out-of-range: exercised with
    zero, tiny, negative, infinite and NaN table entries (the SAFE kernel variant)"""
    W, H = 64, 32
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, W * H, dtype=np.uint8)
    img[: W * 8] = 0
    for special in (1e-4, 1e-7, 0.0, -0.3, np.inf, np.nan, 1e-30, 3e38):
        lut = M.QUANTIZE_BASE.copy()
        lut[::3] = special
        for beh in BEHAVIOURS:
            rc, got = run_ref_api(beh, img, lut, W, H, 0, H)
            want = np.full(W * H, CANARY, dtype=np.uint8)
            O.run_behaviour(beh, img, lut, W, H, 0, H, out=want)
            assert rc == 0 and np.array_equal(got, want), (beh, special, int((got != want).sum()))


def test_status_codes_on_device():
    a = torch.zeros(64 * 16, dtype=torch.uint8, device="cuda")
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(a, a, M.QUANTIZE_BASE, 56, 16, 0, 16) == M.sdr_NotSupported
    assert M.simdDCT_EncodeQuantizeReorderStereoBuffer(a, a, M.QUANTIZE_BASE, 64, 8, 0, 8) == M.sdr_NotSupported  # documented: needs H % 16
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(None, a, M.QUANTIZE_BASE, 64, 16, 0, 16) == M.sdr_InvalidParameter


# ------------------------------------------------------------------ engine-own variants
@pytest.mark.parametrize("bits", [8, 12])
def test_i16_fwd_inv_roundtrip_match_oracle(bits):
    lut = lut_x(40)
    for (W, H) in ((8, 8), (64, 16), (200, 40), (1024, 128)):
        src = synth.plane_i16_np(W, H, "photo", bits=bits)
        d = dev(src)
        for table in (None, lut):
            for mode, fn in (("fwd", M.fwd_i16), ("inv", M.inv_i16), ("roundtrip", M.roundtrip_i16)):
                out = torch.full((H, W), -21846, dtype=torch.int16, device="cuda")
                fn(d, out, W, H, lut=table)
                want = O.i16(mode, src, W, H, lut=table)
                assert np.array_equal(out.cpu().numpy(), want), (mode, W, H, bits, table is not None)
        # bit-exact round trip (config 2) and fwd -> inv through HBM
        rt = torch.empty_like(d)
        M.roundtrip_i16(d, rt, W, H)
        assert torch.equal(rt, d)


def test_i16_saturation_pitch_and_ranges():
    W, H, pitch = 64, 32, 96
    rng = np.random.default_rng(5)
    src = rng.integers(-32768, 32768, (H, pitch), dtype=np.int16)
    src[:8, :8] = 32767
    src[8:16, :8] = -32768
    d = dev(src)
    for mode, fn in (("fwd", M.fwd_i16), ("inv", M.inv_i16), ("roundtrip", M.roundtrip_i16)):
        out = torch.full((H, pitch), 77, dtype=torch.int16, device="cuda")
        fn(d, out, W, H, by0=1, by1=3, pitch_in=pitch, pitch_out=pitch)
        want = np.full((H, pitch), 77, dtype=np.int16)
        o = O.oracle()
        f = getattr(o, {"fwd": "orc_fwd_i16", "inv": "orc_inv_i16", "roundtrip": "orc_roundtrip_i16"}[mode])
        assert f(src.ctypes.data, want.ctypes.data, pitch, pitch, None, W, H, 1, 3) == 0
        assert np.array_equal(out.cpu().numpy(), want), mode


def test_i16_roundtrip_without_saturations_at_the_edge_of_its_bound():
    """Tables with every entry >= 8.01 take the round-trip build that leaves the int16 saturation of the quantised coefficient out (it
    cannot fire: |coefficient| <= 8 * 32768).  The largest coefficients int16 samples can produce -- block (u, v) = the sign pattern of
    basis function (u, v) at -32768 / 32767 -- through 8.01 (no saturations), 8.0 (keeps them) and a mixed table: the oracle's values,
    in the tile kernel (512-wide), the linear kernel and the plane batch"""
    xs = np.arange(8)
    cosm = np.cos((2 * xs[None, :] + 1) * xs[:, None] * np.pi / 16)
    worst = np.zeros((16, 512), dtype=np.int16)
    for u in range(8):
        for v in range(8):
            blk = np.where(np.outer(cosm[v], cosm[u]) > 0, 32767, -32768).astype(np.int16)
            worst[0:8, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = blk
            worst[8:16, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = -1 - blk
    for q in (np.full(64, 8.01, dtype=np.float32), np.full(64, 8.0, dtype=np.float32), np.where(np.arange(64) % 5 == 0, 7.9, 8.5).astype(np.float32), np.full(64, 100.0, dtype=np.float32)):
        for W, src in ((512, worst), (256, np.ascontiguousarray(worst[:, :256])), (200, np.ascontiguousarray(worst[:, 56:256]))):
            want = O.i16("roundtrip", src, W, 16, lut=q)
            coef = O.i16("fwd", src, W, 16, lut=q)
            if W == 512 and q[0] <= 8.01:
                assert int(np.abs(coef.astype(np.int64)).max()) >= 32700  # the pattern does reach the edge of int16
            out = torch.full((16, W), 77, dtype=torch.int16, device="cuda")
            M.roundtrip_i16(dev(src), out, W, 16, lut=q)
            assert np.array_equal(out.cpu().numpy(), want), (float(q.min()), W)
            out2 = torch.full((16, W), 77, dtype=torch.int16, device="cuda")
            M.roundtrip_i16_planes([(dev(src), out2, W, 16, q)])
            assert np.array_equal(out2.cpu().numpy(), want), (float(q.min()), W, "planes")


def test_f32_matches_oracle_and_double():
    # widths % 512 == 0 take the kernel's wide-load form (lane pairs swap half rows), the others the plain form
    for (W, H) in ((8, 8), (256, 64), (1000, 24), (512, 16), (1024, 40), (1536, 8)):
        src = synth.plane_u8_np(W, H, "photo").astype(np.float32) - 100.5
        d = dev(src)
        out = torch.empty_like(d)
        M.fwd_f32(d, out, W, H)
        got = out.cpu().numpy()
        assert np.array_equal(got, O.f32("fwd", src, W, H))  # bit-exact vs the oracle
        want = O.f32("f64ref", src, W, H)
        blk = lambda a: a.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        # config 5 tolerance: 1e-5 relative to the block's max-abs coefficient
        rel = np.abs(blk(got.astype(np.float64)) - blk(want)).max(1) / np.abs(blk(want)).max(1)
        assert rel.max() < 1e-5
        back = torch.empty_like(d)
        M.inv_f32(out, back, W, H)
        assert np.array_equal(back.cpu().numpy(), O.f32("inv", got, W, H))
        if H >= 24:  # a block-row sub-range leaves the other rows alone (both load forms)
            part = torch.full_like(d, 3.25)
            M.fwd_f32(d, part, W, H, by0=1, by1=2)
            wantp = np.full((H, W), 3.25, dtype=np.float32)
            wantp[8:16] = O.f32("fwd", src, W, H)[8:16]
            assert np.array_equal(part.cpu().numpy(), wantp)


def test_plane_batch_420_one_call():
    """config 3 in miniature: Y + Cb + Cr with per-plane tables, plus a batch > 4 planes"""
    shapes = [(256, 128), (128, 64), (128, 64), (64, 8), (72, 24)]
    luts = [lut_x(30), lut_x(60), lut_x(60) * np.float32(1.5), None, lut_x(10)]
    srcs = [synth.plane_i16_np(w, h, "photo", seed=synth.SEED + i) for i, (w, h) in enumerate(shapes)]
    d_in = [dev(s) for s in srcs]
    d_out = [torch.empty_like(t) for t in d_in]
    M.roundtrip_i16_planes([(a, b, w, h, l) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)])
    for s, o, (w, h), l in zip(srcs, d_out, shapes, luts):
        assert np.array_equal(o.cpu().numpy(), O.i16("roundtrip", s, w, h, lut=l)), (w, h)


# ------------------------------------------------------------------ BASELINE.json sizes
def test_full_size_8192_whole_plane_vs_oracle():
    """8192x8192 (configs 1/2): EVERY block of the plane against the oracle (threaded over block-row
    stripes on the host), plus the size-independent properties."""
    W = H = 8192
    # config 2: fused fwd -> inv of an int16 plane is a bit-exact round trip
    src = synth.plane_i16_torch(W, H, "photo")
    dst = torch.empty_like(src)
    M.roundtrip_i16(src, dst, W, H)
    assert torch.equal(src, dst)
    host = synth.plane_i16_np(W, H, "photo")
    assert np.array_equal(src.cpu().numpy(), host)  # host and device generators agree at full size
    # forward alone (the round-trip identity does not constrain it), whole plane, with and without a table
    coef = torch.empty_like(src)
    for table in (None, lut_x(40)):
        M.fwd_i16(src, coef, W, H, lut=table)
        want = O.i16_par("fwd", host, W, H, lut=table)
        assert np.array_equal(coef.cpu().numpy(), want), table is not None
        # inverse alone, whole plane, on those coefficients
        M.inv_i16(coef, dst, W, H, lut=table)
        assert np.array_equal(dst.cpu().numpy(), O.i16_par("inv", want, W, H, lut=table)), table is not None
    M.fwd_i16(src, coef, W, H)
    dc = coef[::8, ::8].to(torch.float64)
    sums = src.to(torch.float64).reshape(H // 8, 8, W // 8, 8).sum(dim=(1, 3)) / 8.0
    assert (dc - sums).abs().max().item() <= 0.5 + 1e-3
    # quantised fused round trip == its unfused parts composed on the GPU (independent of the oracle)
    lut = lut_x(40)
    fused = torch.empty_like(src)
    M.roundtrip_i16(src, fused, W, H, lut=lut)
    M.fwd_i16(src, coef, W, H, lut=lut)
    M.inv_i16(coef, dst, W, H, lut=lut)
    assert torch.equal(fused, dst)
    assert np.array_equal(fused.cpu().numpy(), O.i16_par("roundtrip", host, W, H, lut=lut))
    del coef, dst, fused
    # config 1: u8 q32 over the full plane: the whole output vs the oracle, == two half-range calls
    # (linearity of sharding), through the reference API's sizeY = 2H form
    img = synth.plane_u8_torch(W, H, "photo")
    lut = lut_x(2000)
    full = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, full, lut, W, 2 * H, 0, 2 * H) == 0
    halves = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    M.fwd_quant_u8(img, halves, lut, W, H, 0, 400)
    M.fwd_quant_u8(img, halves, lut, W, H, 400, 1024)
    assert torch.equal(full, halves)
    assert np.array_equal(full.cpu().numpy(), O.q32_native_par(synth.plane_u8_np(W, H, "photo"), lut, W, H))


# ------------------------------------------------------------------ remaining BASELINE.json configs
JPEG_LUMA, JPEG_CHROMA = synth.JPEG_LUMA, synth.JPEG_CHROMA  # ITU-T T.81 Annex K.1


def test_config3_420_full_size_one_call():
    """configs[2]: Y 7680x4320 + Cb/Cr 3840x2160, per-plane JPEG Annex-K tables, fwd -> quantise ->
    dequantise -> inverse, all three planes in ONE C-ABI call; every plane compared in full with
    the oracle, with the single-plane entry point, with the unfused fwd(lut) -> inv(lut) composition
    on the GPU, and its quantised coefficients with rint(double-precision DCT / table) within 1."""
    shapes = [(7680, 4320), (3840, 2160), (3840, 2160)]
    luts = [JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA]
    d_in = [synth.plane_i16_torch(w, h, "photo", seed=synth.SEED + i) for i, (w, h) in enumerate(shapes)]
    d_out = [torch.full_like(t, 12345) for t in d_in]
    M.roundtrip_i16_planes([(a, b, w, h, l) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)])
    for i, ((w, h), l) in enumerate(zip(shapes, luts)):
        single = torch.empty_like(d_in[i])
        M.roundtrip_i16(d_in[i], single, w, h, lut=l)
        assert torch.equal(single, d_out[i]), i
        host = d_in[i].cpu().numpy()
        assert np.array_equal(d_out[i].cpu().numpy(), O.i16_par("roundtrip", host, w, h, lut=l)), i
        # independent of the engine's own restatement: the quantised coefficients against the
        # double-precision DCT-II by definition, and the fused call against its unfused parts
        coef = torch.empty_like(d_in[i])
        M.fwd_i16(d_in[i], coef, w, h, lut=l)
        ideal = O.f32_par("f64ref", host.astype(np.float32), w, h) / np.tile(l.reshape(8, 8).astype(np.float64), (h // 8, w // 8))
        diff = np.abs(coef.cpu().numpy() - np.rint(ideal))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (i, diff.max(), (diff != 0).mean())
        back = torch.empty_like(d_in[i])
        M.inv_i16(coef, back, w, h, lut=l)
        assert torch.equal(back, d_out[i]), i
        del coef, back
        err = (d_out[i].to(torch.int32) - d_in[i].to(torch.int32)).abs()
        assert 0 < err.float().mean().item() < 16 and err.max().item() < 160  # lossy (quality-50 tables on noisy content), but a codec


def test_config4_real_shape_256_planes_forward_sharded():
    """configs[3] at its REAL shape on one GPU: 256 independent 4096x4096 int16 planes (8 GiB in,
    8 GiB out), forward only.  (a) the whole batch stacked as one tall plane in ONE launch, every
    plane compared in full with the oracle; (b) sharded as 8 ranks would, by whole planes (32 per
    rank) and (c) by block rows inside every plane (64 of 512 per rank, config 4's wording): both
    reassemble the unsharded bytes.  Only the RCCL all-gather across 8 real GPUs is not run here."""
    W = H = 4096
    planes, world = 256, 8
    rows = H // 8
    tall_in = torch.empty((planes * H, W), dtype=torch.int16, device="cuda")
    for p in range(planes):
        tall_in[p * H:(p + 1) * H] = synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + 100 + p)
    whole = torch.full_like(tall_in, -21846)
    M.fwd_i16(tall_in, whole, W, planes * H)  # (a) one launch, 4.29 Gpx
    pin_i = torch.empty((H, W), dtype=torch.int16).pin_memory()
    pin_o = torch.empty((H, W), dtype=torch.int16).pin_memory()
    for p in range(planes):
        pin_i.copy_(tall_in[p * H:(p + 1) * H])
        pin_o.copy_(whole[p * H:(p + 1) * H])
        torch.cuda.synchronize()
        assert np.array_equal(pin_o.numpy(), O.i16_par("fwd", pin_i.numpy(), W, H)), p
    # (b) plane-sharded: rank r transforms planes [32r, 32r+32) = one contiguous slab of the tall plane
    sharded = torch.zeros_like(tall_in)
    for rank in range(world):
        p0, p1 = M.shard_planes(planes, world, rank)
        M.fwd_i16(tall_in, sharded, W, planes * H, by0=p0 * rows, by1=p1 * rows)
    assert torch.equal(whole, sharded)
    # (c) block-row sharded inside every plane
    sharded.zero_()
    for rank in range(world):
        b0, b1 = M.shard_rows(rows, world, rank)
        for p in range(planes):
            M.fwd_i16(tall_in, sharded, W, planes * H, by0=p * rows + b0, by1=p * rows + b1)
    assert torch.equal(whole, sharded)


def test_config5_f32_full_size_vs_double():
    """configs[4]: float32 DCT-II on 8192x8192, the WHOLE plane: bit-exact vs the oracle and within
    1e-5 of the double-precision definition relative to each block's max-abs coefficient (SURVEY.md
    8c: element-wise relative error is unattainable near zero for any float32 DCT); Parseval and
    inverse(forward) == identity."""
    W = H = 8192
    src = synth.plane_u8_torch(W, H, "photo").to(torch.float32) - 128.0
    out = torch.empty_like(src)
    M.fwd_f32(src, out, W, H)
    host = src.cpu().numpy()
    got = out.cpu().numpy()
    assert np.array_equal(got, O.f32_par("fwd", host, W, H))
    want = O.f32_par("f64ref", host, W, H)
    err = np.abs(got.astype(np.float64) - want).reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    ref = np.abs(want).reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
    assert (err / ref).max() < 1e-5, (err / ref).max()
    del want, err, ref
    # Parseval: an orthonormal transform preserves the energy of every block
    e_in = (src.double() ** 2).sum().item()
    e_out = (out.double() ** 2).sum().item()
    assert abs(e_in - e_out) / e_in < 1e-6
    back = torch.empty_like(src)
    M.inv_f32(out, back, W, H)
    assert np.array_equal(back.cpu().numpy(), O.f32_par("inv", got, W, H))
    assert (back - src).abs().max().item() < 1e-3


def test_shim_is_reentrant_from_two_host_threads():
    """SURVEY 8b threading: concurrent calls on disjoint row ranges from host threads (the
    reference's intended multi-core use), host pointers, per-thread staging buffers."""
    import threading

    W, H = 512, 256
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(2000)
    out = np.zeros(W * H, dtype=np.uint8)
    errs = []

    def work(y0, y1):
        for _ in range(5):
            rc = M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, 2 * H, y0, y1)
            if rc != 0:
                errs.append(rc)

    ts = [threading.Thread(target=work, args=(0, 240)), threading.Thread(target=work, args=(256, 2 * H))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    rc, want = O.q32_native(img.reshape(H, W), lut, W, H, 0, H // 8)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("beh", ["stereo_sse", "encq_sse", "stereo_scalar", "encq_scalar"])
def test_scattered_layouts_from_two_host_threads(beh):
    """the layouts that write scattered bytes (stereo: 64 coefficient planes; SSE encq: half of every
    pair) with plain host pointers from two threads on disjoint row ranges: each call hands back only
    the bytes its range writes, so neither thread clobbers the other's rows, and bytes the reference
    leaves untouched keep the caller's canary"""
    import threading

    fn, level, _, _ = BEHAVIOURS[beh]
    W, H = 512, 512
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(8)
    out = np.full(W * H, CANARY, dtype=np.uint8)
    errs = []
    M.set_max_simd(level)
    try:
        def work(y0, y1):
            for _ in range(4):
                rc = fn(img, out, lut, W, H, y0, y1)
                if rc != 0:
                    errs.append(rc)

        # the reference's test is startY <= 2y <= endY (encq scalar: y): (0, 223) and (240, H) are disjoint in both,
        # with an unprocessed block row in between that receives the SSE encq tier's trailing spill (:1676)
        ts = [threading.Thread(target=work, args=(0, 223)), threading.Thread(target=work, args=(240, H))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        M.set_max_simd(M.SIMD_AVX2)
    assert not errs
    want = np.full(W * H, CANARY, dtype=np.uint8)
    O.run_behaviour(beh, img, lut, W, H, 0, 223, out=want)
    O.run_behaviour(beh, img, lut, W, H, 240, H, out=want)
    assert np.array_equal(out, want), int((out != want).sum())


def test_worker_threads_give_their_staging_back_on_exit():
    """a thread that makes host-pointer calls and exits WITHOUT mdct_shim_release() must not leak
    its HBM mirrors / pinned buffers / streams (thread pools with short-lived workers)"""
    import threading

    W, H = 4096, 4096
    img = synth.plane_u8_np(W, H, "noise").reshape(-1)
    out = np.zeros(W * H, dtype=np.uint8)
    lut = lut_x(2000)

    def work():
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, 2 * H, 0, 2 * H) == 0

    def run_one():
        t = threading.Thread(target=work)
        t.start()
        t.join()

    run_one()  # first use: lazy runtime allocations settle
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(6):
        run_one()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # each worker staged 2 x 16 MiB in HBM; six leaked workers would hold ~200 MiB
    assert free0 - free1 < 48 * 1024 * 1024, (free0, free1)


def test_shim_warmup_moves_the_one_time_costs_out_of_the_first_call():
    """mdct_shim_warmup (include/simd_dct_shim.h): on a fresh host thread the first host-pointer call after a warm-up
    costs what later calls cost (no staging allocation, no helper-thread start inside it), and writes the same bytes"""
    import threading
    import time

    from simd_dct_amd import _lib

    W, H = 4096, 4096
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(2000)
    rc, want = O.q32_native(img.reshape(H, W), lut, W, H, 0, H // 8)  # the sizeY = 2H call form covers the whole plane
    res = {}

    def work(warm):
        out = np.zeros(W * H, dtype=np.uint8)
        if warm:
            assert _lib.load().mdct_shim_warmup(W * H) == 0
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, 2 * H, 0, 2 * H) == 0
            ts.append(time.perf_counter() - t0)
        assert np.array_equal(out, want)
        res[warm] = ts

    for warm in (False, True):  # each on its own thread: staging is per host thread
        t = threading.Thread(target=work, args=(warm,))
        t.start()
        t.join()
    steady = min(res[True][1:] + res[False][1:])
    assert res[True][0] < 3 * steady + 2e-3, res   # warmed: the first call is an ordinary call
    assert _lib.load().mdct_shim_warmup(0) == 0     # idempotent, any size


def test_pitched_output_strips():
    """mdct_fwd_quant_u8_pitched: Q32 and BLOCK strips at a caller-chosen pitch; the padding keeps
    the canary, the strips equal the tight result"""
    for (W, H) in ((64, 8), (320, 40), (1984, 72), (1024, 128)):
        img = synth.plane_u8_np(W, H, "photo")
        src = dev(img)
        rows = H // 8
        for layout, profile, lut, beh in ((M.LAYOUT_Q32, M.PROFILE_REF_AVX, lut_x(2000), None), (M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, lut_x(8), "encq_scalar")):
            tight = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
            M.fwd_quant_u8(src, tight, lut, W, H, 0, rows, layout=layout, profile=profile)
            for pitch in (8 * W, 8 * W + 16, 8 * W + 4096):
                for (b0, b1) in ((0, rows), (rows // 2, rows), (0, 1)):
                    out = torch.full((rows, pitch), CANARY, dtype=torch.uint8, device="cuda")
                    M.fwd_quant_u8(src, out, lut, W, H, b0, b1, layout=layout, profile=profile, pitch_out=pitch)
                    want = np.full((rows, pitch), CANARY, dtype=np.uint8)
                    want[b0:b1, :8 * W] = tight.cpu().numpy().reshape(rows, 8 * W)[b0:b1]
                    assert np.array_equal(out.cpu().numpy(), want), (W, H, layout, pitch, b0, b1)


def test_cxx_cli_on_the_reference_api(tmp_path):
    """the C++ host tool (counterpart of the reference's main.cpp, tools/simd_dct_cli.cpp) calls
    the three reference symbols with host and with device pointers; its dumped output must be
    the oracle's bytes"""
    import subprocess

    import __graft_entry__ as G

    cli = G.build_cli()
    W, H = 512, 256
    img = synth.plane_u8_np(W, H, "photo")
    for mode, beh, level in (("enc-quant32", "q32_avx", "avx2"), ("enc-quant-stereo", "stereo_sse", "sse4.1"), ("enc-quant", "encq_scalar", "none"), ("enc-quant", "encq_scalar", "sse2"),
                             ("enc-quant", "encq_sse", "ssse3"), ("enc-quant-stereo", "stereo_sse", "sse2"), ("enc-quant-stereo", "stereo_scalar", "none")):
        for extra in ([], ["--resident"]):
            dump = tmp_path / f"{mode}{level}{len(extra)}.bin"
            r = subprocess.run([cli, "synthetic:photo", str(W), str(H), "--mode", mode, "--quality", "8", "--runs", "3", "--max-simd", level, "--to", str(dump)] + extra,
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "sdr_Success" in r.stdout
            # the reference's table (main.cpp:72-73: min / mean clk/byte, min / mean MiB/s) plus GB/s and the roofline shares
            head = next(l for l in r.stdout.splitlines() if l.startswith("mode "))
            for col in ("min clk/byte", "mean clk/byte (sigma)", "min MiB/s (nominal)", "mean MiB/s", "alg. GB/s", "% of 8 TB/s", "% of measured copy", "% of measured link"):
                assert col in head, (col, head)
            row = [c.strip() for c in next(l for l in r.stdout.splitlines() if l.startswith(mode + " ")).split("|")]
            assert len(row) == len(head.split("|")) and float(row[2]) > 0 and float(row[4].split()[0]) > 0  # clk/byte and ns/byte were measured
            assert (row[-2] != "-") == bool(extra) and ("measured copy of" in r.stdout) == bool(extra)  # copy roofline: --resident only
            assert (row[-1] != "-") == (not extra) and ("(measured link," in r.stdout) == (not extra)  # the host <-> HBM link: host pointers only
            if not extra:
                assert 0.0 < float(row[-1]) <= 100.0, row  # a host-pointer call cannot beat the link it travels on
            got = np.fromfile(dump, dtype=np.uint8)
            rc, want = O.run_behaviour(beh, img, lut_x(8), W, H, 0, H)
            assert np.array_equal(got, want), (mode, extra)


def test_cxx_cli_async_calls_and_whole_node_batch_run(tmp_path):
    """(i) --resident --async: the reference-API calls issued back to back with one final wait, beside the synchronous rows; the dump
    is still the oracle's bytes.  (ii) --gpus 1 --batch: north_star's whole-node run through the C-ABI from C++ (tools/node_pipeline.h:
    chunked mdct_batch_run on one stream, mdct_allgather_rows of every chunk on a second one behind an event) -- with one rank the
    collective is trivial, the control flow, the three timings and the sampled verification are the real thing; world 2 and 8 of the
    same header run on the CPU (tests/test_node_pipeline.py)."""
    import subprocess

    import __graft_entry__ as G

    cli = G.build_cli()
    W, H = 1024, 512
    dump = tmp_path / "async.bin"
    r = subprocess.run([cli, "synthetic:photo", str(W), str(H), "--mode", "enc-quant32", "--quality", "8", "--runs", "16", "--resident", "--async", "--to", str(dump)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    head = next(l for l in r.stdout.splitlines() if l.startswith("mode (async)"))
    assert "calls back to back, one final wait: us per call" in head
    rows = [l for l in r.stdout.splitlines() if l.startswith("enc-quant32 ")]
    assert len(rows) == 2  # the synchronous row and the asynchronous one
    per_call_us = float(rows[1].split("|")[2])
    sync_ns_per_byte = float(rows[0].split("|")[4])
    assert 0 < per_call_us <= sync_ns_per_byte * W * H * 1e-3 * 1.5  # not slower than waiting for every call
    rc, want = O.run_behaviour("q32_avx", synth.plane_u8_np(W, H, "photo"), lut_x(8), W, H, 0, H)
    assert np.array_equal(np.fromfile(dump, dtype=np.uint8), want)
    r = subprocess.run([cli, "synthetic:photo", "0", "0", "--gpus", "1", "--batch", "24x2048x1024", "--chunk", "5", "--runs", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "24 planes per rank in 6 chunk(s) of 4" in r.stdout and "sampled slots verified" in r.stdout  # chunk 5 shrunk to a divisor of 24
    line = next(l for l in r.stdout.splitlines() if "compute only" in l and "pipelined" in l)
    secs = [float(x.split()[-2]) for x in line.split("|")]
    assert len(secs) == 3 and all(0 < t < 5 for t in secs)
    bad = subprocess.run([cli, "synthetic:photo", "0", "0", "--gpus", "1", "--batch", "24x2044x1024"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "Invalid Parameter" in bad.stdout


def test_host_pointer_pipeline_multi_chunk():
    """plain host memory through the reference API on a plane large enough for the shim's
    chunked two-stream pipeline (several ~4 MiB strips), full range and a partial range;
    rows outside the range keep the caller's bytes"""
    W, H = 4096, 2048
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(2000)
    # whole plane through the sizeY = 2H form
    out = np.full(W * H, CANARY, dtype=np.uint8)
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, 2 * H, 0, 2 * H) == 0
    rc, want = O.q32_native(img.reshape(H, W), lut, W, H, 0, H // 8)
    assert np.array_equal(out, want)
    # reference semantics on the real geometry: top half only, and a sub-range of it
    for (y0, y1) in ((0, H), (512, 1200)):
        out = np.full(W * H, CANARY, dtype=np.uint8)
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, H, y0, y1) == 0
        want = np.full(W * H, CANARY, dtype=np.uint8)
        O.run_behaviour("q32_avx", img, lut, W, H, y0, y1, out=want)
        assert np.array_equal(out, want), (y0, y1)
    # scalar encq tier (BLOCK layout) takes the same pipeline
    M.set_max_simd(0)
    try:
        out = np.full(W * H, CANARY, dtype=np.uint8)
        assert M.simdDCT_EncodeQuantizeBuffer(img, out, lut_x(8), W, H, 0, H) == 0
        want = np.full(W * H, CANARY, dtype=np.uint8)
        O.run_behaviour("encq_scalar", img, lut_x(8), W, H, 0, H, out=want)
        assert np.array_equal(out, want)
    finally:
        M.set_max_simd(M.SIMD_AVX2)


@pytest.mark.parametrize("beh", ["stereo_sse", "encq_sse", "stereo_scalar"])
def test_host_pointer_pipeline_multi_chunk_scattered_layouts(beh):
    """the chunked pipeline on the layouts whose chunks are not one strip -- stereo: 2 input pieces (the stacked images) and 64 output pieces
    (the coefficient planes) per chunk, one 2-D copy each way; SSE encq: of every 128 output bytes only the tier's 64 go from the bounce
    buffer to the caller, the last pair's surviving spill (simd_dct.cpp:1676) follows the last chunk -- on a plane of many chunks: the full
    range, partial ranges, two host threads on disjoint ranges, pageable and pinned caller memory; everything the tier does not write keeps
    the caller's canary"""
    import threading
    from simd_dct_amd import _lib

    lib = _lib.load()
    fn, level, _, _ = BEHAVIOURS[beh]
    W, H = 4096, 4096
    img = np.ascontiguousarray(synth.plane_u8_np(W, H, "photo").reshape(-1))
    lut = lut_x(8)
    M.set_max_simd(level)
    try:
        for pinned in (False, True):
            for (y0, y1) in ((0, H), (496, 2000), (1024, 1039)):
                out = np.full(W * H, CANARY, dtype=np.uint8)
                if pinned:
                    assert lib.mdct_shim_pin(img.ctypes.data, img.nbytes) == 0 and lib.mdct_shim_pin(out.ctypes.data, out.nbytes) == 0
                try:
                    assert fn(img, out, lut, W, H, y0, y1) == 0
                finally:
                    if pinned:
                        assert lib.mdct_shim_unpin(img.ctypes.data) == 0 and lib.mdct_shim_unpin(out.ctypes.data) == 0
                want = np.full(W * H, CANARY, dtype=np.uint8)
                O.run_behaviour(beh, img, lut, W, H, y0, y1, out=want)
                assert np.array_equal(out, want), (beh, pinned, y0, y1, int((out != want).sum()))
        # two threads, disjoint ranges of the same planes, an unprocessed block row between them (it receives the SSE encq tier's spill)
        out = np.full(W * H, CANARY, dtype=np.uint8)
        errs = []

        def work(y0, y1):
            for _ in range(2):
                if fn(img, out, lut, W, H, y0, y1) != 0:
                    errs.append((y0, y1))

        ts = [threading.Thread(target=work, args=(0, 1903)), threading.Thread(target=work, args=(1920, H))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs
        want = np.full(W * H, CANARY, dtype=np.uint8)
        O.run_behaviour(beh, img, lut, W, H, 0, 1903, out=want)
        O.run_behaviour(beh, img, lut, W, H, 1920, H, out=want)
        assert np.array_equal(out, want), int((out != want).sum())
    finally:
        M.set_max_simd(M.SIMD_AVX2)


def test_host_pipeline_helpers_under_four_concurrent_callers():
    """four host threads, each on its own quarter of a 8192 x 8192 plane through the reference API with pageable memory:
    every call runs the multi-chunk pipeline with its own three copy helpers (CopyPool, csrc/shim.hip); repeated, then a
    thread that exits gives helpers and buffers back and a fresh one starts over"""
    import threading

    W, H = 8192, 8192
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(2000)
    out = np.full(W * H, CANARY, dtype=np.uint8)
    errs = []

    def work(y0, y1, reps):
        for _ in range(reps):
            rc = M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, 2 * H, y0, y1)  # sizeY = 2H: the whole plane is in range
            if rc != 0:
                errs.append(rc)

    quarter = 2 * H // 4
    for round_ in range(2):  # the second round runs on new threads: the first round's helpers ended with their threads
        ts = [threading.Thread(target=work, args=(k * quarter, (k + 1) * quarter - 16, 3)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errs
    want = np.zeros(W * H, dtype=np.uint8)
    O.q32_native_par(img.reshape(H, W), lut, W, H, out=want)
    assert np.array_equal(out, want)


def test_stacked_batch_is_one_tall_plane():
    """config 4 usage: a batch of independent planes stacked in memory is transformed by ONE call on
    the tall plane (blocks are independent), identical to per-plane calls"""
    W, H, N = 256, 64, 5
    planes = [synth.plane_i16_np(W, H, "photo", seed=synth.SEED + i) for i in range(N)]
    stacked = dev(np.concatenate(planes, axis=0))
    out = torch.empty_like(stacked)
    M.fwd_i16(stacked, out, W, N * H)
    for i, p in enumerate(planes):
        assert np.array_equal(out[i * H:(i + 1) * H].cpu().numpy(), O.i16("fwd", p, W, H)), i


def test_launches_are_graph_capturable():
    """include/mdct.h promises that launches neither allocate nor synchronise: capture a forward, a
    quantised round trip and a q32 call into one hipGraph, replay it on new data, compare with the oracle"""
    W, H = 512, 128
    lut = lut_x(40)
    src = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    coef, rt = torch.empty_like(src), torch.empty_like(src)
    img = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    q = torch.empty_like(img)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm up outside capture (lazy device probe)
        M.fwd_i16(src, coef, W, H)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        M.fwd_i16(src, coef, W, H)
        M.roundtrip_i16(src, rt, W, H, lut=lut)
        M.fwd_quant_u8(img, q, lut_x(2000), W, H, 0, H // 8)
    for seed in (1, 2):
        hs = synth.plane_i16_np(W, H, "photo", seed=seed)
        hi = synth.plane_u8_np(W, H, "noise", seed=seed)
        src.copy_(dev(hs))
        img.copy_(dev(hi.reshape(-1)))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(coef.cpu().numpy(), O.i16("fwd", hs, W, H))
        assert np.array_equal(rt.cpu().numpy(), O.i16("roundtrip", hs, W, H, lut=lut))
        rc, want = O.q32_native(hi, lut_x(2000), W, H, 0, H // 8)
        assert np.array_equal(q.cpu().numpy(), want)


def test_randomised_geometry_sweep():
    """seeded random widths / heights / row ranges / tiers through the native C-ABI vs the oracle's
    reference-semantics functions (which the native ranges are mapped onto), canary-checked"""
    rng = np.random.default_rng(20261003)
    combos = [("q32_avx", 64, 8), ("stereo_sse", 16, 16), ("stereo_scalar", 16, 16), ("encq_sse", 16, 8), ("encq_scalar", 8, 8)]
    for it in range(40):
        beh, xm, ym = combos[it % len(combos)]
        W = int(rng.integers(1, 12)) * xm * (4 if xm < 64 else 1)
        H = int(rng.integers(2, 20)) * 16
        img = rng.integers(0, 256, W * H, dtype=np.uint8)
        lut = (lut_x(float(rng.choice([1.0, 8.0, 100.0, 2000.0]))) * rng.uniform(0.3, 3.0, 64).astype(np.float32)).astype(np.float32)
        y0 = int(rng.integers(0, H))
        y1 = int(rng.integers(y0, 2 * H))
        rc, got = run_ref_api(beh, img, lut, W, H, y0, y1)
        want = np.full(W * H, CANARY, dtype=np.uint8)
        rc2, want = O.run_behaviour(beh, img, lut, W, H, y0, y1, out=want)
        assert rc == rc2 == 0, (beh, W, H, y0, y1, M.last_error())
        assert np.array_equal(got, want), (beh, W, H, y0, y1, int((got != want).sum()))
    for it in range(20):  # engine-own int16 paths: random geometry, pitch, range, table
        W = int(rng.integers(1, 40)) * 8
        H = int(rng.integers(1, 12)) * 8
        pitch = W + 8 * int(rng.integers(0, 3))
        src = rng.integers(-2048, 2048, (H, pitch), dtype=np.int16)
        table = None if it % 3 == 0 else (lut_x(30) * rng.uniform(0.5, 2, 64).astype(np.float32)).astype(np.float32)
        b0 = int(rng.integers(0, H // 8 + 1))
        b1 = int(rng.integers(b0, H // 8 + 1))
        mode = ("fwd", "inv", "roundtrip")[it % 3]
        fn = {"fwd": M.fwd_i16, "inv": M.inv_i16, "roundtrip": M.roundtrip_i16}[mode]
        out = torch.full((H, pitch), 4321, dtype=torch.int16, device="cuda")
        fn(dev(src), out, W, H, lut=table, by0=b0, by1=b1, pitch_in=pitch, pitch_out=pitch)
        want = np.full((H, pitch), 4321, dtype=np.int16)
        o = O.oracle()
        f = getattr(o, {"fwd": "orc_fwd_i16", "inv": "orc_inv_i16", "roundtrip": "orc_roundtrip_i16"}[mode])
        lp = None
        if table is not None:
            keep, lp = O._lut(table)
        assert f(src.ctypes.data, want.ctypes.data, pitch, pitch, lp, W, H, b0, b1) == 0
        assert np.array_equal(out.cpu().numpy(), want), (mode, W, H, pitch, b0, b1)


def test_shim_mixed_pointers_and_async_stream():
    """host->device, device->host and device->device (asynchronous, on a caller-chosen stream)
    calls of the reference API all give the oracle's bytes"""
    from simd_dct_amd import _lib

    W, H = 256, 128
    img = synth.plane_u8_np(W, H, "photo").reshape(-1)
    lut = lut_x(2000)
    want = np.full(W * H, CANARY, dtype=np.uint8)
    O.run_behaviour("q32_avx", img, lut, W, H, 0, H, out=want)
    d_img = dev(img)
    # host in, device out
    d_out = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, d_out, lut, W, H, 0, H) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want)
    # device in, host out
    h_out = np.full(W * H, CANARY, dtype=np.uint8)
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(d_img, h_out, lut, W, H, 0, H) == 0
    assert np.array_equal(h_out, want)
    # device to device, asynchronous on a side stream
    lib = _lib.load()
    side = torch.cuda.Stream()
    d_out2 = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    keep, lp = M.api._lut_ptr(lut)
    lib.mdct_shim_set_stream(side.cuda_stream)  # per-thread setters + the plain C handle
    lib.mdct_shim_set_async(1)
    try:
        assert lib.mdct_shim_call(0, d_img.data_ptr(), d_out2.data_ptr(), lp, W, H, 0, H) == 0
        side.synchronize()
        assert np.array_equal(d_out2.cpu().numpy(), want)
    finally:
        lib.mdct_shim_set_async(0)
        lib.mdct_shim_set_stream(None)
    # stream and async passed per call
    d_out2.fill_(CANARY)
    torch.cuda.synchronize()
    assert lib.mdct_shim_call_on(0, d_img.data_ptr(), d_out2.data_ptr(), lp, W, H, 0, H, side.cuda_stream, 1) == 0
    side.synchronize()
    assert np.array_equal(d_out2.cpu().numpy(), want)
    # the Python mirror runs device-tensor calls on torch's CURRENT stream
    d_out2.fill_(CANARY)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        src_side = d_img.clone()  # produced on `side`; the call must be ordered after it
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(src_side, d_out2, lut, W, H, 0, H) == 0
    assert np.array_equal(d_out2.cpu().numpy(), want)


def test_config0_full_size_hash_of_the_real_reference(golden):
    """the GPU's bytes for the whole 8192x8192 plane hash to what the REAL reference produced
    (tests/golden/ref_vectors.json, generated where /root/reference exists)"""
    import hashlib

    meta, _ = golden
    W = H = 8192
    img = synth.plane_u8_torch(W, H, "photo").reshape(-1)
    out = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut_x(2000), W, H, 0, H) == 0  # main.cpp's own call
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == meta["config0_sha256"]["q32_avx__photo__8192x8192__x2000__half"]
    out.zero_()
    assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut_x(2000), W, 2 * H, 0, 2 * H) == 0
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == meta["config0_sha256"]["q32_avx__photo__8192x8192__x2000__full"]
    out.zero_()
    assert M.simdDCT_EncodeQuantizeReorderStereoBuffer(img, out, lut_x(8), W, H, 0, H) == 0
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == meta["config0_sha256"]["stereo_sse__photo__8192x8192__x8"]
    # the remaining tiers: main.cpp's own calls through the drop-in API under its --max-simd caps ("half"), and the
    # whole plane through the native entry point ("full": what bench.py times and hashes)
    sha = meta["config0_sha256"]
    try:
        out.zero_()
        assert M.simdDCT_EncodeQuantizeBuffer(img, out, lut_x(8), W, H, 0, H) == 0
        assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == sha["encq_sse__photo__8192x8192__x8__half"]
        M.set_max_simd(0)  # --max-simd none: the scalar tiers
        out.zero_()
        assert M.simdDCT_EncodeQuantizeBuffer(img, out, lut_x(8), W, H, 0, H) == 0
        assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == sha["encq_scalar__photo__8192x8192__x8__half"]
        out.zero_()
        assert M.simdDCT_EncodeQuantizeReorderStereoBuffer(img, out, lut_x(8), W, H, 0, H) == 0
        assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == sha["stereo_scalar__photo__8192x8192__x8"]
    finally:
        M.set_max_simd(M.SIMD_AVX2)
    for layout, profile, key in ((M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, "encq_sse__photo__8192x8192__x8__full"),
                                 (M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, "encq_scalar__photo__8192x8192__x8__full")):
        out.zero_()
        M.fwd_quant_u8(img, out, lut_x(8), W, H, 0, H // 8, layout=layout, profile=profile)
        assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == sha[key], key


def test_u8_i16_codec_pair_matches_oracle():
    """8-bit pixels -> int16 coefficients -> 8-bit pixels (JPEG-style pair), with and without a
    table and level shift, odd pitches and an unaligned pixel plane; bit-exact vs the oracle"""
    rng = np.random.default_rng(99)
    for (W, H) in ((8, 8), (72, 24), (512, 64), (1024, 128)):
        img = synth.plane_u8_np(W, H, "photo", seed=W)
        for table in (None, JPEG_LUMA):
            for shift in (True, False):
                d_img = dev(img)
                coef = torch.full((H, W), 31000, dtype=torch.int16, device="cuda")
                M.fwd_u8_i16(d_img, coef, W, H, lut=table, level_shift=shift)
                want = O.u8_i16("fwd", img, W, H, lut=table, level_shift=shift)
                assert np.array_equal(coef.cpu().numpy(), want), (W, H, table is not None, shift)
                back = torch.full((H, W), 7, dtype=torch.uint8, device="cuda")
                M.inv_i16_u8(coef, back, W, H, lut=table, level_shift=shift)
                assert np.array_equal(back.cpu().numpy(), O.u8_i16("inv", want, W, H, lut=table, level_shift=shift))
        # random (not DCT-shaped) coefficients exercise the decoder's saturation
        wild = rng.integers(-3000, 3000, (H, W), dtype=np.int16)
        out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
        M.inv_i16_u8(dev(wild), out, W, H, lut=JPEG_LUMA)
        assert np.array_equal(out.cpu().numpy(), O.u8_i16("inv", wild, W, H, lut=JPEG_LUMA))
    # unaligned pixel plane (the coefficient plane must be 16-byte aligned, the pixels need not be)
    W, H = 256, 32
    img = synth.plane_u8_np(W, H, "noise")
    buf = torch.zeros(W * H + 8, dtype=torch.uint8, device="cuda")
    buf[3:3 + W * H] = dev(img.reshape(-1))
    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
    M.fwd_u8_i16(buf[3:], coef, W, H)
    assert np.array_equal(coef.cpu().numpy(), O.u8_i16("fwd", img, W, H))
    # quality: 8192x8192 through JPEG luma quantisation and back stays a picture
    W = H = 8192
    big = synth.plane_u8_torch(W, H, "photo")
    c = torch.empty((H, W), dtype=torch.int16, device="cuda")
    r = torch.empty_like(big)
    M.fwd_u8_i16(big, c, W, H, lut=JPEG_LUMA)
    M.inv_i16_u8(c, r, W, H, lut=JPEG_LUMA)
    mse = ((r.float() - big.float()) ** 2).mean().item()
    assert 10 * np.log10(255.0 ** 2 / mse) > 24.0
    lossless = torch.empty_like(big)
    M.fwd_u8_i16(big, c, W, H)
    M.inv_i16_u8(c, lossless, W, H)
    assert (lossless.int() - big.int()).abs().max().item() <= 2  # integer coefficients cost at most 2 grey levels


@pytest.mark.skipif(not os.environ.get("MDCT_SOAK"), reason="opt-in soak: MDCT_SOAK=<cases> python -m pytest tests -m gpu -k soak")
def test_soak_random_differential():
    """opt-in long differential run: MDCT_SOAK random cases across every entry point vs the oracle"""
    n = int(os.environ["MDCT_SOAK"])
    rng = np.random.default_rng(int(os.environ.get("MDCT_SOAK_SEED", "1")))
    combos = [("q32_avx", 64), ("stereo_sse", 16), ("stereo_scalar", 16), ("encq_sse", 16), ("encq_scalar", 8)]
    specials = [0.0, -0.7, 1e-6, 1e-30, np.inf, np.nan, 3e38]
    progress = os.environ.get("MDCT_SOAK_PROGRESS")  # a file that gets one line every 5000 cases (long runs on a watched box)
    SOAK_WORK = torch.zeros((64,), dtype=torch.int64, device="cuda")  # the one-launch encoder's row chain: zeroed once for the whole run
    for it in range(n):
        if progress and it % 5000 == 0:
            with open(progress, "a") as f:
                f.write(f"case {it} of {n}\n")
        beh, xm = combos[it % len(combos)]
        W = int(rng.integers(1, 24)) * xm
        H = int(rng.integers(1, 24)) * 16
        kind = rng.integers(0, 3)
        img = rng.integers(0, 256, W * H, dtype=np.uint8) if kind == 0 else (np.full(W * H, rng.integers(0, 256), dtype=np.uint8) if kind == 1 else synth.plane_u8_np(W, H, "photo", seed=it).reshape(-1))
        lut = (lut_x(float(rng.choice([0.01, 1.0, 8.0, 100.0, 2000.0, 1e5]))) * rng.uniform(0.2, 5.0, 64).astype(np.float32)).astype(np.float32)
        if it % 7 == 0:
            lut[rng.integers(0, 64, 5)] = rng.choice(specials)
        y0 = int(rng.integers(0, H + 16))
        y1 = int(rng.integers(0, 2 * H + 16))
        host = bool(it % 3 == 0)
        rc, got = run_ref_api(beh, img, lut, W, H, y0, y1, host=host)
        want = np.full(W * H, CANARY, dtype=np.uint8)
        rc2, want = O.run_behaviour(beh, img, lut, W, H, y0, y1, out=want)
        assert rc == rc2 == 0, (it, beh, W, H, y0, y1, M.last_error())
        assert np.array_equal(got, want), (it, beh, W, H, y0, y1, host, int((got != want).sum()))
        # engine-own
        W2, H2 = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 30)) * 8
        src = rng.integers(-32768, 32768, (H2, W2), dtype=np.int16) if it % 4 == 0 else rng.integers(-2048, 2048, (H2, W2), dtype=np.int16)
        table = None if it % 2 else (lut_x(float(rng.choice([0.5, 10, 200]))) * rng.uniform(0.3, 3, 64).astype(np.float32)).astype(np.float32)
        for mode, fn in (("fwd", M.fwd_i16), ("inv", M.inv_i16), ("roundtrip", M.roundtrip_i16)):
            out = torch.empty((H2, W2), dtype=torch.int16, device="cuda")
            fn(dev(src), out, W2, H2, lut=table)
            assert np.array_equal(out.cpu().numpy(), O.i16(mode, src, W2, H2, lut=table)), (it, mode, W2, H2)
        # plane batches: 1..11 separately allocated planes of random shapes (partial tiles), pitches and tables, all three modes,
        # kernel-argument form and (every 4th case) the device-table form
        nbp = int(rng.integers(1, 12))
        # (a quarter of the planes: rows that end in half a tile -- 32 / 96 / 160 blocks -- which the batch kernels tile over PAIRS of block rows, round 6)
        soak_w = lambda: int(rng.choice([32, 96, 160])) * 8 if rng.random() < 0.25 else int(rng.integers(1, 100)) * 8
        bshapes = [(soak_w(), int(rng.integers(1, 8)) * 8) for _ in range(nbp)]
        if it % 5 == 0:
            bshapes = [bshapes[0]] * nbp  # equal shapes: the division-free plane index
        bpad = [8 * int(rng.integers(0, 3)) for _ in range(nbp)]
        btabs = [None if rng.random() < 0.3 else (table if (table is not None and rng.random() < 0.5) else (lut_x(float(rng.choice([0.5, 9, 150]))) * rng.uniform(0.3, 3, 64).astype(np.float32)).astype(np.float32))
                 for _ in range(nbp)]
        bsrc = []
        for (bw, bh), pd in zip(bshapes, bpad):
            full = rng.integers(-2048, 2048, (bh, bw + pd), dtype=np.int16)
            bsrc.append(full)
        bmode = ("fwd", "inv", "roundtrip")[it % 3]
        b_in = [dev(a) for a in bsrc]
        b_out = [torch.full((bh, bw + pd), 771, dtype=torch.int16, device="cuda") for (bw, bh), pd in zip(bshapes, bpad)]
        bdesc = [(a, o, bw, bh, l, bw + pd, bw + pd) for a, o, (bw, bh), pd, l in zip(b_in, b_out, bshapes, bpad, btabs)]
        if it % 4 == 0:
            bb = M.Batch(bmode, bdesc)
            bb.run()
            bb.close()
        else:
            M.i16_batch(bmode, bdesc)
        for a, o, (bw, bh), pd, l in zip(bsrc, b_out, bshapes, bpad, btabs):
            g = o.cpu().numpy()
            assert np.array_equal(g[:, :bw], O.i16(bmode, np.ascontiguousarray(a[:, :bw]), bw, bh, lut=l)) and (g[:, bw:] == 771).all(), (it, bmode, bw, bh, pd, "batch")
        px = rng.integers(0, 256, (H2, W2), dtype=np.uint8)
        c = torch.empty((H2, W2), dtype=torch.int16, device="cuda")
        M.fwd_u8_i16(dev(px), c, W2, H2, lut=table, level_shift=bool(it & 1))
        assert np.array_equal(c.cpu().numpy(), O.u8_i16("fwd", px, W2, H2, lut=table, level_shift=bool(it & 1)))
        p = torch.empty((H2, W2), dtype=torch.uint8, device="cuda")
        M.inv_i16_u8(dev(src), p, W2, H2, lut=table, level_shift=bool(it & 1))
        assert np.array_equal(p.cpu().numpy(), O.u8_i16("inv", src, W2, H2, lut=table, level_shift=bool(it & 1)))
        # the fused 8-bit round trip (round 5): one plane with a row range and pitches, and a batch of 1..6 planes; tame tables (fast build),
        # wild ones (saturating quantiser, clamping output stage) and none; contents incl. the extremes that saturate the output
        shift = bool(it & 2)
        wild = lut_x(float(rng.choice([0.003, 0.02, 3e4]))) if it % 3 == 0 else table
        px2 = px if it % 5 else (rng.integers(0, 2, (H2, W2)) * 255).astype(np.uint8)
        rb0 = int(rng.integers(0, H2 // 8))
        rb1 = int(rng.integers(rb0, H2 // 8 + 1))
        pin_, pout_ = W2 + int(rng.integers(0, 9)), W2 + int(rng.integers(0, 9))
        src_p = np.full((H2, pin_), 7, dtype=np.uint8)
        src_p[:, :W2] = px2
        got_p = torch.full((H2, pout_), 0xA5, dtype=torch.uint8, device="cuda")
        M.roundtrip_u8(dev(src_p), got_p, W2, H2, lut=wild, level_shift=shift, by0=rb0, by1=rb1, pitch_in=pin_, pitch_out=pout_)
        want_p = np.full((H2, pout_), 0xA5, dtype=np.uint8)
        O.roundtrip_u8(src_p, W2, H2, lut=wild, level_shift=shift, by0=rb0, by1=rb1, pitch_in=pin_, pitch_out=pout_, out=want_p)
        assert np.array_equal(got_p.cpu().numpy(), want_p), (it, "roundtrip_u8", W2, H2, rb0, rb1, pin_, pout_, shift)
        nup = int(rng.integers(1, 7))
        ushapes = [(soak_w(), int(rng.integers(1, 8)) * 8) for _ in range(nup)]
        utabs = [None if rng.random() < 0.25 else (wild if rng.random() < 0.3 else (lut_x(float(rng.choice([1, 16, 150]))) * rng.uniform(0.3, 3, 64).astype(np.float32)).astype(np.float32)) for _ in range(nup)]
        usrc = [rng.integers(0, 256, (uh, uw), dtype=np.uint8) for (uw, uh) in ushapes]
        u_in = [dev(a) for a in usrc]
        u_out = [torch.full((uh, uw), 0xA5, dtype=torch.uint8, device="cuda") for (uw, uh) in ushapes]
        udesc = [(a, o, uw, uh, l) for a, o, (uw, uh), l in zip(u_in, u_out, ushapes, utabs)]
        if it % 4 == 1:
            ub = M.Batch("roundtrip_u8", udesc, level_shift=shift)
            ub.run()
            ub.close()
        else:
            M.roundtrip_u8_batch(udesc, level_shift=shift)
        for a, o, (uw, uh), l in zip(usrc, u_out, ushapes, utabs):
            assert np.array_equal(o.cpu().numpy(), O.roundtrip_u8(a, uw, uh, lut=l, level_shift=shift)), (it, "u8 batch", uw, uh)
        # either half of the same batch: pixels -> coefficients, then (on every third case: arbitrary) coefficients -> pixels
        u_co = [torch.full((uh, uw), -21846, dtype=torch.int16, device="cuda") for (uw, uh) in ushapes]
        hdesc = [(a, c, uw, uh, l) for a, c, (uw, uh), l in zip(u_in, u_co, ushapes, utabs)]
        if it % 4 == 2:
            hb = M.Batch("fwd_u8_i16", hdesc, level_shift=shift)
            hb.run()
            hb.close()
        else:
            M.u8_i16_batch("fwd", hdesc, level_shift=shift)
        for a, c, (uw, uh), l in zip(usrc, u_co, ushapes, utabs):
            assert np.array_equal(c.cpu().numpy(), O.u8_i16("fwd", a, uw, uh, lut=l, level_shift=shift)), (it, "u8 fwd batch", uw, uh)
        if it % 3 == 1:
            uco_np = [rng.integers(-32768, 32768, (uh, uw), dtype=np.int16) for (uw, uh) in ushapes]
            u_co = [dev(a) for a in uco_np]
        else:
            uco_np = [c.cpu().numpy() for c in u_co]
        u_back = [torch.full((uh, uw), 0xA5, dtype=torch.uint8, device="cuda") for (uw, uh) in ushapes]
        idesc = [(o, c, uw, uh, l) for o, c, (uw, uh), l in zip(u_back, u_co, ushapes, utabs)]
        if it % 4 == 3:
            hb = M.Batch("inv_i16_u8", idesc, level_shift=shift)
            hb.run()
            hb.close()
        else:
            M.u8_i16_batch("inv", idesc, level_shift=shift)
        for c, o, (uw, uh), l in zip(uco_np, u_back, ushapes, utabs):
            assert np.array_equal(o.cpu().numpy(), O.u8_i16("inv", c, uw, uh, lut=l, level_shift=shift)), (it, "u8 inv batch", uw, uh)
        # the reference's q32 product on a plane list (widths % 64): ordinary tables, sometimes one that needs the exact-convert build
        nq = int(rng.integers(1, 6))
        qshapes = [(int(rng.integers(1, 14)) * 64, int(rng.integers(1, 6)) * 8) for _ in range(nq)]
        qtabs = [lut_x(float(rng.choice([8, 100, 2000, 5e4]))) * rng.uniform(0.5, 2, 64).astype(np.float32) for _ in range(nq)]
        if it % 7 == 0:
            qtabs[0] = np.full(64, float(rng.choice([1e-4, 3e-6])), dtype=np.float32)
        qsrc = [rng.integers(0, 256, (qh, qw), dtype=np.uint8) for (qw, qh) in qshapes]
        qpad = [int(rng.integers(0, 3)) * 16 for _ in range(nq)]
        q_in = [dev(a) for a in qsrc]
        q_out = [torch.full((qh // 8, 8 * qw + pd), 0xA5, dtype=torch.uint8, device="cuda") for (qw, qh), pd in zip(qshapes, qpad)]
        qdesc = [(a, o, qw, qh, l, qw, 8 * qw + pd) for a, o, (qw, qh), l, pd in zip(q_in, q_out, qshapes, qtabs, qpad)]
        if it % 4 == 0:
            qb = M.Batch("q32", qdesc)
            qb.run()
            qb.close()
        else:
            M.fwd_quant32_u8_batch(qdesc)
        for a, o, (qw, qh), l in zip(qsrc, q_out, qshapes, qtabs):
            g = o.cpu().numpy()
            assert np.array_equal(g[:, : 8 * qw].reshape(-1), O.q32_native(a, l, qw, qh, 0, qh // 8)[1]) and (g[:, 8 * qw:] == 0xA5).all(), (it, "q32 batch", qw, qh)
        f = rng.normal(0, 300, (H2, W2)).astype(np.float32)
        fo = torch.empty((H2, W2), dtype=torch.float32, device="cuda")
        M.fwd_f32(dev(f), fo, W2, H2)
        assert np.array_equal(fo.cpu().numpy(), O.f32("fwd", f, W2, H2))
        M.inv_f32(dev(f), fo, W2, H2)
        assert np.array_equal(fo.cpu().numpy(), O.f32("inv", f, W2, H2))
        # the stages after the transform: fused pixels -> records, records of an arbitrary coefficient plane, Huffman rows of both
        nblk = (W2 // 8) * (H2 // 8)
        sparse = (src * (rng.random((H2, W2)) < rng.choice([0.02, 0.2, 0.9]))).astype(np.int16)
        for recs_want, make in ((O.u8_records(px, W2, H2, lut=table, level_shift=bool(it & 1)), lambda l, r, c: M.fwd_u8_records(dev(px), W2, H2, l, r, c, lut=table, level_shift=bool(it & 1))),
                                (O.zigzag_rle("i16", sparse, W2, H2), lambda l, r, c: M.zigzag_rle_i16(dev(sparse), W2, H2, l, r, c))):
            lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
            rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
            ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
            make(lv, rn, ct)
            for got, want in zip((lv, rn, ct), recs_want):
                assert np.array_equal(got.cpu().numpy(), want), (it, W2, H2)
            stride = M.huffman_seg_stride(W2)
            seg = torch.zeros(((H2 // 8) * stride,), dtype=torch.uint8, device="cuda")
            nb = torch.zeros((H2 // 8,), dtype=torch.int32, device="cuda")
            M.huffman_rows(lv, rn, ct, W2, H2, seg, nb, chroma=bool(it & 2))
            ws, wn, _ = O.huffman_rows(*recs_want, W2, H2, chroma=bool(it & 2))
            gs = seg.cpu().numpy()
            assert np.array_equal(nb.cpu().numpy().astype(np.uint32), wn), (it, W2, H2)
            for r in range(H2 // 8):
                assert np.array_equal(gs[r * stride:r * stride + wn[r]], ws[r * stride:r * stride + wn[r]]), (it, W2, H2, r)
        # ... and the same rows from pixels in ONE kernel, then on into the finished scan, in one launch and in two
        want_lv, want_rn, want_ct = O.u8_records(px, W2, H2, lut=table, level_shift=bool(it & 1))
        ws, wn, _ = O.huffman_rows(want_lv, want_rn, want_ct, W2, H2, chroma=bool(it & 2))
        wscan, woff = O.jpeg_pack_rows(ws, wn, stride, first_rst=it % 8)
        seg = torch.zeros(((H2 // 8) * stride,), dtype=torch.uint8, device="cuda")
        nb = torch.zeros((H2 // 8,), dtype=torch.int32, device="cuda")
        ff = torch.zeros((H2 // 8,), dtype=torch.int32, device="cuda")
        M.fwd_u8_huffman_rows(dev(px), W2, H2, seg, nb, lut=table, level_shift=bool(it & 1), chroma=bool(it & 2), ff_counts=ff)
        gs = seg.cpu().numpy()
        assert np.array_equal(nb.cpu().numpy().astype(np.uint32), wn), (it, W2, H2, "fused")
        for r in range(H2 // 8):
            assert np.array_equal(gs[r * stride:r * stride + wn[r]], ws[r * stride:r * stride + wn[r]]), (it, W2, H2, r, "fused")
        total = int(woff[-1])
        for one_launch in (False, True):
            scan = torch.full((total + 8,), 0x33, dtype=torch.uint8, device="cuda")
            off = torch.zeros((H2 // 8 + 1,), dtype=torch.int64, device="cuda")
            if one_launch:
                M.fwd_u8_jpeg_scan(dev(px), W2, H2, seg, SOAK_WORK, scan, off, lut=table, level_shift=bool(it & 1), chroma=bool(it & 2), first_rst=it % 8, out_capacity=total)
            else:
                M.jpeg_pack_rows(seg, nb, stride, H2 // 8, scan, off, first_rst=it % 8, out_capacity=total, ff_counts=ff)
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), woff) and np.array_equal(scan.cpu().numpy()[:total], wscan[:total]) and (scan[total:] == 0x33).all(), (it, W2, H2, one_launch)


RELINKED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "simd_dct_relinked")


@pytest.mark.skipif(not os.path.exists(RELINKED), reason="oracle/_ref/simd_dct_relinked not built (make -C oracle relink, needs /root/reference)")
def test_reference_harness_relinked_against_the_engine(tmp_path):
    """the drop-in claim at link level: the reference's OWN main.cpp (unchanged, compiled from where
    it lies by oracle/Makefile) linked against libmdct_hip.so instead of the reference's
    simd_dct.cpp, run with its own command line; its --to dump must be the oracle's bytes"""
    import subprocess

    W, H = 512, 256
    img = synth.plane_u8_np(W, H, "photo")
    raw = tmp_path / "in.raw"
    img.tofile(raw)
    # the harness's own --max-simd clears the reference's CPU-flag globals (main.cpp:283-438); the shim
    # follows them (they are linked into this binary), so the scalar / SSE tiers are selectable as before
    for mode, beh, scale, written, simd in (("enc-quant32", "q32_avx", 2000, W * H // 2, []), ("enc-quant-stereo", "stereo_sse", 8, W * H, []),
                                            ("enc-quant-stereo", "stereo_scalar", 8, W * H, ["--max-simd", "none"]), ("enc-quant", "encq_scalar", 8, W * H // 2, ["--max-simd", "sse2"]),
                                            ("enc-quant-stereo", "stereo_sse", 8, W * H, ["--max-simd", "ssse3"])):
        dump = tmp_path / f"{mode}{len(simd) and simd[1]}.bin"
        r = subprocess.run([RELINKED, str(raw), str(W), str(H), "--mode", mode, "--quality", str(scale), "--runs", "2", "--to", str(dump)] + simd, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.fromfile(dump, dtype=np.uint8)
        rc, want = O.run_behaviour(beh, img, lut_x(scale), W, H, 0, H)
        # the harness malloc()s its output buffer: only the bytes the function writes are defined
        assert np.array_equal(got[:written], want[:written]), mode


@pytest.mark.parametrize("autopin", ["1", ""])
def test_shim_autopin_is_opt_in_and_pins_from_the_third_sighting(autopin):
    """MDCT_SHIM_AUTOPIN=1: a host range passed for the third time is page-locked in place and DMA'd from / to directly
    (what the reference's harness gets without a source change: main.cpp:510-523 reuses its two buffers for every run); the SSE encq
    tier's output -- half of every block pair is not the tier's -- always goes through the bounce buffers; mdct_shim_release() lets go.
    Without the variable nothing is ever registered.  Outputs are the oracle's either way."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, MDCT_SHIM_AUTOPIN=autopin)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_autopin_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    HOST = 1  # hipMemoryTypeHost
    for beh in ("q32_avx", "stereo_sse", "encq_sse"):
        e = rep[beh]
        assert e["ok"] and e["ok_after_release"], (beh, e)
        types = [tuple(t) for t in e["types"]]
        if autopin:
            assert types[0] == (0, 0) and types[1] == (0, 0), (beh, types)  # first and second sighting: pageable
            want_out = 0 if beh == "encq_sse" else HOST
            assert all(t == (HOST, want_out) for t in types[2:]), (beh, types)
        else:
            assert all(t == (0, 0) for t in types), (beh, types)
        assert tuple(e["after_release"]) == (0, 0), (beh, e)


def test_shim_pinned_host_buffers_dma_in_place():
    """mdct_shim_pin: page-locked caller buffers are DMA'd in place by the host-pointer pipeline
    (no bounce memcpy); results unchanged, pin/unpin status codes"""
    from simd_dct_amd import _lib

    lib = _lib.load()
    W, H = 4096, 1024
    img = np.ascontiguousarray(synth.plane_u8_np(W, H, "photo").reshape(-1))
    out = np.full(W * H, CANARY, dtype=np.uint8)
    lut = lut_x(2000)
    assert lib.mdct_shim_pin(img.ctypes.data, img.nbytes) == 0
    assert lib.mdct_shim_pin(out.ctypes.data, out.nbytes) == 0
    try:
        assert M.simdDCT_EncodeQuantize32ReorderBuffer(img, out, lut, W, H, 0, H) == 0
        want = np.full(W * H, CANARY, dtype=np.uint8)
        O.run_behaviour("q32_avx", img, lut, W, H, 0, H, out=want)
        assert np.array_equal(out, want)
    finally:
        assert lib.mdct_shim_unpin(img.ctypes.data) == 0
        assert lib.mdct_shim_unpin(out.ctypes.data) == 0
    assert lib.mdct_shim_unpin(out.ctypes.data) == 2  # not registered any more
    assert lib.mdct_shim_pin(None, 16) == 1
