"""The fused 8-bit round trip (mdct_roundtrip_u8, mdct_roundtrip_u8_batch, mdct_batch_create_u8): u8 planes in, u8 planes out,
forward -> quantise -> dequantise -> inverse in one pass -- BASELINE.json configs[2] as SURVEY.md 8(d) states it (2 bytes per pixel).
The reference's pixel type is uint8 everywhere (simd_dct.cpp:2107-2143) but it has no inverse: parity is "unpinned by the reference",
pinned by the engine's own definition -- the composition mdct_fwd_u8_i16 -> mdct_inv_i16_u8 (oracle: orc_fwd_u8_i16 -> orc_inv_i16_u8).

CPU: the checker's fused function equals the checker's composition; status codes decided before a device is touched.
GPU (-m gpu): bit-exact against the checker over shapes / pitches / ranges / alignments / tables (tame and wild), equal to the two-call
path on the device, configs[2] at full size (Y 7680x4320 + Cb/Cr 3840x2160, Annex-K tables) in one launch."""
import ctypes

import numpy as np
import pytest

import __graft_entry__ as G
from simd_dct_amd import _lib, api, synth

JPEG_LUMA, JPEG_CHROMA = synth.JPEG_LUMA, synth.JPEG_CHROMA


def _lut(scale):
    return (api.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)


def _tables():
    tiny = np.full(64, 0.01, dtype=np.float32)  # |coefficient / lut| leaves int16: the saturating quantiser
    tiny[0] = 0.05
    mixed = JPEG_LUMA.copy()
    mixed[[3, 17, 40]] = [0.03, -0.04, 0.06]  # below 1/16: general build
    huge = np.full(64, 7000.0, dtype=np.float32)  # |lut|_2 = 56000 > 40000: general build (clamping output stage)
    edge = np.full(64, 4999.0, dtype=np.float32)  # |lut|_2 = 39992: the fast build at the edge of its licence
    neg = -JPEG_CHROMA
    return {"none": None, "luma": JPEG_LUMA, "chroma": JPEG_CHROMA, "base100": _lut(100), "ones": np.ones(64, dtype=np.float32), "sixteenth": np.full(64, 0.0625, dtype=np.float32),
            "tiny": tiny, "mixed": mixed, "huge": huge, "edge": edge, "negative": neg}


def _worst_planes(W, H):
    """contents that drive the output stage to and past both ends of [0, 255]"""
    y, x = np.mgrid[0:H, 0:W]
    check = (((x ^ y) & 1) * 255).astype(np.uint8)
    blocks = ((((x >> 3) ^ (y >> 3)) & 1) * 255).astype(np.uint8)
    stripes = ((x & 4 > 0) * 255).astype(np.uint8)
    rng = np.random.default_rng(5)
    extremes = (rng.integers(0, 2, size=(H, W)) * 255).astype(np.uint8)
    return {"zeros": np.zeros((H, W), np.uint8), "white": np.full((H, W), 255, np.uint8), "check": check, "blocks": blocks, "stripes": stripes, "extremes": extremes}


# ----------------------------------------------------------------------------------------------- CPU
def test_oracle_fused_equals_its_composition():
    """orc_roundtrip_u8 == orc_inv_i16_u8(orc_fwd_u8_i16(x)): every table class, both level shifts, pitched rows, a row range, worst-case contents"""
    import oracle as O

    W, H = 136, 48
    planes = {"photo": synth.plane_u8_np(W, H, "photo"), "noise": synth.plane_u8_np(W, H, "noise", seed=7)}
    planes.update(_worst_planes(W, H))
    for tname, lut in _tables().items():
        for shift in (True, False):
            for pname, img in planes.items():
                want = O.u8_i16("inv", O.u8_i16("fwd", img, W, H, lut=lut, level_shift=shift), W, H, lut=lut, level_shift=shift)
                got = O.roundtrip_u8(img, W, H, lut=lut, level_shift=shift)
                assert np.array_equal(got, want), (tname, shift, pname)
    # pitches and a sub-range: only rows [8, 32) of a pitched plane are written
    img = np.zeros((H, W + 24), np.uint8)
    img[:, :W] = planes["photo"]
    out = np.full((H, W + 40), 0xA5, np.uint8)
    O.roundtrip_u8(img, W, H, lut=JPEG_LUMA, by0=1, by1=4, pitch_in=W + 24, pitch_out=W + 40, out=out)
    full = O.roundtrip_u8(planes["photo"], W, H, lut=JPEG_LUMA)
    assert np.array_equal(out[8:32, :W], full[8:32]) and (out[:8] == 0xA5).all() and (out[32:] == 0xA5).all() and (out[:, W:] == 0xA5).all()
    # threaded stripes == one call
    assert np.array_equal(O.roundtrip_u8(planes["noise"], W, H, lut=JPEG_CHROMA, threads=3), O.roundtrip_u8(planes["noise"], W, H, lut=JPEG_CHROMA))


def test_quantised_round_trip_is_close_to_the_input():
    """a sanity anchor that does not go through the engine's own arithmetic: with the Annex-K luminance table the reconstruction of a smooth
    plane stays within the quantiser's reach of the input, and with no table within one grey level"""
    import oracle as O

    W, H = 256, 64
    y, x = np.mgrid[0:H, 0:W]
    smooth = (128 + 60 * np.sin(x / 23.0) + 40 * np.cos(y / 11.0)).astype(np.uint8)
    exact = O.roundtrip_u8(smooth, W, H, lut=None)
    assert np.abs(exact.astype(int) - smooth.astype(int)).max() <= 1
    lossy = O.roundtrip_u8(smooth, W, H, lut=JPEG_LUMA)
    err = lossy.astype(int) - smooth.astype(int)
    assert np.abs(err).max() <= 24 and np.sqrt((err ** 2).mean()) < 4.0


def test_u8_roundtrip_status_codes_without_device():
    G.build_hip()
    lib = _lib.load()
    b = np.zeros(64 * 16, dtype=np.uint8)
    ok = (b, b, 64, 16, None)
    assert api.roundtrip_u8(b, None, 64, 16, check=False) == 1  # null pointer
    assert api.roundtrip_u8(b, b, 60, 16, check=False) == 2  # not a multiple of 8x8
    assert api.roundtrip_u8(b, b, 64, 16, pitch_in=32, check=False) == 1  # pitch below the width
    assert api.roundtrip_u8(b, b, 64, 16, by0=1, by1=3, check=False) == 1  # range beyond the plane
    assert api.roundtrip_u8_batch([ok, (b, None, 64, 16, None)], check=False) == 1
    assert api.roundtrip_u8_batch([ok, (b, b, 64, 12, None)], check=False) == 2
    bad = np.ones(64, dtype=np.float32)
    bad[9] = np.inf
    assert api.roundtrip_u8_batch([ok, (b, b, 64, 16, bad)], check=False) == 1 and "table" in api.last_error()
    assert lib.mdct_roundtrip_u8_batch(None, 1, 1, None) == 1 and lib.mdct_roundtrip_u8_batch(None, -1, 1, None) == 1
    h = ctypes.c_void_p()
    assert lib.mdct_batch_create_u8(None, None, 0, 1) == 1
    arr, _keep = api._plane_array([(b, b, 64, 12, None)])
    assert lib.mdct_batch_create_u8(ctypes.byref(h), arr, 1, 1) == 2 and not h


# ----------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu
CANARY = 0xA5


@pytest.fixture(scope="module")
def cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    api.init(0)
    return torch


@gpu
def test_roundtrip_u8_matches_oracle_over_tables_and_contents(cuda):
    """one plane, every table class (the fast build, the saturating / clamping general build), both level shifts, contents that saturate"""
    import oracle as O

    torch = cuda
    W, H = 584, 48  # 73 blocks per row: a full tile and a 9-block partial one
    planes = {"photo": synth.plane_u8_np(W, H, "photo"), "noise": synth.plane_u8_np(W, H, "noise", seed=7)}
    planes.update(_worst_planes(W, H))
    for tname, lut in _tables().items():
        for shift in (True, False):
            for pname, img in planes.items():
                src = torch.from_numpy(img).cuda()
                dst = torch.full_like(src, CANARY)
                api.roundtrip_u8(src, dst, W, H, lut=lut, level_shift=shift)
                want = O.roundtrip_u8(img, W, H, lut=lut, level_shift=shift)
                assert np.array_equal(dst.cpu().numpy(), want), (tname, shift, pname)


@gpu
def test_roundtrip_u8_equals_the_two_call_path(cuda):
    """bit for bit mdct_fwd_u8_i16 followed by mdct_inv_i16_u8 on the device (the definition, include/mdct.h)"""
    torch = cuda
    for (W, H) in ((1024, 64), (520, 24), (8, 8)):
        for kind, seed in (("photo", 1), ("noise", 2)):
            src = synth.plane_u8_torch(W, H, kind, seed=synth.SEED + seed)
            for lut in (None, JPEG_LUMA, _lut(2000), np.full(64, 0.02, dtype=np.float32)):
                for shift in (True, False):
                    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
                    two = torch.empty_like(src)
                    api.fwd_u8_i16(src, coef, W, H, lut=lut, level_shift=shift)
                    api.inv_i16_u8(coef, two, W, H, lut=lut, level_shift=shift)
                    one = torch.full_like(src, CANARY)
                    api.roundtrip_u8(src, one, W, H, lut=lut, level_shift=shift)
                    assert torch.equal(one, two), (W, H, kind, shift)


@gpu
def test_roundtrip_u8_pitches_ranges_and_alignment(cuda):
    """pitched rows whose padding survives, block-row sub-ranges that leave the other rows alone, planes at every byte alignment"""
    import oracle as O

    torch = cuda
    W, H = 200, 56
    img = synth.plane_u8_np(W, H, "photo", seed=99)
    want = O.roundtrip_u8(img, W, H, lut=JPEG_CHROMA)
    for off_in, off_out, pad_in, pad_out in ((0, 0, 0, 0), (1, 3, 5, 11), (7, 2, 24, 8), (4, 5, 1, 3)):
        pin, pout = W + pad_in, W + pad_out
        buf_in = torch.full((H * pin + 16,), 0x11, dtype=torch.uint8, device="cuda")
        buf_out = torch.full((H * pout + 16,), CANARY, dtype=torch.uint8, device="cuda")
        vin = buf_in[off_in:off_in + H * pin].view(H, pin)
        vin[:, :W] = torch.from_numpy(img).cuda()
        vout = buf_out[off_out:off_out + H * pout]
        for by0, by1 in ((0, H // 8), (2, 5), (6, 7), (3, 3)):
            buf_out.fill_(CANARY)
            api.roundtrip_u8(vin, vout, W, H, lut=JPEG_CHROMA, by0=by0, by1=by1, pitch_in=pin, pitch_out=pout)
            got = buf_out.cpu().numpy()
            body = got[off_out:off_out + H * pout].reshape(H, pout)
            assert np.array_equal(body[by0 * 8:by1 * 8, :W], want[by0 * 8:by1 * 8]), (off_in, off_out, by0, by1)
            mask = np.ones_like(got, dtype=bool)
            rows = np.zeros((H, pout), dtype=bool)
            rows[by0 * 8:by1 * 8, :W] = True
            mask[off_out:off_out + H * pout] = ~rows.reshape(-1)
            assert (got[mask] == CANARY).all(), ("bytes outside the range were written", off_in, off_out, by0, by1)


def _out_pad(pad):
    return pad + 5 if pad else 0  # the output pitch differs from the input's (a paired plane's straddling tile hops by each side's own pitch)


def _u8_planes(torch, shapes, luts, pad=0, seed0=0):
    srcs, d_in, d_out = [], [], []
    for i, (w, h) in enumerate(shapes):
        s = synth.plane_u8_np(w, h, "photo" if i % 3 else "noise", seed=synth.SEED + seed0 + i)
        if pad:
            full = np.full((h, w + pad), 0x33, dtype=np.uint8)
            full[:, :w] = s
            s = full
        srcs.append(s)
        d_in.append(torch.from_numpy(s).cuda())
        d_out.append(torch.full((h, w + _out_pad(pad)), CANARY, dtype=torch.uint8, device="cuda"))
    desc = [(a, b, w, h, l, w + pad, w + _out_pad(pad)) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)]
    return srcs, d_in, d_out, desc


def _check_u8(srcs, d_out, shapes, luts, pad, shift, tag):
    import oracle as O

    for i, (s, o, (w, h), l) in enumerate(zip(srcs, d_out, shapes, luts)):
        got = o.cpu().numpy()
        want = O.roundtrip_u8(np.ascontiguousarray(s[:, :w]), w, h, lut=l, level_shift=shift)
        assert np.array_equal(got[:, :w], want), (tag, i, w, h)
        if pad:
            assert (got[:, w:] == CANARY).all(), (tag, i, "padding written")


@gpu
def test_u8_batch_mixed_shapes_and_tables(cuda):
    """separately allocated planes of different shapes (partial tiles, one-block planes), shared / distinct / no / wild tables, pitched rows:
    the no-allocation call and the device-resident batch give the checker's planes; more shapes than the compare chain; more planes than
    one argument block"""
    torch = cuda
    T = _tables()
    # (3840 x 16, 256 x 24, 768 x 40: rows ending in half a tile -- k_u8_batch tiles them over pairs of block rows, the odd last row alone)
    shapes = [(1920, 64), (8, 8), (72, 24), (520, 16), (256, 24), (200, 40), (3840, 16), (768, 40)]
    for luts, pad, shift in (([None] * 8, 0, True), ([JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA, None, JPEG_LUMA, _lut(10), None, T["edge"]], 24, True),
                             ([JPEG_LUMA, T["tiny"], JPEG_CHROMA, T["huge"], None, T["mixed"], JPEG_LUMA, JPEG_LUMA], 8, False)):
        for form in ("args", "device"):
            srcs, d_in, d_out, desc = _u8_planes(torch, shapes, luts, pad)
            if form == "args":
                api.roundtrip_u8_batch(desc, level_shift=shift)
            else:
                b = api.Batch("roundtrip_u8", desc, level_shift=shift)
                assert b.launches == 1
                b.run()
                b.run()
                b.close()
            torch.cuda.synchronize()
            _check_u8(srcs, d_out, shapes, luts, pad, shift, form)
    rng = np.random.default_rng(3)
    shapes = [(8 * int(rng.integers(1, 160)), 8 * int(rng.integers(1, 6))) for _ in range(70)]
    tabs = [JPEG_LUMA, JPEG_CHROMA, _lut(50), _lut(500), T["ones"]]
    luts = [tabs[i % 5] if i % 7 else None for i in range(70)]
    for form in ("args", "device"):
        srcs, d_in, d_out, desc = _u8_planes(torch, shapes, luts, 8, seed0=300)
        if form == "args":
            api.roundtrip_u8_batch(desc)
        else:
            b = api.Batch("roundtrip_u8", desc)
            assert b.launches == 1
            b.run()
        torch.cuda.synchronize()
        _check_u8(srcs, d_out, shapes, luts, 8, True, form)


@gpu
def test_round_trips_in_place(cuda):
    """to == from (include/mdct.h): the 8-bit and the int16 fused round trips, single plane and batch, give the bytes of the out-of-place call"""
    torch = cuda
    for (W, H) in ((1032, 72), (512, 8)):
        src8 = synth.plane_u8_torch(W, H, "noise", seed=5)
        want8 = torch.empty_like(src8)
        api.roundtrip_u8(src8, want8, W, H, lut=JPEG_LUMA)
        buf8 = src8.clone()
        api.roundtrip_u8(buf8, buf8, W, H, lut=JPEG_LUMA)
        assert torch.equal(buf8, want8)
        b1, b2 = src8.clone(), src8.clone()
        api.roundtrip_u8_batch([(b1, b1, W, H, JPEG_LUMA), (b2, b2, W, H, JPEG_LUMA)])
        assert torch.equal(b1, want8) and torch.equal(b2, want8)
        src16 = synth.plane_i16_torch(W, H, "photo", seed=6, bits=12)
        for lut in (None, _lut(30)):
            want16 = torch.empty_like(src16)
            api.roundtrip_i16(src16, want16, W, H, lut=lut)
            buf16 = src16.clone()
            api.roundtrip_i16(buf16, buf16, W, H, lut=lut)
            assert torch.equal(buf16, want16)
            c1 = src16.clone()
            api.i16_batch("roundtrip", [(c1, c1, W, H, lut)])
            assert torch.equal(c1, want16)


@gpu
def test_u8_batch_empty_lists_and_empty_planes(cuda):
    """nothing to do is not an error: an empty list, planes without blocks between real ones, an empty row range; nothing is written"""
    torch = cuda
    lib = _lib.load()
    assert api.roundtrip_u8_batch([]) == 0
    b = api.Batch("roundtrip_u8", [])
    assert b.launches == 0 and b.run() == 0
    b.close()
    src = synth.plane_u8_torch(264, 16, "photo", seed=3)
    dst = torch.full_like(src, CANARY)
    none = torch.full((8, 8), CANARY, dtype=torch.uint8, device="cuda")
    api.roundtrip_u8_batch([(src, none, 0, 16, None, 264, 264), (src, dst, 264, 16, JPEG_LUMA), (src, none, 264, 0, None)])
    api.roundtrip_u8(src, none, 264, 16, by0=1, by1=1)
    torch.cuda.synchronize()
    import oracle as O

    assert (none == CANARY).all() and np.array_equal(dst.cpu().numpy(), O.roundtrip_u8(src.cpu().numpy(), 264, 16, lut=JPEG_LUMA))
    assert lib.mdct_batch_launches(None) == 0


@gpu
def test_u8_batch_is_graph_capturable(cuda):
    torch = cuda
    shapes = [(1920, 32), (960, 16), (960, 16)]
    luts = [JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA]
    srcs, d_in, d_out, desc = _u8_planes(torch, shapes, luts)
    b = api.Batch("roundtrip_u8", desc)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.run(stream=s)
        api.roundtrip_u8_batch(desc, stream=s)
    s.synchronize()
    for form in ("device", "args"):
        for o in d_out:
            o.fill_(CANARY)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            if form == "device":
                b.run(stream=s)
            else:
                api.roundtrip_u8_batch(desc, stream=s)
        g.replay()
        torch.cuda.synchronize()
        _check_u8(srcs, d_out, shapes, luts, 0, True, "graph/" + form)


@gpu
def test_config3_frame_u8_in_one_launch(cuda):
    """BASELINE.json configs[2] as SURVEY.md 8(d) states it: Y 7680x4320 + Cb/Cr 3840x2160 8-bit planes, Annex-K tables, forward ->
    quantise -> dequantise -> inverse, ONE launch, 99,532,800 bytes.  Every plane in full against the threaded checker; the no-allocation
    call, the device batch and the two-call path on the device agree byte for byte; the reconstruction is a JPEG-quality picture."""
    import oracle as O

    torch = cuda
    shapes = [(w, h) for w, h, _, _ in synth.CONFIG3_PLANES]
    luts = [JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA]
    d_in = [synth.plane_u8_torch(w, h, "photo", seed=synth.SEED + k) for w, h, k, _ in synth.CONFIG3_PLANES]
    outs = {}
    for form in ("args", "device"):
        d_out = [torch.full_like(t, CANARY) for t in d_in]
        desc = [(a, b, w, h, l) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)]
        if form == "args":
            api.roundtrip_u8_batch(desc)
        else:
            b = api.Batch("roundtrip_u8", desc)
            assert b.launches == 1
            b.run()
        outs[form] = d_out
    torch.cuda.synchronize()
    threads = O.host_threads()
    for i, ((w, h), l) in enumerate(zip(shapes, luts)):
        assert torch.equal(outs["args"][i], outs["device"][i]), i
        coef = torch.empty((h, w), dtype=torch.int16, device="cuda")
        two = torch.empty_like(d_in[i])
        api.fwd_u8_i16(d_in[i], coef, w, h, lut=l)
        api.inv_i16_u8(coef, two, w, h, lut=l)
        assert torch.equal(two, outs["args"][i]), ("two-call path", i)
        src = d_in[i].cpu().numpy()
        want = O.roundtrip_u8(src, w, h, lut=l, threads=threads)
        got = outs["args"][i].cpu().numpy()
        assert np.array_equal(got, want), ("oracle", i)
        err = got.astype(np.int32) - src.astype(np.int32)
        # the synthetic "photo" carries +-24 levels of uniform noise (std 14.1) that the Annex-K tables quantise away: the error is that noise
        assert np.sqrt((err ** 2).mean()) < 16.0, i


# ----------------------------------------------------------------------------------------------- the two halves on plane batches
def test_u8_i16_batch_status_codes_without_device():
    G.build_hip()
    lib = _lib.load()
    px = np.zeros(64 * 16, dtype=np.uint8)
    co = np.zeros(64 * 16 + 8, dtype=np.int16)
    ok = (px, co, 64, 16, None)
    for mode in ("fwd", "inv"):
        assert api.u8_i16_batch(mode, [ok, (px, None, 64, 16, None)], check=False) == 1  # null pointer
        assert api.u8_i16_batch(mode, [ok, (px, co, 60, 16, None)], check=False) == 2  # not a multiple of 8x8
        assert api.u8_i16_batch(mode, [ok, (px, co[1:], 64, 8, None)], check=False) == 1 and "16-byte" in api.last_error()  # coefficient rows misaligned
        assert api.u8_i16_batch(mode, [(px, co, 64, 16, None, 64, 68)], check=False) == 1  # coefficient pitch not a multiple of 8 elements
    assert lib.mdct_fwd_u8_i16_batch(None, 1, 1, None) == 1
    h = ctypes.c_void_p()
    arr, _keep = api._plane_array([ok])
    assert lib.mdct_batch_create_u8_i16(ctypes.byref(h), 2, arr, 1, 1) == 1 and not h  # MDCT_MODE_ROUNDTRIP is not a mode of this call


@gpu
def test_u8_i16_batches_equal_the_single_plane_calls_and_the_oracle(cuda):
    """mdct_fwd_u8_i16_batch / mdct_inv_i16_u8_batch: mixed shapes (partial tiles), pitches on both sides, shared / distinct / no / wild tables, both
    level shifts, both forms (kernel arguments, device table): every plane equals mdct_fwd_u8_i16 / mdct_inv_i16_u8 on the device and the checker;
    the inverse is fed arbitrary int16 coefficients, extremes included (its output stage clamps)"""
    import oracle as O

    torch = cuda
    T = _tables()
    shapes = [(1920, 32), (8, 8), (72, 24), (520, 16), (256, 24), (768, 40), (3840, 16)]  # the last three: paired rows (kDescPaired)
    rng = np.random.default_rng(8)
    for luts, pad, shift in (([None] * 7, 0, True), ([JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA, None, T["tiny"], _lut(10), T["mixed"]], 16, True), ([T["huge"], JPEG_LUMA, T["ones"], T["sixteenth"], None, T["negative"], T["edge"]], 8, False)):
        for form in ("args", "device"):
            px_np = [synth.plane_u8_np(w, h, "photo" if i % 2 else "noise", seed=60 + i) for i, (w, h) in enumerate(shapes)]
            px = [torch.from_numpy(np.pad(a, ((0, 0), (0, pad)), constant_values=9)).cuda() for a in px_np]
            coef = [torch.full((h, w + pad), -21846, dtype=torch.int16, device="cuda") for (w, h) in shapes]
            desc = [(p, c, w, h, l, w + pad, w + pad) for p, c, (w, h), l in zip(px, coef, shapes, luts)]
            if form == "args":
                api.u8_i16_batch("fwd", desc, level_shift=shift)
            else:
                b = api.Batch("fwd_u8_i16", desc, level_shift=shift)
                assert b.launches == 1
                b.run()
                b.close()
            torch.cuda.synchronize()
            for i, ((w, h), l) in enumerate(zip(shapes, luts)):
                got = coef[i].cpu().numpy()
                assert np.array_equal(got[:, :w], O.u8_i16("fwd", px_np[i], w, h, lut=l, level_shift=shift)) and (got[:, w:] == -21846).all(), ("fwd", form, i)
                single = torch.empty((h, w), dtype=torch.int16, device="cuda")
                api.fwd_u8_i16(torch.from_numpy(px_np[i]).cuda(), single, w, h, lut=l, level_shift=shift)
                assert torch.equal(single, coef[i][:, :w]), ("fwd vs single", i)
            # the inverse on arbitrary coefficients
            co_np = [rng.integers(-32768, 32768, (h, w), dtype=np.int16) if i % 3 == 0 else rng.integers(-600, 600, (h, w), dtype=np.int16) for i, (w, h) in enumerate(shapes)]
            co = [torch.from_numpy(np.pad(a, ((0, 0), (0, pad)), constant_values=5)).cuda() for a in co_np]
            out = [torch.full((h, w + pad), CANARY, dtype=torch.uint8, device="cuda") for (w, h) in shapes]
            idesc = [(o, c, w, h, l, w + pad, w + pad) for o, c, (w, h), l in zip(out, co, shapes, luts)]
            if form == "args":
                api.u8_i16_batch("inv", idesc, level_shift=shift)
            else:
                b = api.Batch("inv_i16_u8", idesc, level_shift=shift)
                assert b.launches == 1
                b.run()
                b.close()
            torch.cuda.synchronize()
            for i, ((w, h), l) in enumerate(zip(shapes, luts)):
                got = out[i].cpu().numpy()
                assert np.array_equal(got[:, :w], O.u8_i16("inv", co_np[i], w, h, lut=l, level_shift=shift)) and (got[:, w:] == CANARY).all(), ("inv", form, i)
                single = torch.empty((h, w), dtype=torch.uint8, device="cuda")
                api.inv_i16_u8(torch.from_numpy(co_np[i]).cuda(), single, w, h, lut=l, level_shift=shift)
                assert torch.equal(single, out[i][:, :w]), ("inv vs single", i)


@gpu
def test_config3_frame_forward_then_inverse_batch_equals_the_fused_round_trip(cuda):
    """the configs[2] frame at full size: one forward launch (pixels -> quantised coefficients of Y, Cb, Cr), one inverse launch, and the
    fused 8-bit round trip give the same pixels; the coefficients equal the checker's (chroma plane in full)"""
    import oracle as O

    torch = cuda
    luts = [JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA]
    px = [synth.plane_u8_torch(w, h, "photo", seed=synth.SEED + k) for w, h, k, _ in synth.CONFIG3_PLANES]
    coef = [torch.empty((h, w), dtype=torch.int16, device="cuda") for w, h, _, _ in synth.CONFIG3_PLANES]
    back = [torch.full_like(p, CANARY) for p in px]
    fused = [torch.full_like(p, CANARY) for p in px]
    dims = [(w, h) for w, h, _, _ in synth.CONFIG3_PLANES]
    fb = api.Batch("fwd_u8_i16", [(p, c, w, h, l) for p, c, (w, h), l in zip(px, coef, dims, luts)])
    ib = api.Batch("inv_i16_u8", [(o, c, w, h, l) for o, c, (w, h), l in zip(back, coef, dims, luts)])
    assert fb.launches == 1 and ib.launches == 1
    fb.run()
    ib.run()
    api.roundtrip_u8_batch([(p, o, w, h, l) for p, o, (w, h), l in zip(px, fused, dims, luts)])
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(back[i], fused[i]), i
    w, h = dims[1]
    assert np.array_equal(coef[1].cpu().numpy(), O.u8_i16("fwd", px[1].cpu().numpy(), w, h, lut=JPEG_CHROMA))


@gpu
def test_u8_i16_batches_empty_many_planes_and_graph_capture(cuda):
    """the two halves: an empty list and planes without blocks are not errors and write nothing; 70 planes with seven tables (more than the kernel
    arguments hold: the argument form splits into several launches, the device form stays one); both forms captured into a graph and replayed"""
    import oracle as O

    torch = cuda
    assert api.u8_i16_batch("fwd", []) == 0 and api.u8_i16_batch("inv", []) == 0
    for mode in ("fwd_u8_i16", "inv_i16_u8"):
        b = api.Batch(mode, [])
        assert b.launches == 0 and b.run() == 0
        b.close()
    px = synth.plane_u8_torch(264, 16, "photo", seed=3)
    co = torch.full((16, 264), -21846, dtype=torch.int16, device="cuda")
    none_c = torch.full((8, 8), -21846, dtype=torch.int16, device="cuda")
    api.u8_i16_batch("fwd", [(px, none_c, 0, 16, None, 264, 8), (px, co, 264, 16, JPEG_LUMA), (px, none_c, 264, 0, None, 264, 264)])
    torch.cuda.synchronize()
    assert (none_c == -21846).all() and np.array_equal(co.cpu().numpy(), O.u8_i16("fwd", px.cpu().numpy(), 264, 16, lut=JPEG_LUMA))

    rng = np.random.default_rng(4)
    shapes = [(8 * int(rng.integers(1, 160)), 8 * int(rng.integers(1, 6))) for _ in range(70)]
    tabs = [JPEG_LUMA, JPEG_CHROMA, _lut(50), _lut(500), _tables()["ones"], _lut(7), _tables()["tiny"]]
    luts = [tabs[i % 7] if i % 9 else None for i in range(70)]
    px_np = [synth.plane_u8_np(w, h, "noise", seed=400 + i) for i, (w, h) in enumerate(shapes)]
    pxs = [torch.from_numpy(a).cuda() for a in px_np]
    want_c = [O.u8_i16("fwd", a, w, h, lut=l) for a, (w, h), l in zip(px_np, shapes, luts)]
    want_p = [O.u8_i16("inv", c, w, h, lut=l) for c, (w, h), l in zip(want_c, shapes, luts)]
    s = torch.cuda.Stream()
    for form in ("args", "device", "graph/args", "graph/device"):
        cos = [torch.full((h, w), -21846, dtype=torch.int16, device="cuda") for (w, h) in shapes]
        back = [torch.full((h, w), CANARY, dtype=torch.uint8, device="cuda") for (w, h) in shapes]
        fdesc = [(p, c, w, h, l) for p, c, (w, h), l in zip(pxs, cos, shapes, luts)]
        idesc = [(o, c, w, h, l) for o, c, (w, h), l in zip(back, cos, shapes, luts)]
        fb, ib = api.Batch("fwd_u8_i16", fdesc), api.Batch("inv_i16_u8", idesc)
        assert fb.launches == 1 and ib.launches == 1

        def both(stream=None):
            if form.endswith("args"):
                api.u8_i16_batch("fwd", fdesc, stream=stream)
                api.u8_i16_batch("inv", idesc, stream=stream)
            else:
                fb.run(stream=stream)
                ib.run(stream=stream)

        if form.startswith("graph"):
            with torch.cuda.stream(s):
                both(s)  # tables seen once outside the capture
            s.synchronize()
            for t in cos:
                t.fill_(-21846)
            for t in back:
                t.fill_(CANARY)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                both(s)
            g.replay()
        else:
            both()
        torch.cuda.synchronize()
        for i in range(70):
            assert np.array_equal(cos[i].cpu().numpy(), want_c[i]), (form, "fwd", i, shapes[i])
            assert np.array_equal(back[i].cpu().numpy(), want_p[i]), (form, "inv", i, shapes[i])
        fb.close()
        ib.close()
