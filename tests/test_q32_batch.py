"""The reference's primary product on a plane list (mdct_fwd_quant32_u8_batch, mdct_batch_create_q32): per plane exactly
simdDCT_EncodeQuantize32ReorderBuffer's AVX2-tier bytes (simd_dct.cpp:2064-2262, the designated oracle a3) over every block row, any number of
separately allocated planes with their own tables in one launch.  Parity is pinned by the reference: the checker's q32 restatement is the one
tests/test_oracle_vs_reference.py holds against the real reference build.

CPU: status codes in the reference dispatcher's order, decided before a device is touched.
GPU (-m gpu): bit-exact against the checker and the single-plane call over mixed shapes (partial tiles), input pitches, pitched output strips with
canaries, ordinary and extreme tables (the exact-convert build), both forms, many planes, graph capture; a full 8K 4:2:0 frame in one launch."""
import ctypes

import numpy as np
import pytest

import __graft_entry__ as G
from simd_dct_amd import _lib, api, synth

gpu = pytest.mark.gpu
CANARY = 0xA5


@pytest.fixture(scope="module")
def cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    api.init(0)
    return torch


def _lut(scale):
    return (api.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)


def test_q32_batch_status_codes_without_device():
    G.build_hip()
    lib = _lib.load()
    px = np.zeros(128 * 16, dtype=np.uint8)
    out = np.zeros(128 * 16, dtype=np.uint8)
    lut = _lut(2000)
    ok = (px, out, 128, 16, lut)
    assert api.fwd_quant32_u8_batch([ok, (px, None, 128, 16, lut)], check=False) == 1  # null pointer -> 1 before any shape test (simd_dct.cpp:115)
    assert api.fwd_quant32_u8_batch([ok, (px, out, 128, 16, None)], check=False) == 1  # the table is required
    assert api.fwd_quant32_u8_batch([ok, (px, out, 96, 16, lut)], check=False) == 2 and "64" in api.last_error()  # sizeX % 64 (:117)
    assert api.fwd_quant32_u8_batch([ok, (px, out, 128, 12, lut)], check=False) == 2
    assert api.fwd_quant32_u8_batch([(None, out, 96, 16, lut)], check=False) == 1  # null wins over shape, as in the reference
    assert api.fwd_quant32_u8_batch([(px, out, 128, 16, lut, 128, 8 * 128 - 16)], check=False) == 1 and "strip pitch" in api.last_error()
    assert api.fwd_quant32_u8_batch([(px, out, 128, 16, lut, 128, 8 * 128 + 8)], check=False) == 1  # not a multiple of 16
    assert api.fwd_quant32_u8_batch([(px, out, 128, 16, lut, 120, None)], check=False) == 1  # input pitch below the width
    # not in place: an output that meets the plane's own input is refused -- the same buffer, and a partial overlap
    assert api.fwd_quant32_u8_batch([ok, (px, px, 128, 16, lut)], check=False) == 1 and "overlaps" in api.last_error()
    both = np.zeros(2 * 128 * 16, dtype=np.uint8)
    assert api.fwd_quant32_u8_batch([(both[:2048], both[1024:3072], 128, 16, lut)], check=False) == 1 and "overlaps" in api.last_error()
    assert api.fwd_quant32_u8_batch([(both[:2048], both[2048:], 128, 16, lut)], check=False) != 1 or "overlaps" not in api.last_error()  # adjacent is fine
    assert lib.mdct_fwd_quant32_u8_batch(None, 1, None) == 1
    h = ctypes.c_void_p()
    assert lib.mdct_batch_create_q32(None, None, 0) == 1


def _planes(torch, shapes, luts, pad_in, pad_out, seed0=0):
    """device inputs (pitched), canary-filled pitched outputs, descriptors, host copies of the inputs"""
    srcs, d_in, d_out, desc = [], [], [], []
    for i, ((w, h), lut) in enumerate(zip(shapes, luts)):
        a = synth.plane_u8_np(w, h, "photo" if i % 2 else "noise", seed=seed0 + i)
        srcs.append(a)
        d_in.append(torch.from_numpy(np.pad(a, ((0, 0), (0, pad_in)), constant_values=3)).cuda())
        d_out.append(torch.full((h // 8, 8 * w + pad_out), CANARY, dtype=torch.uint8, device="cuda"))
        desc.append((d_in[-1], d_out[-1], w, h, lut, w + pad_in, 8 * w + pad_out))
    return srcs, d_in, d_out, desc


def _check(srcs, d_out, shapes, luts, what):
    import oracle as O

    for i, (a, o, (w, h), lut) in enumerate(zip(srcs, d_out, shapes, luts)):
        rc, want = O.q32_native(a, lut, w, h, 0, h // 8)
        got = o.cpu().numpy()
        assert rc == 0 and np.array_equal(got[:, : 8 * w].reshape(-1), want), (what, i, w, h)
        assert (got[:, 8 * w:] == CANARY).all(), (what, i, "gap between the strips written")


@gpu
def test_q32_batch_equals_the_checker_and_the_single_plane_call(cuda):
    torch = cuda
    # (3840 x 16, 256 x 24, 768 x 40: rows that end in half a tile -- tiled over PAIRS of block rows, kDescPaired, the odd last row alone)
    shapes = [(1920, 32), (64, 8), (3840, 16), (512, 24), (4160, 8), (128, 40), (7680, 8), (256, 24), (768, 40)]
    wild = np.full(64, 1e-4, dtype=np.float32)  # 255 / (lut * 0.95) beyond 2^17: the exact-convert build for the whole call
    wild[5] = np.float32(np.inf)
    for luts, pad_in, pad_out in (([_lut(2000)] * 9, 0, 0), ([_lut(2000), _lut(8), _lut(100), -_lut(40), _lut(2000), _lut(1e5), _lut(0.9), _lut(8), _lut(300)], 24, 64),
                                  ([_lut(2000), wild, _lut(8), _lut(100), wild, _lut(3), _lut(2000), wild, _lut(50)], 8, 16)):
        for form in ("args", "device"):
            srcs, d_in, d_out, desc = _planes(torch, shapes, luts, pad_in, pad_out, seed0=70)
            if form == "args":
                api.fwd_quant32_u8_batch(desc)
            else:
                b = api.Batch("q32", desc)
                assert b.launches == 1
                b.run()
                b.run()
                b.close()
            torch.cuda.synchronize()
            _check(srcs, d_out, shapes, luts, form)
            for i, ((w, h), lut) in enumerate(zip(shapes, luts)):  # the call it batches
                single = torch.full_like(d_out[i], CANARY)
                api.fwd_quant_u8(d_in[i], single, lut, w, h, 0, h // 8, pitch_in=w + pad_in, pitch_out=8 * w + pad_out)
                assert torch.equal(single, d_out[i]), (form, "vs mdct_fwd_quant_u8_pitched", i)


@gpu
def test_q32_batch_empty_many_planes_and_graph_capture(cuda):
    torch = cuda
    assert api.fwd_quant32_u8_batch([]) == 0
    b = api.Batch("q32", [])
    assert b.launches == 0 and b.run() == 0
    b.close()
    src = synth.plane_u8_torch(256, 16, "photo", seed=3)
    none = torch.full((64,), CANARY, dtype=torch.uint8, device="cuda")
    api.fwd_quant32_u8_batch([(src, none, 0, 16, _lut(9)), (src, none, 256, 0, _lut(9))])  # planes without blocks: nothing to do
    torch.cuda.synchronize()
    assert (none == CANARY).all()

    rng = np.random.default_rng(6)
    shapes = [(64 * int(rng.integers(1, 20)), 8 * int(rng.integers(1, 6))) for _ in range(70)]
    tabs = [_lut(s) for s in (2000, 8, 100, 40, 3, 700, 15)]
    luts = [tabs[i % 7] for i in range(70)]
    s = torch.cuda.Stream()
    for form in ("args", "device", "graph/args", "graph/device"):
        srcs, d_in, d_out, desc = _planes(torch, shapes, luts, 8, 32, seed0=500)
        b = api.Batch("q32", desc)
        assert b.launches == 1

        def run(stream=None):
            if form.endswith("args"):
                api.fwd_quant32_u8_batch(desc, stream=stream)  # 70 planes, 7 tables: several launches in this form
            else:
                b.run(stream=stream)

        if form.startswith("graph"):
            with torch.cuda.stream(s):
                run(s)
            s.synchronize()
            for o in d_out:
                o.fill_(CANARY)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run(s)
            g.replay()
        else:
            run()
        torch.cuda.synchronize()
        _check(srcs, d_out, shapes, luts, form)
        b.close()


@gpu
def test_q32_frame_420_in_one_launch(cuda):
    """an 8K 4:2:0 frame (Y 7680x4320 + Cb/Cr 3840x2160: BASELINE.json configs[2]'s planes) as the reference's q32 product, one launch where the
    reference's caller makes three calls: every plane in full against the threaded checker and against the single-plane call on the device"""
    import oracle as O

    torch = cuda
    shapes = [(w, h) for w, h, _, _ in synth.CONFIG3_PLANES]
    luts = [_lut(2000), _lut(1200), _lut(1200)]
    d_in = [synth.plane_u8_torch(w, h, "photo", seed=synth.SEED + k) for w, h, k, _ in synth.CONFIG3_PLANES]
    outs = {}
    for form in ("args", "device"):
        d_out = [torch.full((w * h,), CANARY, dtype=torch.uint8, device="cuda") for (w, h) in shapes]
        desc = [(a, o, w, h, l) for a, o, (w, h), l in zip(d_in, d_out, shapes, luts)]
        if form == "args":
            api.fwd_quant32_u8_batch(desc)
        else:
            b = api.Batch("q32", desc)
            assert b.launches == 1
            b.run()
            b.close()
        torch.cuda.synchronize()
        outs[form] = d_out
    for i, ((w, h), lut) in enumerate(zip(shapes, luts)):
        want = O.q32_native_par(d_in[i].cpu().numpy(), lut, w, h)
        assert np.array_equal(outs["args"][i].cpu().numpy(), want), i
        assert torch.equal(outs["args"][i], outs["device"][i])
        single = torch.empty_like(outs["args"][i])
        api.fwd_quant_u8(d_in[i], single, lut, w, h, 0, h // 8)
        assert torch.equal(single, outs["args"][i])
