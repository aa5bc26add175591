"""Maximum-size GPU tests (-m gpu): planes whose byte offsets do not fit 32 bits.

The input is a short strip of rows repeated down the plane, so the output is periodic too: every
period must equal the first one (compared on the device), and the first one must equal the oracle's
result for a plane that is just that strip.  5.5 GB per plane on a 288 GB card; a 32-bit offset
anywhere in a kernel shows up as a period past the 4 GiB mark that differs from the first."""
import numpy as np
import pytest

import oracle as O
import simd_dct_amd as M
from simd_dct_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CANARY = 0xA5
P = 128  # pixel rows per period: 16 block rows, 8 stereo double rows


@pytest.fixture(scope="module", autouse=True)
def device():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    M.init(0)
    yield
    torch.cuda.empty_cache()


def lut_x(scale):
    return (M.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)


def all_periods_equal_first(t, n_per):
    v = t.reshape(n_per, -1)
    return bool((v == v[0]).all().item())


def test_u8_products_beyond_4GiB():
    """all five reference behaviours through the native C-ABI on a 32768 x 167936 plane (5.5 GB in, 5.5 GB out)"""
    W, n_per = 32768, 1312
    H = P * n_per
    assert W * H > 2**32 + 2**30
    strip = synth.plane_u8_np(W, P, "photo", seed=11)
    src = torch.from_numpy(strip).cuda().repeat(n_per, 1).contiguous()
    assert src.numel() == W * H
    out = torch.empty(W * H, dtype=torch.uint8, device="cuda")
    # the oracle on a plane that is 2.25 periods tall: the reference semantics (top half only) then cover the first period
    Hs = 2 * P + 32
    small = np.ascontiguousarray(np.tile(strip, (3, 1))[:Hs])
    for beh, layout, profile, scale in (("q32_avx", M.LAYOUT_Q32, M.PROFILE_REF_AVX, 2000.0), ("stereo_sse", M.LAYOUT_STEREO, M.PROFILE_REF_SSE, 8.0),
                                        ("stereo_scalar", M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, 8.0), ("encq_sse", M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, 8.0),
                                        ("encq_scalar", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, 8.0)):
        lut = lut_x(scale)
        out.fill_(CANARY)
        rows = H // 16 if layout == M.LAYOUT_STEREO else H // 8
        M.fwd_quant_u8(src, out, lut, W, H, 0, rows, layout=layout, profile=profile)
        torch.cuda.synchronize()
        if layout == M.LAYOUT_STEREO:
            # two images stacked (the second starts at row H/2, a whole number of periods down); coefficient c of
            # stream position p = (double row * 2 + eye) * W/8 + block sits at c * (W*H/64) + p: 64 planes, each
            # periodic in p with 2 * 16 block rows; the oracle's plane is two periods tall = one per eye
            per_pos = 2 * W * P // 64
            planes = out.reshape(64, n_per // 2, per_pos)
            assert bool((planes == planes[:, :1]).all().item()), beh
            want = np.full(W * 2 * P, CANARY, dtype=np.uint8)
            rc, want = O.run_behaviour(beh, small[: 2 * P], lut, W, 2 * P, 0, 2 * P, out=want)
            assert rc == 0
            assert np.array_equal(planes[:, 0].cpu().numpy(), want.reshape(64, per_pos)), beh
            continue
        want = np.full(W * Hs, CANARY, dtype=np.uint8)
        rc, want = O.run_behaviour(beh, small, lut, W, Hs, 0, 2 * P, out=want)
        assert rc == 0
        assert all_periods_equal_first(out, n_per), beh
        assert np.array_equal(out[: W * P].cpu().numpy(), want[: W * P]), beh
    del src, out


def test_f32_beyond_4GiB():
    """float32 forward on 16384 x 83968 (5.5 GB in, 5.5 GB out)"""
    W, n_per = 16384, 656
    H = P * n_per
    assert W * H * 4 > 2**32 + 2**30
    strip = (synth.plane_i16_np(W, P, "photo", seed=12).astype(np.float32) / np.float32(128))
    src = torch.from_numpy(strip).cuda().repeat(n_per, 1).contiguous()
    out = torch.empty_like(src)
    M.fwd_f32(src, out, W, H)
    torch.cuda.synchronize()
    assert all_periods_equal_first(out, n_per)
    want = O.f32("fwd", strip, W, P)
    assert np.array_equal(out[:P].cpu().numpy().view(np.uint32), want.view(np.uint32))
    M.inv_f32(out, src, W, H)  # and back, into the input buffer
    torch.cuda.synchronize()
    assert all_periods_equal_first(src, n_per)
    assert np.array_equal(src[:P].cpu().numpy().view(np.uint32), O.f32("inv", want, W, P).view(np.uint32))
    del src, out


def test_codec_stages_beyond_4GiB():
    """int16 forward + table -> zig-zag + run/level records -> Huffman rows on 8192 x 335872
    (5.5 GB of coefficients, 8.3 GB of records, 8.9 GB of segment space)"""
    W, n_per = 8192, 2624
    H = P * n_per
    assert W * H * 2 > 2**32 + 2**30
    lut = lut_x(60)
    strip = synth.plane_i16_np(W, P, "photo", seed=13)
    src = torch.from_numpy(strip).cuda().repeat(n_per, 1).contiguous()
    coef = torch.empty_like(src)
    M.fwd_i16(src, coef, W, H, lut=lut)
    torch.cuda.synchronize()
    assert all_periods_equal_first(coef, n_per)
    want_coef = O.i16("fwd", strip, W, P, lut=lut)
    assert np.array_equal(coef[:P].cpu().numpy(), want_coef)
    del src
    nblk = (W // 8) * (H // 8)
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    M.zigzag_rle_i16(coef, W, H, lv, rn, ct)
    torch.cuda.synchronize()
    del coef
    # levels / runs beyond a block's count are unspecified padding only if the kernel leaves them alone: it writes whole records
    wl, wr, wc = O.zigzag_rle("i16", want_coef, W, P)
    per_blk = (W // 8) * (P // 8)
    assert all_periods_equal_first(ct, n_per) and np.array_equal(ct[:per_blk].cpu().numpy(), wc)
    assert all_periods_equal_first(lv, n_per) and all_periods_equal_first(rn, n_per)
    got_l, got_r = lv[:per_blk].cpu().numpy(), rn[:per_blk].cpu().numpy()
    valid = np.arange(64)[None, :] < wc[:, None]
    assert np.array_equal(got_l[valid], wl[valid]) and np.array_equal(got_r[valid], wr[valid])
    stride = M.huffman_seg_stride(W)
    seg = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    nb = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
    assert seg.numel() > 2**33
    M.huffman_rows(lv, rn, ct, W, H, seg, nb)
    torch.cuda.synchronize()
    assert all_periods_equal_first(nb, n_per) and all_periods_equal_first(seg, n_per)
    ws, wn, wstride = O.huffman_rows(wl, wr, wc, W, P, fill=0x5A)
    assert wstride == stride
    got_n, got_s = nb[: P // 8].cpu().numpy().astype(np.uint32), seg[: (P // 8) * stride].cpu().numpy()
    assert np.array_equal(got_n, wn)
    for r in range(P // 8):
        assert np.array_equal(got_s[r * stride:r * stride + wn[r]], ws[r * stride:r * stride + wn[r]]), r
