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


def test_operands_straddling_a_4GiB_address_boundary():
    """address arithmetic, not sizes: each operand in turn of the byte-source scans, the fused records kernel, the Huffman
    rows and the scan packer is placed so that its low 32 address bits cross 0x80000000 or wrap through 0 inside the
    operand (a kernel that keeps half an address in 32 bits, or lets one sign-extend, faults or reads elsewhere)"""
    arena = torch.zeros((5 << 30,), dtype=torch.uint8, device="cuda")  # contains one address of every residue mod 2^32
    a0 = arena.data_ptr()

    def hot(nbytes, into):
        """a view of `nbytes` of the arena whose middle (rounded to 256 B) sits at an address == `into` (mod 2^32)"""
        want = (into - (nbytes // 2 // 256) * 256) % (1 << 32)
        start = a0 + ((want - a0) % (1 << 32))
        assert start + nbytes <= a0 + arena.numel()
        v = arena[start - a0:start - a0 + nbytes]
        assert v.data_ptr() % (1 << 32) > (v.data_ptr() + nbytes - 1) % (1 << 32) or (v.data_ptr() % (1 << 32) < (1 << 31) <= (v.data_ptr() + nbytes - 1) % (1 << 32))
        return v

    def operands(sizes, which, into):
        """device byte buffers of the given sizes: number `which` straddles the boundary, the others are ordinary allocations"""
        return [hot(n, into) if k == which else torch.zeros((n,), dtype=torch.uint8, device="cuda") for k, n in enumerate(sizes)]

    W, H = 1024, 512
    nblk = (W // 8) * (H // 8)
    ctb = (nblk + 255) // 256 * 256
    lut8, lut2000, lut100 = lut_x(8), lut_x(2000), lut_x(100)
    img_np = synth.plane_u8_np(W, H, "photo", seed=17)
    d_img = torch.from_numpy(img_np.reshape(-1)).cuda()
    stride = M.huffman_seg_stride(W)
    wl, wr, wc = O.u8_records(img_np, W, H, lut=lut100)
    ws, wn, _ = O.huffman_rows(wl, wr, wc, W, H)
    wo, woff = O.jpeg_pack_rows(ws, wn, stride)
    d_lv, d_rn, d_ct = torch.from_numpy(wl).cuda(), torch.from_numpy(wr).cuda(), torch.from_numpy(wc).cuda()
    d_seg, d_nb = torch.from_numpy(ws).cuda(), torch.from_numpy(wn.view(np.int32)).cuda()
    for into in (1 << 31, 0):
        # the three byte layouts as sources of the scan: source, levels, runs, counts in turn on the boundary
        for kind, layout, profile, lut, rows in (("q32", M.LAYOUT_Q32, M.PROFILE_REF_AVX, lut2000, H // 8), ("stereo", M.LAYOUT_STEREO, M.PROFILE_REF_SSE, lut8, H // 16),
                                                 ("block", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, lut8, H // 8)):
            coded0 = torch.empty(W * H, dtype=torch.uint8, device="cuda")
            M.fwd_quant_u8(d_img, coded0, lut, W, H, 0, rows, layout=layout, profile=profile)
            want = O.zigzag_rle(kind, coded0.cpu().numpy(), W, H)
            for which in range(4):
                coded, lvb, rn, ctp = operands((W * H, nblk * 128, nblk * 64, ctb), which, into)
                coded.copy_(coded0)
                lv, rn, ct = lvb.view(torch.int16).reshape(nblk, 64), rn.reshape(nblk, 64), ctp[:nblk]
                if kind == "q32":
                    M.zigzag_rle_q32(coded, W, H, lv, rn, ct)
                else:
                    M.zigzag_rle_u8(coded, layout, W, H, lv, rn, ct)
                for got, w in zip((lv, rn, ct), want):
                    assert np.array_equal(got.cpu().numpy(), w), (kind, which, hex(into))
        # the forward products themselves, input and output in turn
        for beh, layout, profile, lut, rows in (("q32_avx", M.LAYOUT_Q32, M.PROFILE_REF_AVX, lut2000, H // 8), ("stereo_sse", M.LAYOUT_STEREO, M.PROFILE_REF_SSE, lut8, H // 16),
                                                ("encq_sse", M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, lut8, H // 8), ("encq_scalar", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, lut8, H // 8)):
            ref = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
            M.fwd_quant_u8(d_img, ref, lut, W, H, 0, rows, layout=layout, profile=profile)
            for which in range(2):
                src, dst = operands((W * H, W * H), which, into)
                src.copy_(d_img)
                dst.fill_(CANARY)
                M.fwd_quant_u8(src, dst, lut, W, H, 0, rows, layout=layout, profile=profile)
                assert torch.equal(dst, ref), (beh, which, hex(into))
        # pixels -> records (fused)
        for which in range(4):
            src, lvb, rn, ctp = operands((W * H, nblk * 128, nblk * 64, ctb), which, into)
            src.copy_(d_img)
            lv, rn, ct = lvb.view(torch.int16).reshape(nblk, 64), rn.reshape(nblk, 64), ctp[:nblk]
            M.fwd_u8_records(src, W, H, lv, rn, ct, lut=lut100)
            assert np.array_equal(lv.cpu().numpy(), wl) and np.array_equal(rn.cpu().numpy(), wr) and np.array_equal(ct.cpu().numpy(), wc), (which, hex(into))
        # records -> Huffman rows
        for which in range(5):
            lvb, rnb, ctp, seg, nbb = operands((nblk * 128, nblk * 64, ctb, (H // 8) * stride, 1024), which, into)
            lv, rn, ct, nb = lvb.view(torch.int16).reshape(nblk, 64), rnb.reshape(nblk, 64), ctp[:nblk], nbb.view(torch.int32)[: H // 8]
            lv.copy_(d_lv); rn.copy_(d_rn); ct.copy_(d_ct)
            M.huffman_rows(lv, rn, ct, W, H, seg, nb)
            gs = seg.cpu().numpy()
            assert np.array_equal(nb.cpu().numpy().astype(np.uint32), wn), (which, hex(into))
            for r in range(H // 8):
                assert np.array_equal(gs[r * stride:r * stride + wn[r]], ws[r * stride:r * stride + wn[r]]), (which, r, hex(into))
        # row segments -> packed scan
        for which in range(4):
            seg, nbb, scan, offb = operands(((H // 8) * stride, 1024, W * H, 4096), which, into)
            nb, off = nbb.view(torch.int32)[: H // 8], offb.view(torch.int64)[: H // 8 + 1]
            seg.copy_(d_seg); nb.copy_(d_nb)
            M.jpeg_pack_rows(seg, nb, stride, H // 8, scan, off)
            total = int(woff[-1])
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), woff) and np.array_equal(scan[:total].cpu().numpy(), wo[:total]), (which, hex(into))
    del arena


def test_more_than_65535_block_rows_take_the_linear_kernels():
    """the 2-D tile launches (k_q32_tile, k_fwd_quant_u8<TILED>, k_i16_tile, k_f32_tile) put the block row in blockIdx.y, which
    ends at 65535: taller planes must fall back to the linear kernels and still be right.  Periodic input as above: every
    period equals the first on the device, the first equals the oracle."""
    n_per = 4112  # x 16 block rows = 65792 block rows
    H = P * n_per
    # q32 (512 wide: one tile per row) and the 256-thread layouts (2048 wide)
    for W, beh, layout, profile, scale in ((512, "q32_avx", M.LAYOUT_Q32, M.PROFILE_REF_AVX, 2000.0), (2048, "encq_scalar", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, 8.0)):
        assert H // 8 > 65535
        strip = synth.plane_u8_np(W, P, "photo", seed=5)
        src = torch.from_numpy(strip).cuda().repeat(n_per, 1).contiguous()
        out = torch.full((W * H,), CANARY, dtype=torch.uint8, device="cuda")
        M.fwd_quant_u8(src, out, lut_x(scale), W, H, 0, H // 8, layout=layout, profile=profile)
        torch.cuda.synchronize()
        assert all_periods_equal_first(out, n_per), beh
        if beh == "q32_avx":
            rc, want = O.q32_native(strip, lut_x(scale), W, P, 0, P // 8)
        else:
            rc, want = O.run_behaviour(beh, np.ascontiguousarray(np.tile(strip, (2, 1))), lut_x(scale), W, 2 * P, 0, 2 * P)
            want = want[: W * P]
        assert np.array_equal(out[: W * P].cpu().numpy(), want), beh
        del src, out
    W = 512
    s16 = synth.plane_i16_np(W, P, "photo", seed=6)
    src = torch.from_numpy(s16).cuda().repeat(n_per, 1).contiguous()
    dst = torch.empty_like(src)
    for mode, fn in (("fwd", M.fwd_i16), ("roundtrip", M.roundtrip_i16)):
        dst.fill_(-1)
        fn(src, dst, W, H)
        torch.cuda.synchronize()
        assert all_periods_equal_first(dst, n_per), mode
        assert np.array_equal(dst[:P].cpu().numpy(), O.i16(mode, s16, W, P)), mode
    f = src.to(torch.float32)
    g = torch.empty_like(f)
    M.fwd_f32(f, g, W, H)
    torch.cuda.synchronize()
    assert all_periods_equal_first(g, n_per)
    assert np.array_equal(g[:P].cpu().numpy(), O.f32("fwd", s16.astype(np.float32), W, P))


def test_plane_batch_beyond_4GiB_and_more_than_65535_block_rows():
    """the plane-batch kernel (a 1-D grid of tiles: no 65535-row limit of its own) on planes whose byte offsets pass 2^32: one 8192 x 335872
    int16 plane (5.5 GB, 41984 block rows) and, in the same call, a 2056-wide one with a partial last tile and 70000 block rows -- every
    period equals the first, the first equals the oracle; both forms of the call; forward and the fused round trip with a table"""
    lut = lut_x(40)
    shapes = [(8192, P * 2624), (2056, 560000)]
    assert shapes[0][0] * shapes[0][1] * 2 > 2**32 + 2**30 and shapes[1][1] // 8 > 65535 and shapes[1][1] % P == 0
    strips = [synth.plane_i16_np(w, P, "photo", seed=31 + k, bits=12) for k, (w, h) in enumerate(shapes)]
    srcs = [torch.from_numpy(s).cuda().repeat(h // P, 1).contiguous() for s, (w, h) in zip(strips, shapes)]
    outs = [torch.empty_like(t) for t in srcs]
    for mode, table in (("fwd", None), ("roundtrip", lut)):
        wants = [O.i16(mode, s, w, P, lut=table) for s, (w, h) in zip(strips, shapes)]
        for form in ("args", "device"):
            for o in outs:
                o.fill_(-21846)
            desc = [(a, o, w, h, table) for a, o, (w, h) in zip(srcs, outs, shapes)]
            if form == "args":
                M.i16_batch(mode, desc)
            else:
                b = M.Batch(mode, desc)
                assert b.launches == 1
                b.run()
            torch.cuda.synchronize()
            for o, want, (w, h) in zip(outs, wants, shapes):
                assert all_periods_equal_first(o, h // P), (mode, form, w)
                assert np.array_equal(o[:P].cpu().numpy(), want), (mode, form, w)
    del srcs, outs
