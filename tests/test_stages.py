"""The stages either side of the transform (include/mdct.h): zig-zag scan + run/level
records after the quantiser, 4:2:0 split / subsample before config 3's transform.  CPU: the checker
against the published table of ITU-T T.81 Figure A.6, against a numpy restatement, and the product's
own table; GPU: the kernels bit-exact against the checker, plus reconstruction at full size."""
import numpy as np
import pytest

import oracle as O
from simd_dct_amd import api, synth

# ITU-T T.81 (09/92) Figure A.6 "Zig-zag sequence of quantized DCT coefficients": the number printed in cell
# (row v, column u) is the position of that coefficient in the scan.
T81_FIGURE_A6 = np.array([
    [0, 1, 5, 6, 14, 15, 27, 28],
    [2, 4, 7, 13, 16, 26, 29, 42],
    [3, 8, 12, 17, 25, 30, 41, 43],
    [9, 11, 18, 24, 31, 40, 44, 53],
    [10, 19, 23, 32, 39, 45, 52, 54],
    [20, 22, 33, 38, 46, 51, 55, 60],
    [21, 34, 37, 47, 50, 56, 59, 61],
    [35, 36, 48, 49, 57, 58, 62, 63]])


def test_zigzag_tables_are_t81_figure_a6():
    want = np.argsort(T81_FIGURE_A6.reshape(-1))  # scan position k -> natural index
    assert np.array_equal(O.zigzag_table(), want)      # the checker walks the anti-diagonals
    assert np.array_equal(api.zigzag_table(), want)    # the product carries literals


def _numpy_records(nat):
    """nat: [nblk, 64] natural-order coefficients -> (levels, runs, counts) by definition"""
    zz = np.argsort(T81_FIGURE_A6.reshape(-1))
    scan = nat[:, zz]
    levels = np.zeros_like(scan, dtype=np.int16)
    runs = np.zeros(scan.shape, dtype=np.uint8)
    counts = np.zeros(scan.shape[0], dtype=np.uint8)
    for b in range(scan.shape[0]):
        nzpos = np.flatnonzero(scan[b])
        counts[b] = len(nzpos)
        levels[b, :len(nzpos)] = scan[b, nzpos]
        runs[b, :len(nzpos)] = np.diff(np.concatenate(([-1], nzpos))) - 1
    return scan.astype(np.int16), levels, runs, counts


def _blocks_i16(plane, W, H):
    return plane.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)


def test_oracle_scan_and_rle_by_definition():
    W, H = 128, 48
    rng = np.random.default_rng(4)
    coef = (rng.integers(-300, 300, (H, W)) * (rng.random((H, W)) < 0.15)).astype(np.int16)
    coef[:8, :8] = 0            # an all-zero block: count 0
    coef[8:16, :8] = 7          # a full block: 64 pairs, all runs 0
    scan, lv, rn, ct = _numpy_records(_blocks_i16(coef, W, H))
    got_lv, got_rn, got_ct = O.zigzag_rle("i16", coef, W, H)
    assert np.array_equal(got_lv, lv) and np.array_equal(got_rn, rn) and np.array_equal(got_ct, ct)
    assert ct[0] == 0 and ct[W // 8] == 64
    plain, _, _ = O.zigzag_rle("i16", coef, W, H, rle=False)
    assert np.array_equal(plain, scan)
    # the q32 byte layout of the same coefficients (bias +127, 8 blocks interleaved: simd_dct.cpp:2221-2230)
    small = np.clip(coef, -127, 128)
    nat = _blocks_i16(small, W, H)
    q32 = (nat + 127).astype(np.uint8).reshape(H // 8, W // 64, 8, 64).transpose(0, 1, 3, 2).reshape(-1)
    scan, lv, rn, ct = _numpy_records(nat)
    got_lv, got_rn, got_ct = O.zigzag_rle("q32", q32, W, H)
    assert np.array_equal(got_lv, lv) and np.array_equal(got_rn, rn) and np.array_equal(got_ct, ct)


def test_oracle_scan_of_the_stereo_and_block_layouts():
    """the checker's stereo / block-layout scans against the layouts' definitions (numpy), fed with the pinned
    oracle's own bytes for those tiers"""
    W, H = 128, 64
    img = synth.plane_u8_np(W, H, "photo")
    lut = (api.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
    # stereo: 64 planes, stream position p; natural index i at plane i
    rc, st = O.run_behaviour("stereo_sse", img, lut, W, H, 0, H)
    nat = st.reshape(64, -1).T.astype(np.int32) - 127  # [position][coef]
    scan, lv, rn, ct = _numpy_records(nat)
    got = O.zigzag_rle("stereo", st, W, H)
    assert np.array_equal(got[0], lv) and np.array_equal(got[1], rn) and np.array_equal(got[2], ct)
    # scalar encq: 64 bytes per block, stored transposed; only the top half of the plane is written
    rc, bl = O.run_behaviour("encq_scalar", img, lut, W, H, 0, H)
    nat = bl.reshape(-1, 8, 8).transpose(0, 2, 1).reshape(-1, 64).astype(np.int32) - 127
    scan, lv, rn, ct = _numpy_records(nat)
    got = O.zigzag_rle("block", bl, W, H)
    assert np.array_equal(got[0], lv) and np.array_equal(got[1], rn) and np.array_equal(got[2], ct)


def _ycc(W, H, seed=1):
    y = synth.plane_u8_np(W, H, "photo", seed=seed)
    cb = synth.plane_u8_np(W, H, "noise", seed=seed + 1)
    cr = synth.plane_u8_np(W, H, "photo", seed=seed + 2)[::-1]
    return np.ascontiguousarray(np.stack([y, cb, cr], axis=-1))


def test_oracle_split420_by_definition():
    W, H = 64, 32
    ycc = _ycc(W, H)
    y, cb, cr = O.split420(ycc, W, H)
    assert np.array_equal(y, ycc[:, :, 0].astype(np.int16) - 128)
    for k, got in ((1, cb), (2, cr)):
        c = ycc[:, :, k].astype(np.int32)
        box = c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2]
        assert np.array_equal(got, ((box + 2) >> 2) - 128)


def test_stage_argument_checks_without_device():
    a = np.zeros((16, 64), dtype=np.int16)
    lv = np.zeros((16, 64), dtype=np.int16)
    rn = np.zeros((16, 64), dtype=np.uint8)
    ct = np.zeros(16, dtype=np.uint8)
    assert api.zigzag_rle_i16(None, 64, 16, lv, check=False) == 1
    assert api.zigzag_rle_i16(a, 60, 16, lv, check=False) == 2
    assert api.zigzag_rle_i16(a, 64, 16, lv, rn, None, check=False) == 1      # runs without counts
    assert api.zigzag_rle_i16(a, 64, 16, lv, rn, ct, by1=3, check=False) == 1  # range beyond the plane
    assert api.zigzag_rle_q32(a.view(np.uint8), 56, 16, lv, check=False) == 2  # q32 needs sizeX % 64
    px = np.zeros((16, 64), dtype=np.uint8)
    assert api.fwd_u8_records(px, 64, 16, lv, rn, None, check=False) == 1
    assert api.fwd_u8_records(px, 60, 16, lv, rn, ct, check=False) == 2
    assert api.fwd_u8_records(px, 64, 16, lv, rn, ct, pitch=32, check=False) == 1
    assert api.fwd_u8_records(px, 64, 16, lv, rn, ct, by1=3, check=False) == 1
    ycc = np.zeros((16, 16, 3), dtype=np.uint8)
    y = np.zeros((16, 16), dtype=np.int16)
    c = np.zeros((8, 8), dtype=np.int16)
    assert api.split420_u8(ycc, 24, 16, y, c, c, check=False) == 2
    assert api.split420_u8(ycc, 16, 16, y, c, c, pitch=40, check=False) == 1
    y8, c8 = np.zeros((16, 16), dtype=np.uint8), np.zeros((8, 8), dtype=np.uint8)
    assert api.split420_u8_planes(ycc, 24, 16, y8, c8, c8, check=False) == 2
    assert api.split420_u8_planes(ycc, 16, 16, y8, c8, c8, pitch_c=4, check=False) == 1
    assert api.split420_u8_planes(ycc, 16, 16, y8, None, c8, check=False) == 1


# ------------------------------------------------------------------------------------------ GPU
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_scan_and_rle_match_the_checker():
    api.init(0)
    rng = np.random.default_rng(12)
    lut = (api.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
    for (W, H) in ((64, 8), (128, 48), (320, 40), (1024, 256)):
        nblk = (W // 8) * (H // 8)
        # int16 coefficients: real quantised DCT output (sparse) and random dense / saturating values
        src = synth.plane_i16_np(W, H, "photo", seed=W)
        coefs = [O.i16("fwd", src, W, H, lut=(api.QUANTIZE_BASE * np.float32(60)).astype(np.float32)),
                 rng.integers(-32768, 32768, (H, W), dtype=np.int16),
                 np.zeros((H, W), dtype=np.int16)]
        for coef in coefs:
            for rle in (True, False):
                for (b0, b1) in ((0, H // 8), (H // 16, H // 8)):
                    lv = torch.full((nblk, 64), 0x1111, dtype=torch.int16, device="cuda")
                    rn = torch.full((nblk, 64), 0x11, dtype=torch.uint8, device="cuda") if rle else None
                    ct = torch.full((nblk,), 0x11, dtype=torch.uint8, device="cuda") if rle else None
                    api.zigzag_rle_i16(_dev(coef), W, H, lv, rn, ct, by0=b0, by1=b1)
                    want = O.zigzag_rle("i16", coef, W, H, rle=rle, by0=b0, by1=b1, fill=0x1111)
                    assert np.array_equal(lv.cpu().numpy(), want[0]), (W, H, rle, b0)
                    if rle:
                        assert np.array_equal(rn.cpu().numpy(), want[1]) and np.array_equal(ct.cpu().numpy(), want[2]), (W, H, b0)
        if W % 64 == 0:  # the reference's own product as the source: GPU q32 bytes -> records
            img = synth.plane_u8_np(W, H, "photo", seed=W + 1)
            q = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
            api.fwd_quant_u8(_dev(img), q, lut, W, H, 0, H // 8)
            rc, qh = O.q32_native(img, lut, W, H, 0, H // 8)
            for rle in (True, False):
                lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
                rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda") if rle else None
                ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda") if rle else None
                api.zigzag_rle_q32(q, W, H, lv, rn, ct)
                want = O.zigzag_rle("q32", qh, W, H, rle=rle)
                assert np.array_equal(lv.cpu().numpy(), want[0]), (W, H, rle)
                if rle:
                    assert np.array_equal(rn.cpu().numpy(), want[1]) and np.array_equal(ct.cpu().numpy(), want[2])


@pytest.mark.gpu
def test_fused_pixels_to_records_equals_the_two_stage_path():
    """mdct_fwd_u8_records == mdct_fwd_u8_i16 + mdct_zigzag_rle_i16 == the checker's composition, bit for bit: one lane,
    partial waves, pitched and unaligned pixel planes, no table / no level shift, sub-ranges leaving the rest alone, 8192^2"""
    api.init(0)
    K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
    for (W, H, pitch) in ((8, 8, 8), (64, 24, 64), (200, 40, 203), (1000, 16, 1024), (2048, 64, 2048)):
        nblk = (W // 8) * (H // 8)
        for kind, lut, shift in (("photo", K1, True), ("noise", None, True), ("photo", (K1 / 8).astype(np.float32), False)):
            wide = synth.plane_u8_np(pitch, H + 1, kind, seed=W + H)
            img = np.ascontiguousarray(wide[:H, :W])
            d_wide = _dev(wide.reshape(-1)[1:])  # an odd base address: no alignment requirement on the pixel plane
            want = O.u8_records(wide.reshape(-1)[1:1 + pitch * H].reshape(H, pitch)[:, :W], W, H, lut=lut, level_shift=shift)
            lv = torch.full((nblk, 64), 0x5A5A, dtype=torch.int16, device="cuda")
            rn = torch.full((nblk, 64), 0x5A, dtype=torch.uint8, device="cuda")
            ct = torch.full((nblk,), 0x5A, dtype=torch.uint8, device="cuda")
            api.fwd_u8_records(d_wide, W, H, lv, rn, ct, lut=lut, level_shift=shift, pitch=pitch)
            for got, w in zip((lv, rn, ct), want):
                assert np.array_equal(got.cpu().numpy(), w), (W, H, kind)
            # the two-stage path on the device gives the same records
            src = _dev(wide.reshape(-1)[1:1 + pitch * H].reshape(H, pitch)[:, :W])
            coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
            lv2, rn2, ct2 = torch.empty_like(lv), torch.empty_like(rn), torch.empty_like(ct)
            api.fwd_u8_i16(src, coef, W, H, lut=lut, level_shift=shift)
            api.zigzag_rle_i16(coef, W, H, lv2, rn2, ct2)
            assert torch.equal(lv, lv2) and torch.equal(rn, rn2) and torch.equal(ct, ct2)
            del img
    # the largest values 8-bit pixels can produce (flat 255 / 0, the sign pattern of every basis function) through the smallest table for
    # which the kernel leaves its int16 saturations out (every entry 1/16: |coefficient| * 16 <= 32640), and through one just below it
    xs = np.arange(8)
    cosm = np.cos((2 * xs[None, :] + 1) * xs[:, None] * np.pi / 16)
    worst = np.zeros((24, 512), dtype=np.uint8)
    for u in range(8):
        for v in range(8):
            blk = np.where(np.outer(cosm[v], cosm[u]) > 0, 255, 0).astype(np.uint8)
            worst[0:8, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = blk
            worst[8:16, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = 255 - blk
    worst[16:24, :256] = 255
    for q in (np.full(64, 0.0625, dtype=np.float32), np.full(64, 0.0624, dtype=np.float32), np.full(64, -0.0625, dtype=np.float32)):
        for shift in (False, True):
            want = O.u8_records(worst, 512, 24, lut=q, level_shift=shift)
            assert int(np.abs(want[0].astype(np.int64)).max()) >= 16000
            lv = torch.full((192, 64), 0x5A5A, dtype=torch.int16, device="cuda")
            rn = torch.full((192, 64), 0x5A, dtype=torch.uint8, device="cuda")
            ct = torch.full((192,), 0x5A, dtype=torch.uint8, device="cuda")
            api.fwd_u8_records(_dev(worst), 512, 24, lv, rn, ct, lut=q, level_shift=shift)
            for got, w in zip((lv, rn, ct), want):
                assert np.array_equal(got.cpu().numpy(), w), (float(q[0]), shift)
    assert api.fwd_u8_records(_dev(np.zeros((16, 64), dtype=np.uint8)), 64, 16, lv, rn, ct, lut=np.zeros(64, dtype=np.float32), check=False) == 1  # a zero table entry
    # the int16-plane variant: mdct_fwd_i16_records == mdct_fwd_i16 + mdct_zigzag_rle_i16 == the checker's composition
    for (W, H, pitch) in ((8, 8, 8), (200, 40, 208), (1024, 72, 1024)):
        nblk = (W // 8) * (H // 8)
        for bits, lut in ((8, K1), (12, None), (12, (K1 * 4).astype(np.float32))):
            wide = synth.plane_i16_np(pitch, H, "photo", seed=W, bits=bits)
            src = np.ascontiguousarray(wide[:, :W])
            want = O.zigzag_rle("i16", O.i16("fwd", src, W, H, lut=lut), W, H)
            lv = torch.full((nblk, 64), 0x5A5A, dtype=torch.int16, device="cuda")
            rn = torch.full((nblk, 64), 0x5A, dtype=torch.uint8, device="cuda")
            ct = torch.full((nblk,), 0x5A, dtype=torch.uint8, device="cuda")
            api.fwd_i16_records(_dev(wide), W, H, lv, rn, ct, lut=lut, pitch=pitch)
            for got, w in zip((lv, rn, ct), want):
                assert np.array_equal(got.cpu().numpy(), w), (W, H, bits)
    assert api.fwd_i16_records(_dev(np.zeros((16, 72), dtype=np.int16)), 64, 16, lv, rn, ct, pitch=68, check=False) == 1  # rows not 16-byte aligned
    # a sub-range writes its own records only
    W, H = 512, 64
    bpr = W // 8
    img = synth.plane_u8_np(W, H, "photo", seed=5)
    want = O.u8_records(img, W, H, lut=K1)
    lv = torch.full(((W // 8) * (H // 8), 64), 0x5A5A, dtype=torch.int16, device="cuda")
    rn = torch.full(((W // 8) * (H // 8), 64), 0x5A, dtype=torch.uint8, device="cuda")
    ct = torch.full(((W // 8) * (H // 8),), 0x5A, dtype=torch.uint8, device="cuda")
    api.fwd_u8_records(_dev(img), W, H, lv, rn, ct, lut=K1, by0=2, by1=5)
    g = [t.cpu().numpy() for t in (lv, rn, ct)]
    for got, w, canary in zip(g, want, (0x5A5A, 0x5A, 0x5A)):
        assert np.array_equal(got[2 * bpr:5 * bpr], w[2 * bpr:5 * bpr])
        assert (got[:2 * bpr] == canary).all() and (got[5 * bpr:] == canary).all()
    # full size: against the two-stage device path (itself checked against the oracle at this size elsewhere)
    W = H = 8192
    img = synth.plane_u8_torch(W, H, "photo")
    nblk = (W // 8) * (H // 8)
    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
    rec = [(torch.empty((nblk, 64), dtype=torch.int16, device="cuda"), torch.empty((nblk, 64), dtype=torch.uint8, device="cuda"), torch.empty((nblk,), dtype=torch.uint8, device="cuda")) for _ in range(2)]
    api.fwd_u8_records(img, W, H, *rec[0], lut=K1)
    api.fwd_u8_i16(img, coef, W, H, lut=K1)
    api.zigzag_rle_i16(coef, W, H, *rec[1])
    for a, b in zip(*rec):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_scan_of_the_stereo_and_block_layouts():
    """the reordered streams the reference's other two functions produce, consumed in place: GPU stereo / scalar-encq
    output -> records, against the checker; sub-ranges leave the other records alone"""
    api.init(0)
    lut = (api.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
    for (W, H) in ((16, 16), (128, 64), (1040, 48), (2048, 512)):
        img = synth.plane_u8_np(W, H, "photo", seed=H)
        nblk = (W // 8) * (H // 8)
        for kind, layout, profile, beh, rows in (("stereo", api.LAYOUT_STEREO, api.PROFILE_REF_SSE, "stereo_sse", H // 16), ("block", api.LAYOUT_BLOCK, api.PROFILE_REF_SCALAR, "encq_scalar", H // 8)):
            out = torch.zeros(W * H, dtype=torch.uint8, device="cuda")
            api.fwd_quant_u8(_dev(img), out, lut, W, H, 0, rows, layout=layout, profile=profile)
            host = out.cpu().numpy()
            for rle in (True, False):
                for (b0, b1) in ((0, rows), (rows // 2, rows)):
                    lv = torch.full((nblk, 64), 0x1111, dtype=torch.int16, device="cuda")
                    rn = torch.full((nblk, 64), 0x11, dtype=torch.uint8, device="cuda") if rle else None
                    ct = torch.full((nblk,), 0x11, dtype=torch.uint8, device="cuda") if rle else None
                    api.zigzag_rle_u8(out, layout, W, H, lv, rn, ct, by0=b0, by1=b1)
                    want = O.zigzag_rle(kind, host, W, H, rle=rle, by0=b0, by1=b1, fill=0x1111)
                    assert np.array_equal(lv.cpu().numpy(), want[0]), (kind, W, H, rle, b0)
                    if rle:
                        assert np.array_equal(rn.cpu().numpy(), want[1]) and np.array_equal(ct.cpu().numpy(), want[2]), (kind, W, H, b0)
    a = torch.zeros(64 * 16, dtype=torch.uint8, device="cuda")
    lv = torch.zeros((16, 64), dtype=torch.int16, device="cuda")
    assert api.zigzag_rle_u8(a, api.LAYOUT_BLOCK_SSE, 64, 16, lv, check=False) == 2


@pytest.mark.gpu
def test_records_reconstruct_the_plane_at_full_size():
    """8192x8192: JPEG-luma-quantised coefficients -> records; expanding the records (torch, on the device)
    gives back every coefficient; sampled block rows equal the checker"""
    api.init(0)
    W = H = 8192
    luma = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
    img = synth.plane_u8_torch(W, H, "photo")
    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
    api.fwd_u8_i16(img, coef, W, H, lut=luma)
    nblk = (W // 8) * (H // 8)
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    api.zigzag_rle_i16(coef, W, H, lv, rn, ct)
    # expand: scan position of pair i = cumsum(run + 1) - 1
    pos = torch.cumsum(rn.to(torch.int64) + 1, dim=1) - 1
    live = torch.arange(64, device="cuda")[None, :] < ct[:, None].to(torch.int64)
    scan = torch.zeros((nblk, 65), dtype=torch.int16, device="cuda")
    scan.scatter_(1, torch.where(live, pos, torch.full_like(pos, 64)), lv)
    zz = torch.from_numpy(api.zigzag_table().astype(np.int64)).cuda()
    nat = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
    nat[:, zz] = scan[:, :64]
    back = nat.reshape(H // 8, W // 8, 8, 8).permute(0, 2, 1, 3).reshape(H, W)
    assert torch.equal(back, coef)
    assert int(ct.max().item()) <= 64 and float(ct.float().mean().item()) < 32  # quantised photo content is sparse
    host = coef[:64].cpu().numpy()
    want = O.zigzag_rle("i16", host, W, 64)
    n = (W // 8) * 8
    assert np.array_equal(lv[:n].cpu().numpy(), want[0]) and np.array_equal(rn[:n].cpu().numpy(), want[1]) and np.array_equal(ct[:n].cpu().numpy(), want[2])


@pytest.mark.gpu
def test_split420_matches_the_checker_and_feeds_config3():
    api.init(0)
    for (W, H) in ((16, 16), (64, 32), (1008, 48), (1920, 1088)):
        ycc = _ycc(W, H, seed=W)
        y = torch.full((H, W), 77, dtype=torch.int16, device="cuda")
        cb = torch.full((H // 2, W // 2), 77, dtype=torch.int16, device="cuda")
        cr = torch.full((H // 2, W // 2), 77, dtype=torch.int16, device="cuda")
        api.split420_u8(_dev(ycc), W, H, y, cb, cr)
        wy, wcb, wcr = O.split420(ycc, W, H)
        assert np.array_equal(y.cpu().numpy(), wy) and np.array_equal(cb.cpu().numpy(), wcb) and np.array_equal(cr.cpu().numpy(), wcr), (W, H)
    # BASELINE.json configs[2] end to end: 7680x4320 interleaved frame -> split -> the three planes through
    # mdct_roundtrip_i16_planes with per-plane tables in one call; the luma plane equals the single-plane entry point
    W, H = 7680, 4320 + 16 - (4320 % 16 or 16)  # heights are multiples of 16 here (4320 is)
    ycc = torch.stack([synth.plane_u8_torch(W, H, "photo", seed=s) for s in (1, 2, 3)], dim=-1).contiguous()
    y = torch.empty((H, W), dtype=torch.int16, device="cuda")
    cb = torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda")
    cr = torch.empty_like(cb)
    api.split420_u8(ycc, W, H, y, cb, cr)
    assert torch.equal(y, ycc[:, :, 0].to(torch.int16) - 128)
    c = ycc[:, :, 1].to(torch.int32)
    assert torch.equal(cb.to(torch.int32), ((c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2) - 128)
    lut = (api.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
    outs = [torch.empty_like(t) for t in (y, cb, cr)]
    api.roundtrip_i16_planes([(y, outs[0], W, H, lut), (cb, outs[1], W // 2, H // 2, lut), (cr, outs[2], W // 2, H // 2, lut)])
    single = torch.empty_like(y)
    api.roundtrip_i16(y, single, W, H, lut=lut)
    assert torch.equal(single, outs[0])


@pytest.mark.gpu
def test_split420_into_8bit_planes_feeds_the_8bit_config3_path():
    """mdct_split420_u8_planes: the 4:2:0 split into unshifted 8-bit planes == the checker's int16 split + 128, any pitches / alignment with
    canaries; BASELINE.json configs[2] as SURVEY.md 8(d) states it end to end: interleaved 7680x4320 frame -> split -> the three 8-bit planes
    through mdct_roundtrip_u8_batch in one launch against the int16 route's reconstruction (split420_u8 -> roundtrip_i16_planes) + 128, clamped"""
    api.init(0)
    for (W, H, py, pc) in ((16, 16, 16, 8), (64, 32, 67, 35), (1008, 48, 1011, 505), (1920, 1088, 1920, 960)):
        ycc = _ycc(W, H, seed=W + 1)
        y = torch.full((H, py), 77, dtype=torch.uint8, device="cuda")
        cb = torch.full((H // 2, pc), 77, dtype=torch.uint8, device="cuda")
        cr = torch.full((H // 2, pc), 77, dtype=torch.uint8, device="cuda")
        api.split420_u8_planes(_dev(ycc), W, H, y, cb, cr, pitch_y=py, pitch_c=pc)
        wy, wcb, wcr = O.split420(ycc, W, H)
        for got, want, w in ((y, wy, W), (cb, wcb, W // 2), (cr, wcr, W // 2)):
            g = got.cpu().numpy()
            assert np.array_equal(g[:, :w].astype(np.int16) - 128, want) and (g[:, w:] == 77).all(), (W, H)
    W, H = 7680, 4320
    ycc = torch.stack([synth.plane_u8_torch(W, H, "photo", seed=s) for s in (1, 2, 3)], dim=-1).contiguous()
    p8 = [torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda"), torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda")]
    api.split420_u8_planes(ycc, W, H, *p8)
    assert torch.equal(p8[0], ycc[:, :, 0])
    p16 = [torch.empty(t.shape, dtype=torch.int16, device="cuda") for t in p8]
    api.split420_u8(ycc, W, H, *p16)
    for a, b in zip(p8, p16):
        assert torch.equal(a.to(torch.int16) - 128, b)
    luts = [synth.JPEG_LUMA, synth.JPEG_CHROMA, synth.JPEG_CHROMA]
    dims = [(W, H), (W // 2, H // 2), (W // 2, H // 2)]
    o8 = [torch.empty_like(t) for t in p8]
    o16 = [torch.empty_like(t) for t in p16]
    api.roundtrip_u8_batch([(a, o, w, h, l) for a, o, (w, h), l in zip(p8, o8, dims, luts)])
    api.roundtrip_i16_planes([(a, o, w, h, l) for a, o, (w, h), l in zip(p16, o16, dims, luts)])
    # the same transform and quantiser on the same (shifted) samples.  Since round 6 the 8-bit inverse carries the output's + 128 in its DC term
    # (sat_u8(rne(idct(z, DC + 128))), one convert per pixel) where the int16 route adds it to the rounded sample: the two agree except where a
    # sample sits within the butterflies' rounding error of a .5 tie -- never by more than one grey level, and rarely
    for a, b in zip(o8, o16):
        d = (a.to(torch.int16) - (b + 128).clamp(0, 255)).abs()
        assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-3, (int(d.max()), float((d != 0).float().mean()))
