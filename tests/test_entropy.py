"""The entropy stage ("zig-zag + run-length / entropy stage consuming the reordered streams"):
baseline Huffman coding of the run/level records, and the JFIF container around it.  No reference counterpart --
but an INDEPENDENT one: libjpeg (through PIL) must decode the files, reproduce our own inverse transform within the
+-1 of two different IDCTs, and write the same Huffman tables itself.  That pins scan order, run/level semantics,
code construction, DC prediction / restart semantics and the tables.  GPU: the kernel's bytes equal the checker's."""
import io
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from simd_dct_amd import api, jfif, synth

Image = pytest.importorskip("PIL.Image")

# ITU-T T.81 Annex K.1 / K.2 example quantisation tables (natural order)
K1_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
K2_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.float32)


def _libjpeg_tables(colour):
    """the DHT specifications libjpeg itself writes (optimize=False = its copy of Annex K.3.3)"""
    img = Image.fromarray(np.zeros((16, 16, 3) if colour else (16, 16), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, "JPEG", quality=50, optimize=False)
    b = buf.getvalue()
    i, got = 2, {}
    while i < len(b):
        m, L = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xC4:
            p = i + 4
            while p < i + 2 + L:
                bits = list(b[p + 1:p + 17])
                n = sum(bits)
                got[b[p]] = (bits, list(b[p + 17:p + 17 + n]))
                p += 17 + n
        if m == 0xDA:
            break
        i += 2 + L
    return got


def test_huffman_tables_are_the_ones_libjpeg_writes():
    got = _libjpeg_tables(colour=True)
    for which, key in ((0, 0x00), (1, 0x10), (2, 0x01), (3, 0x11)):
        assert O.huffman_spec(which) == got[key], which  # the checker's literals
        assert api.huffman_spec(which) == got[key], which  # the product's literals


def _encode_cpu(img, qtable, chroma=False):
    H, W = img.shape
    coef = O.u8_i16("fwd", img, W, H, lut=qtable)
    lv, rn, ct = O.zigzag_rle("i16", coef, W, H)
    seg, nb, stride = O.huffman_rows(lv, rn, ct, W, H, chroma=chroma)
    return coef, dict(segments=seg, seg_bytes=nb, seg_stride=stride, blocks_per_row=W // 8, qtable=qtable)


def test_libjpeg_decodes_the_checkers_stream():
    for (W, H), kind in (((256, 128), "photo"), ((64, 8), "noise"), ((1024, 64), "photo")):
        img = synth.plane_u8_np(W, H, kind)
        coef, comp = _encode_cpu(img, K1_LUMA)
        data = jfif.write_jpeg([comp], W, H)
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("L"))
        own = O.u8_i16("inv", coef, W, H, lut=K1_LUMA)
        assert dec.shape == (H, W)
        assert np.abs(dec.astype(int) - own.astype(int)).max() <= 1, (W, H)  # libjpeg's integer IDCT vs our float one
    # extremes: flat blocks (EOB only), full-scale checkerboards (long codes, 0xFF bytes -> stuffing), runs > 15 (ZRL)
    W, H = 128, 32
    img = np.zeros((H, W), dtype=np.uint8)
    img[:, 32:64] = 255
    img[8:16, 64:96] = (np.indices((8, 32)).sum(0) % 2) * 255
    img[16:24, :] = synth.plane_u8_np(W, 8, "noise")
    k = np.cos((2 * np.arange(8) + 1) * 7 * np.pi / 16)
    img[24:32, 0:8] = np.rint(128 + 100 * np.outer(k, k)).astype(np.uint8)  # mostly the (7,7) coefficient: scan position 63
    k4 = np.cos((2 * np.arange(8) + 1) * 4 * np.pi / 16)  # +-sqrt(1/2): the outer product is +-1/2, the pixels are exact integers, so
    img[24:32, 8:16] = np.rint(128 + 100 * np.outer(k4, k4)).astype(np.uint8)  # ONLY DC and (4,4) survive whatever the rounding: a run of 40 zeros
    q1 = np.ones(64, dtype=np.float32)
    coef, comp = _encode_cpu(img, q1)
    assert (comp["seg_bytes"] > 0).all()
    lv, rn, ct = O.zigzag_rle("i16", coef, W, H)
    assert rn.max() > 15  # the ZRL path is exercised
    data = jfif.write_jpeg([comp], W, H)
    assert b"\xff\x00" in data  # and so is byte stuffing
    dec = np.asarray(Image.open(io.BytesIO(data)).convert("L"))
    assert np.abs(dec.astype(int) - O.u8_i16("inv", coef, W, H, lut=q1).astype(int)).max() <= 1


def test_huffman_argument_checks_without_device():
    lv = np.zeros((16, 64), dtype=np.int16)
    rn = np.zeros((16, 64), dtype=np.uint8)
    ct = np.zeros(16, dtype=np.uint8)
    out = np.zeros(2 * api.huffman_seg_stride(64), dtype=np.uint8)
    nb = np.zeros(2, dtype=np.uint32)
    assert api.huffman_rows(None, rn, ct, 64, 16, out, nb, check=False) == 1
    assert api.huffman_rows(lv, rn, ct, 60, 16, out, nb, check=False) == 2
    assert api.huffman_rows(lv, rn, ct, 64, 16, out, nb, seg_stride=64, check=False) == 1
    assert "208" in api.last_error()
    assert api.huffman_rows(lv, rn, ct, 64, 16, out, nb, by1=3, check=False) == 1


# ------------------------------------------------------------------------------------------ GPU
torch = pytest.importorskip("torch")


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _encode_gpu(d_img, W, H, qtable, chroma=False, from_i16=None, fused=False):
    """u8 plane (or ready int16 plane) on the device -> coefficients, records, row segments; all on the device.
    fused: the records come from mdct_fwd_u8_records / mdct_fwd_i16_records instead of the two-stage path"""
    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
    if from_i16 is None:
        api.fwd_u8_i16(d_img, coef, W, H, lut=qtable)
    else:
        api.fwd_i16(from_i16, coef, W, H, lut=qtable)
    nblk = (W // 8) * (H // 8)
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    if not fused:
        api.zigzag_rle_i16(coef, W, H, lv, rn, ct)
    elif from_i16 is None:
        api.fwd_u8_records(d_img, W, H, lv, rn, ct, lut=qtable)
    else:
        api.fwd_i16_records(from_i16, W, H, lv, rn, ct, lut=qtable)
    stride = api.huffman_seg_stride(W)
    seg = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    nb = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
    api.huffman_rows(lv, rn, ct, W, H, seg, nb, chroma=chroma)
    torch.cuda.synchronize()
    return coef, (lv, rn, ct), dict(segments=seg.cpu().numpy(), seg_bytes=nb.cpu().numpy().astype(np.uint32), seg_stride=stride, blocks_per_row=W // 8, qtable=qtable)


@pytest.mark.gpu
def test_huffman_kernel_equals_the_checker():
    api.init(0)
    rng = np.random.default_rng(3)
    for (W, H) in ((8, 8), (64, 16), (264, 24), (2048, 64), (4104, 16)):  # one lane ... several 256-block chunks with a ragged tail
        for kind, q in (("photo", K1_LUMA), ("noise", np.ones(64, dtype=np.float32)), ("photo", K2_CHROMA)):
            img = synth.plane_u8_np(W, H, kind, seed=W + H)
            chroma = q is K2_CHROMA
            coef, (lv, rn, ct), comp = _encode_gpu(_dev(img), W, H, q, chroma=chroma)
            seg, nb, stride = O.huffman_rows(lv.cpu().numpy(), rn.cpu().numpy(), ct.cpu().numpy(), W, H, chroma=chroma)
            assert np.array_equal(comp["seg_bytes"], nb), (W, H, kind)
            for r in range(H // 8):
                assert np.array_equal(comp["segments"][r * stride:r * stride + nb[r]], seg[r * stride:r * stride + nb[r]]), (W, H, kind, r)
    # arbitrary (not DCT-shaped) records: dense blocks, saturating levels, empty blocks; a sub-range leaves the other rows alone
    W, H = 512, 32
    coef = (rng.integers(-1500, 1500, (H, W)) * (rng.random((H, W)) < 0.5)).astype(np.int16)
    coef[:8, :64] = 0
    lvh, rnh, cth = O.zigzag_rle("i16", coef, W, H)
    stride = api.huffman_seg_stride(W)
    seg = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    nb = torch.full((H // 8,), -1, dtype=torch.int32, device="cuda")
    api.huffman_rows(_dev(lvh), _dev(rnh), _dev(cth), W, H, seg, nb, by0=1, by1=3)
    want_seg, want_nb, _ = O.huffman_rows(lvh, rnh, cth, W, H, by0=1, by1=3, fill=0x5A)
    got_nb = nb.cpu().numpy()
    assert got_nb[0] == -1 and got_nb[3] == -1 and np.array_equal(got_nb[1:3].astype(np.uint32), want_nb[1:3])
    g = seg.cpu().numpy()
    for r in (1, 2):
        assert np.array_equal(g[r * stride:r * stride + want_nb[r]], want_seg[r * stride:r * stride + want_nb[r]])
    assert (g[:stride] == 0x5A).all() and (g[3 * stride:] == 0x5A).all()
    # garbage in: random bytes as records (zero levels, runs up to 255, counts up to 255).  Such a record is coded as its DC
    # alone, so no row outgrows its segment, and the checker agrees byte for byte
    W, H = 2304, 24
    nblk = (W // 8) * (H // 8)
    lvh = rng.integers(-32768, 32768, (nblk, 64)).astype(np.int16)
    lvh[rng.random((nblk, 64)) < 0.05] = 0
    rnh = rng.integers(0, 256, (nblk, 64)).astype(np.uint8)
    rnh[rng.random((nblk, 64)) < 0.9] = 0  # mostly dense, so that a good share of the records stays valid
    cth = rng.integers(0, 256, nblk).astype(np.uint8)
    cth[rng.random(nblk) < 0.7] //= 8
    stride = api.huffman_seg_stride(W)
    seg = torch.full(((H // 8) * stride + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    nb = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
    api.huffman_rows(_dev(lvh), _dev(rnh), _dev(cth), W, H, seg, nb)
    want_seg, want_nb, _ = O.huffman_rows(lvh, rnh, cth, W, H, fill=0x5A)
    g, got_nb = seg.cpu().numpy(), nb.cpu().numpy().astype(np.uint32)
    assert np.array_equal(got_nb, want_nb) and (got_nb <= stride - 8).all()
    for r in range(H // 8):
        assert np.array_equal(g[r * stride:r * stride + want_nb[r]], want_seg[r * stride:r * stride + want_nb[r]]), r
    assert (g[(H // 8) * stride:] == 0x5A).all()


@pytest.mark.gpu
def test_fused_pixels_to_huffman_rows_equals_the_staged_path_and_the_checker():
    """mdct_fwd_u8_huffman_rows / mdct_fwd_i16_huffman_rows (one kernel, records only in LDS): byte for byte the row segments of
    mdct_fwd_u8_records + mdct_huffman_rows on the GPU AND of the checker's composition -- one lane ... several 256-block
    chunks with ragged tails, photo at the Annex K tables, noise at quantiser 1 (64 pairs per block, ZRL, several ring
    windows), flat planes (DC only: the late DC predictor across waves and chunks), chroma tables, no level shift, sub-ranges"""
    api.init(0)
    for (W, H) in ((8, 8), (64, 16), (264, 24), (520, 8), (2048, 64), (4104, 16), (8192, 32)):
        for kind, q, shift in (("photo", K1_LUMA, True), ("noise", np.ones(64, dtype=np.float32), True), ("photo", K2_CHROMA, True), ("flat", K1_LUMA, False), ("photo", None, True)):
            if kind == "flat":  # a ramp of constant blocks: every block is its DC alone, every DC difference is non-zero
                img = (np.arange(W // 8, dtype=np.int64)[None, :] * 37 + np.arange(H // 8, dtype=np.int64)[:, None] * 11) % 256
                img = np.ascontiguousarray(np.kron(img, np.ones((8, 8), dtype=np.int64)).astype(np.uint8))
            else:
                img = synth.plane_u8_np(W, H, kind, seed=W + H)
            chroma = q is K2_CHROMA
            stride = api.huffman_seg_stride(W)
            nblk = (W // 8) * (H // 8)
            d_img = _dev(img)
            # staged on the GPU
            lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
            rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
            ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda")
            api.fwd_u8_records(d_img, W, H, lv, rn, ct, lut=q, level_shift=shift)
            seg_a = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
            nb_a = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
            api.huffman_rows(lv, rn, ct, W, H, seg_a, nb_a, chroma=chroma)
            # fused
            seg_b = torch.full(((H // 8) * stride + 64,), 0x5A, dtype=torch.uint8, device="cuda")
            nb_b = torch.full((H // 8,), -1, dtype=torch.int32, device="cuda")
            ff_b = torch.full((H // 8,), -1, dtype=torch.int32, device="cuda")
            api.fwd_u8_huffman_rows(d_img, W, H, seg_b, nb_b, lut=q, level_shift=shift, chroma=chroma, ff_counts=ff_b)
            torch.cuda.synchronize()
            na, nbb = nb_a.cpu().numpy(), nb_b.cpu().numpy()
            assert np.array_equal(na, nbb), (W, H, kind)
            ga, gb = seg_a.cpu().numpy(), seg_b.cpu().numpy()
            for r in range(H // 8):
                assert np.array_equal(ga[r * stride:r * stride + na[r]], gb[r * stride:r * stride + na[r]]), (W, H, kind, r)
            assert (gb[(H // 8) * stride:] == 0x5A).all()
            # the 0xFF bytes counted on the way out, and the packing that uses them instead of its own counting pass
            ffh = ff_b.cpu().numpy()
            assert [int((gb[r * stride:r * stride + na[r]] == 0xFF).sum()) for r in range(H // 8)] == ffh.tolist(), (W, H, kind)
            cap = int(na.sum()) * 2 + 2 * (H // 8) + 16
            scan_a = torch.full((cap,), 0x33, dtype=torch.uint8, device="cuda")
            scan_b = torch.full((cap,), 0x33, dtype=torch.uint8, device="cuda")
            off_a = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
            off_b = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
            api.jpeg_pack_rows(seg_a, nb_a, stride, H // 8, scan_a, off_a, first_rst=3)
            api.jpeg_pack_rows(seg_b[:(H // 8) * stride], nb_b, stride, H // 8, scan_b, off_b, first_rst=3, ff_counts=ff_b)
            torch.cuda.synchronize()
            assert torch.equal(off_a, off_b) and torch.equal(scan_a, scan_b), (W, H, kind)
            # the checker's composition (CPU), on the smaller cases
            if W * H <= 2048 * 64:
                lvh, rnh, cth = O.u8_records(img, W, H, lut=q, level_shift=shift)
                want_seg, want_nb, _ = O.huffman_rows(lvh, rnh, cth, W, H, chroma=chroma)
                assert np.array_equal(nbb.astype(np.uint32), want_nb), (W, H, kind)
                for r in range(H // 8):
                    assert np.array_equal(gb[r * stride:r * stride + want_nb[r]], want_seg[r * stride:r * stride + want_nb[r]]), (W, H, kind, r)
    # the largest coefficients 8-bit pixels can produce -- block (u, v) is the sign pattern of basis function (u, v), 0 / 255 -- through the
    # smallest table for which the fused kernel leaves its saturations out (every entry 1.01): still the staged path's bytes, with and
    # without the level shift, and through a table just below the threshold (which keeps them)
    xs = np.arange(8)
    cosm = np.cos((2 * xs[None, :] + 1) * xs[:, None] * np.pi / 16)  # [u][x]
    worst = np.zeros((16, 512), dtype=np.uint8)
    for u in range(8):
        for v in range(8):
            blk = np.where(np.outer(cosm[v], cosm[u]) > 0, 255, 0).astype(np.uint8)  # [y][x]
            worst[0:8, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = blk
            worst[8:16, (u * 8 + v) * 8:(u * 8 + v) * 8 + 8] = 255 - blk
    W, H = 512, 16
    stride = api.huffman_seg_stride(W)
    nblk = (W // 8) * (H // 8)
    d_w = _dev(worst)
    for q, shift in ((np.full(64, 1.01, dtype=np.float32), True), (np.full(64, 1.01, dtype=np.float32), False), (np.full(64, 1.0, dtype=np.float32), True),
                     (np.where(np.arange(64) == 37, 1.0, 1.01).astype(np.float32), True)):
        lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
        rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
        ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda")
        api.fwd_u8_records(d_w, W, H, lv, rn, ct, lut=q, level_shift=shift)
        assert int(lv[:, 1:].abs().max().item()) >= 1000  # the pattern does reach the edge of the AC range
        seg_a = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
        nb_a = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
        api.huffman_rows(lv, rn, ct, W, H, seg_a, nb_a)
        seg_b = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
        nb_b = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
        api.fwd_u8_huffman_rows(d_w, W, H, seg_b, nb_b, lut=q, level_shift=shift)
        torch.cuda.synchronize()
        assert torch.equal(nb_a, nb_b) and torch.equal(seg_a, seg_b), (float(q.min()), shift)
    # an int16 plane as input (the chroma planes of mdct_split420_u8), a pitched plane, a sub-range that leaves the other rows alone
    W, H, pitch = 1032, 48, 1040
    s16 = np.zeros((H, pitch), dtype=np.int16)
    s16[:, :W] = synth.plane_i16_np(W, H, "photo", seed=9)
    d16 = _dev(s16)
    nblk = (W // 8) * (H // 8)
    stride = api.huffman_seg_stride(W)
    lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda")
    api.fwd_i16_records(d16, W, H, lv, rn, ct, lut=K2_CHROMA, pitch=pitch)
    seg_a = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    nb_a = torch.full((H // 8,), -1, dtype=torch.int32, device="cuda")
    api.huffman_rows(lv, rn, ct, W, H, seg_a, nb_a, chroma=True, by0=2, by1=5)
    seg_b = torch.full(((H // 8) * stride,), 0x5A, dtype=torch.uint8, device="cuda")
    nb_b = torch.full((H // 8,), -1, dtype=torch.int32, device="cuda")
    api.fwd_i16_huffman_rows(d16, W, H, seg_b, nb_b, lut=K2_CHROMA, chroma=True, by0=2, by1=5, pitch=pitch)
    torch.cuda.synchronize()
    assert torch.equal(nb_a, nb_b) and (nb_b[:2] == -1).all() and (nb_b[5:] == -1).all()
    ga, gb, na = seg_a.cpu().numpy(), seg_b.cpu().numpy(), nb_a.cpu().numpy()
    for r in range(2, 5):
        assert np.array_equal(ga[r * stride:r * stride + na[r]], gb[r * stride:r * stride + na[r]]), r
    assert (gb[:2 * stride] == 0x5A).all() and (gb[5 * stride:] == 0x5A).all()
    # argument checks
    assert api.fwd_u8_huffman_rows(d16, W + 4, H, seg_b, nb_b, check=False) == 2
    assert api.fwd_u8_huffman_rows(d16, W, H, seg_b, nb_b, seg_stride=stride - 4, check=False) == 1


@pytest.mark.gpu
def test_gpu_pipeline_writes_jpegs_libjpeg_opens():
    """pixels -> mdct_fwd_u8_i16 (Annex K tables) -> mdct_zigzag_rle_i16 -> mdct_huffman_rows -> JFIF: decoded by libjpeg, the
    picture equals our own inverse transform within 1 grey level; grey at 4096x2160 and 4:2:0 colour through mdct_split420_u8"""
    api.init(0)
    W, H = 4096, 2160 - 2160 % 16
    img = synth.plane_u8_torch(W, H, "photo")
    coef, _, comp = _encode_gpu(img, W, H, K1_LUMA)
    data = jfif.write_jpeg([comp], W, H)
    dec = np.asarray(Image.open(io.BytesIO(data)).convert("L"))
    own = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    api.inv_i16_u8(coef, own, W, H, lut=K1_LUMA)
    assert np.abs(dec.astype(int) - own.cpu().numpy().astype(int)).max() <= 1
    assert len(data) < W * H // 4  # and it is compressed
    # colour: interleaved YCbCr -> split420 -> three planes -> three non-interleaved scans
    W, H = 640, 480
    ycc = torch.stack([synth.plane_u8_torch(W, H, "photo", seed=s) for s in (5, 6, 7)], dim=-1).contiguous()
    y = torch.empty((H, W), dtype=torch.int16, device="cuda")
    cb = torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda")
    cr = torch.empty_like(cb)
    api.split420_u8(ycc, W, H, y, cb, cr)
    comps, coefs = [], []
    for plane, (w, h), q, chroma in ((y, (W, H), K1_LUMA, False), (cb, (W // 2, H // 2), K2_CHROMA, True), (cr, (W // 2, H // 2), K2_CHROMA, True)):
        c, _, comp = _encode_gpu(None, w, h, q, chroma=chroma, from_i16=plane, fused=True)
        comps.append(comp)
        coefs.append(c)
    data = jfif.write_jpeg(comps, W, H)
    im = Image.open(io.BytesIO(data))
    im.draft("YCbCr", im.size)  # libjpeg's native output: no YCbCr -> RGB -> YCbCr round trip (and its gamut clipping) in between
    assert im.mode == "YCbCr"
    dec = np.asarray(im)
    assert dec.shape == (H, W, 3)
    own_y = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    api.inv_i16_u8(coefs[0], own_y, W, H, lut=K1_LUMA)
    assert np.abs(dec[:, :, 0].astype(int) - own_y.cpu().numpy().astype(int)).max() <= 1
    # chroma comes back upsampled by libjpeg (its own interpolation): compare at the sample centres, loosely
    own_cb = torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda")
    api.inv_i16_u8(coefs[1], own_cb, W // 2, H // 2, lut=K2_CHROMA)
    box = dec[:, :, 1].astype(float).reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
    assert np.abs(box - own_cb.cpu().numpy().astype(float)).mean() < 2.0


@pytest.mark.gpu
def test_encoder_stages_replay_from_one_hip_graph():
    """a small frame is launch-bound: pixels -> coefficients -> records -> Huffman rows captured once as a hipGraph
    (the entry points neither allocate nor synchronise) and replayed on new pictures; every replay decodes with libjpeg"""
    api.init(0)
    W, H = 1920, 1080 - 1080 % 8
    img = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    coef = torch.empty((H, W), dtype=torch.int16, device="cuda")
    nblk = (W // 8) * (H // 8)
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    stride = api.huffman_seg_stride(W)
    seg = torch.zeros(((H // 8) * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")

    def stages():
        api.fwd_u8_i16(img, coef, W, H, lut=K1_LUMA)
        api.zigzag_rle_i16(coef, W, H, lv, rn, ct)
        api.huffman_rows(lv, rn, ct, W, H, seg, nb)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # warm up outside capture (lazy device probe)
        stages()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stages()
    for seed in (21, 22):
        pic = synth.plane_u8_np(W, H, "photo", seed=seed)
        img.copy_(_dev(pic))
        g.replay()
        torch.cuda.synchronize()
        comp = dict(segments=seg.cpu().numpy(), seg_bytes=nb.cpu().numpy().astype(np.uint32), seg_stride=stride, blocks_per_row=W // 8, qtable=K1_LUMA)
        dec = np.asarray(Image.open(io.BytesIO(jfif.write_jpeg([comp], W, H))).convert("L"))
        own = torch.empty((H, W), dtype=torch.uint8, device="cuda")
        api.inv_i16_u8(coef, own, W, H, lut=K1_LUMA)
        assert np.abs(dec.astype(int) - own.cpu().numpy().astype(int)).max() <= 1, seed
        assert np.abs(dec.astype(int) - pic.astype(int)).mean() < 16  # and it is this picture (whose +-24 noise K.1 quantises away)


@pytest.mark.gpu
def test_one_launch_pixels_to_scan_equals_the_two_launch_path():
    """mdct_fwd_u8_jpeg_scan / mdct_fwd_i16_jpeg_scan (the fused kernel with the packing as its tail: rows chained through row_work):
    scan and row offsets byte for byte those of mdct_fwd_*_huffman_rows + mdct_jpeg_pack_rows_counted -- and of the checker's packing of
    the same segments; the SAME row_work over many calls, other row counts, a replayed hipGraph, scans at odd addresses, a capacity that
    cuts the scan, sub-ranges, more rows than are resident at once"""
    api.init(0)
    work = torch.zeros((4096 + 2,), dtype=torch.int64, device="cuda")  # zeroed ONCE; every call below reuses it
    cases = [(8, 8, "photo"), (64, 16, "noise"), (264, 24, "photo"), (2048, 64, "noise"), (4104, 16, "photo"), (1000, 72, "flat"), (8192, 32, "photo"), (64, 32768, "photo"), (8, 8, "noise")]
    for n_case, (W, H, kind) in enumerate(cases):
        q = np.ones(64, dtype=np.float32) if kind == "noise" else K1_LUMA
        if kind == "flat":
            img = (np.arange(W // 8, dtype=np.int64)[None, :] * 37 + np.arange(H // 8, dtype=np.int64)[:, None] * 11) % 256
            img = np.ascontiguousarray(np.kron(img, np.ones((8, 8), dtype=np.int64)).astype(np.uint8))
        else:
            img = synth.plane_u8_np(W, H, kind, seed=W + H)
        n = H // 8
        stride = api.huffman_seg_stride(W)
        d_img = _dev(img)
        seg = torch.full((n * stride,), 0x5A, dtype=torch.uint8, device="cuda")
        nb = torch.zeros((n,), dtype=torch.int32, device="cuda")
        ff = torch.zeros((n,), dtype=torch.int32, device="cuda")
        api.fwd_u8_huffman_rows(d_img, W, H, seg, nb, lut=q, ff_counts=ff)
        total = int(nb.sum().item()) + int(ff.sum().item()) + 2 * (n - 1)
        want = torch.full((total + 24,), 0x33, dtype=torch.uint8, device="cuda")
        woff = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
        rst = n_case % 8
        api.jpeg_pack_rows(seg, nb, stride, n, want[:total], woff, first_rst=rst, ff_counts=ff)
        assert int(woff[-1].item()) == total
        if W * H <= 2048 * 64:  # and the checker's packing of those segments
            cw, coff = O.jpeg_pack_rows(seg.cpu().numpy(), nb.cpu().numpy().astype(np.uint32), stride, first_rst=rst)
            assert np.array_equal(coff.astype(np.int64), woff.cpu().numpy()) and np.array_equal(cw[:total], want[:total].cpu().numpy())
        seg_w = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
        for shift in ((0, 1, 3) if n <= 64 else (2,)):
            got = torch.full((total + 24,), 0x33, dtype=torch.uint8, device="cuda")
            off = torch.full((n + 1,), -7, dtype=torch.int64, device="cuda")
            api.fwd_u8_jpeg_scan(d_img, W, H, seg_w, work, got[shift:], off, lut=q, first_rst=rst, out_capacity=total)
            torch.cuda.synchronize()
            assert torch.equal(off, woff), (W, H, kind, shift)
            assert torch.equal(got[shift: shift + total], want[:total]) and (got[:shift] == 0x33).all() and (got[shift + total:] == 0x33).all(), (W, H, kind, shift)
        if n > 3:  # a capacity that ends inside row 2: rows 0 and 1 arrive, nothing else is touched, the offsets still tell the whole length
            cut = int(woff[2].item()) + 1
            got = torch.full((total + 24,), 0x33, dtype=torch.uint8, device="cuda")
            api.fwd_u8_jpeg_scan(d_img, W, H, seg_w, work, got, off, lut=q, first_rst=rst, out_capacity=cut)
            torch.cuda.synchronize()
            assert torch.equal(off, woff) and torch.equal(got[: int(woff[2].item())], want[: int(woff[2].item())]) and (got[int(woff[2].item()):] == 0x33).all(), (W, H, kind)
    # the same launch captured once and replayed: the epoch of the chain lives in row_work, not in the kernel arguments
    W, H = 2048, 512
    n, stride = H // 8, api.huffman_seg_stride(W)
    imgs = [synth.plane_u8_torch(W, H, "photo", seed=40 + i) for i in range(3)]
    cur = torch.empty_like(imgs[0])
    seg_w = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    got = torch.zeros((W * H,), dtype=torch.uint8, device="cuda")
    off = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        api.fwd_u8_jpeg_scan(cur, W, H, seg_w, work, got, off, lut=K1_LUMA)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        api.fwd_u8_jpeg_scan(cur, W, H, seg_w, work, got, off, lut=K1_LUMA)
    seg = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((n,), dtype=torch.int32, device="cuda")
    ff = torch.zeros((n,), dtype=torch.int32, device="cuda")
    want = torch.zeros((W * H,), dtype=torch.uint8, device="cuda")
    woff = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    for rep in range(6):
        cur.copy_(imgs[rep % 3])
        got.zero_()
        g.replay()
        api.fwd_u8_huffman_rows(cur, W, H, seg, nb, lut=K1_LUMA, ff_counts=ff)
        want.zero_()
        api.jpeg_pack_rows(seg, nb, stride, n, want, woff, ff_counts=ff)
        torch.cuda.synchronize()
        assert torch.equal(off, woff) and torch.equal(got, want), rep
    # an int16 plane, pitched, chroma tables, a sub-range of the block rows (row offsets count from the sub-range's first row)
    W, H, pitch = 1032, 48, 1040
    s16 = np.zeros((H, pitch), dtype=np.int16)
    s16[:, :W] = synth.plane_i16_np(W, H, "photo", seed=9)
    d16 = _dev(s16)
    stride = api.huffman_seg_stride(W)
    seg = torch.empty(((H // 8) * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
    ff = torch.zeros((H // 8,), dtype=torch.int32, device="cuda")
    api.fwd_i16_huffman_rows(d16, W, H, seg, nb, lut=K2_CHROMA, chroma=True, by0=2, by1=5, pitch=pitch, ff_counts=ff)
    want = torch.full((W * H,), 0x33, dtype=torch.uint8, device="cuda")
    woff = torch.zeros((4,), dtype=torch.int64, device="cuda")
    api.jpeg_pack_rows(seg[2 * stride:], nb[2:], stride, 3, want, woff, first_rst=2, ff_counts=ff[2:])
    got = torch.full((W * H,), 0x33, dtype=torch.uint8, device="cuda")
    off = torch.zeros((4,), dtype=torch.int64, device="cuda")
    seg_w = torch.empty(((H // 8) * stride,), dtype=torch.uint8, device="cuda")
    api.fwd_i16_jpeg_scan(d16, W, H, seg_w, work, got, off, lut=K2_CHROMA, chroma=True, by0=2, by1=5, pitch=pitch, first_rst=2)
    torch.cuda.synchronize()
    assert torch.equal(off, woff) and torch.equal(got, want)
    # argument checks
    assert api.fwd_u8_jpeg_scan(d_img, 8, 8, seg_w, None, got, off, check=False) == 1
    assert api.fwd_u8_jpeg_scan(d_img, 8, 8, seg_w, work, got, off, by0=1, by1=1, check=False) == 1
    assert api.fwd_u8_jpeg_scan(d_img, 8, 8, seg_w, work, got, off, first_rst=8, check=False) == 1
    assert api.fwd_u8_jpeg_scan(d_img, 8, 8, seg_w, work[1:].view(torch.uint8)[4:], got, off, check=False) == 1


@pytest.mark.gpu
def test_full_occupancy_slices_of_the_soaks():
    """Always on (the long forms stay opt-in below): every kernel of the library on whole 8192^2 / 4104-wide / 2048-wide planes, 24 launches
    each compared on the device with the first (tools/soak_determinism.py), and 4 s of the one-launch encoder against the two-launch
    path on random sizes, tables and contents (tools/soak_jpeg_scan.py).  The small differential cases of the other tests never fill
    the chip; the barrier without its LDS wait of round 3 (csrc/wg_sync.h) only showed with every CU holding several dense rows."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cmd, ok in (([sys.executable, os.path.join(root, "tools", "soak_determinism.py"), "24"], "soak ok"), ([sys.executable, os.path.join(root, "tools", "soak_jpeg_scan.py"), "4"], None)):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "!!" not in r.stdout and (ok is None or ok in r.stdout), r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_one_launch_scan_failure_word_is_sticky_and_tall_planes_take_two_launches():
    """(i) row_work[1] is the chain's sticky failure word: with it set (as a row that ran out of patience leaves it) a call codes nothing
    and ends with row_offsets[n_rows] == UINT64_MAX -- the one failure indicator include/mdct.h names -- call after call, until the caller
    zeroes row_work; then the same array works again.  (ii) more than 16384 block rows: the call runs the fused coder and the counted
    packing as two launches (the one-launch form's look-back is quadratic in the row count), same scan and offsets, and leaves row_work
    fit for one-launch calls."""
    api.init(0)
    W, H = 512, 128
    n, stride = H // 8, api.huffman_seg_stride(W)
    img = synth.plane_u8_torch(W, H, "photo", seed=5)
    work = torch.zeros((n + 2,), dtype=torch.int64, device="cuda")
    seg_w = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    good = torch.zeros((W * H,), dtype=torch.uint8, device="cuda")
    goff = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    api.fwd_u8_jpeg_scan(img, W, H, seg_w, work, good, goff, lut=K1_LUMA)
    torch.cuda.synchronize()
    total = int(goff[-1].item())
    assert 0 < total < W * H
    work[1] = 1  # what a row whose predecessors did not publish in time leaves behind
    for rep in range(2):
        got = torch.full((W * H,), 0x33, dtype=torch.uint8, device="cuda")
        off = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
        assert api.fwd_u8_jpeg_scan(img, W, H, seg_w, work, got, off, lut=K1_LUMA, check=False) == 0  # the launch itself succeeds ...
        torch.cuda.synchronize()
        assert int(off[-1].item()) == -1 and (got == 0x33).all()  # ... its result says failure (UINT64_MAX), and no scan byte was written
    work.zero_()
    got = torch.zeros((W * H,), dtype=torch.uint8, device="cuda")
    off = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    api.fwd_u8_jpeg_scan(img, W, H, seg_w, work, got, off, lut=K1_LUMA)
    torch.cuda.synchronize()
    assert torch.equal(off, goff) and torch.equal(got, good)
    # (ii) 16 px wide, 17000 block rows
    W2, H2 = 16, 17000 * 8
    n2, stride2 = H2 // 8, api.huffman_seg_stride(W2)
    tall = synth.plane_u8_torch(W2, H2, "photo", seed=6)
    seg = torch.empty((n2 * stride2,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((n2,), dtype=torch.int32, device="cuda")
    ff = torch.zeros((n2,), dtype=torch.int32, device="cuda")
    api.fwd_u8_huffman_rows(tall, W2, H2, seg, nb, lut=K1_LUMA, ff_counts=ff)
    want = torch.zeros((W2 * H2,), dtype=torch.uint8, device="cuda")
    woff = torch.zeros((n2 + 1,), dtype=torch.int64, device="cuda")
    api.jpeg_pack_rows(seg, nb, stride2, n2, want, woff, ff_counts=ff)
    work2 = torch.zeros((n2 + 2,), dtype=torch.int64, device="cuda")
    seg2 = torch.empty((n2 * stride2,), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        got2 = torch.zeros((W2 * H2,), dtype=torch.uint8, device="cuda")
        off2 = torch.zeros((n2 + 1,), dtype=torch.int64, device="cuda")
        api.fwd_u8_jpeg_scan(tall, W2, H2, seg2, work2, got2, off2, lut=K1_LUMA)
        torch.cuda.synchronize()
        assert torch.equal(off2, woff) and torch.equal(got2, want), rep
        assert (work2 == 0).all()  # left as the caller zeroed it: a one-launch call may use it next
    got = torch.zeros((W * H,), dtype=torch.uint8, device="cuda")
    api.fwd_u8_jpeg_scan(img, W, H, seg_w, work2, got, off, lut=K1_LUMA)  # the same work array, now with few rows: the chained form
    torch.cuda.synchronize()
    assert torch.equal(off, goff) and torch.equal(got, good)
    # a sub-range of a tall plane (row numbers of the plane index the segments; offsets count from the sub-range)
    got2 = torch.zeros((W2 * H2,), dtype=torch.uint8, device="cuda")
    off3 = torch.zeros((16601,), dtype=torch.int64, device="cuda")
    api.fwd_u8_jpeg_scan(tall, W2, H2, seg2, work2, got2, off3, lut=K1_LUMA, by0=300, by1=16900, first_rst=300 % 8)
    torch.cuda.synchronize()
    base = int(woff[300].item())
    assert torch.equal(off3[:-1], woff[300:16900] - base) and int(off3[-1].item()) == int(woff[16900].item()) - base - 2  # (the last row of a range carries no restart marker)
    assert torch.equal(got2[: int(off3[-1].item())], want[base: int(woff[16900].item()) - 2])


@pytest.mark.gpu
def test_dense_rows_that_need_several_ring_windows_code_the_same_bytes_every_time():
    """Rows whose code does not fit the coder's LDS ring at once go through it in windows; between two windows the slots are cleared and
    OR-ed into again.  The barrier between them once came out of the compiler without its LDS wait, and under the LDS traffic of the
    one-launch encoder (other rows of the CU copying their segments) a clear still in flight overtook the next window's first ds_or:
    one launch in a few hundred lost bits of a word (found by tools/soak_jpeg_scan.py; explicit s_waitcnt in huffman_rows.h since).
    2048-wide planes (one 256-block chunk per row) at ~7.5 bit/px: every launch of the one-launch encoder, the fused kernel and the staged
    coder compared with the first on the device; a sample of the first's rows against the checker"""
    api.init(0)
    W, H = 2048, 7680  # 960 rows: four workgroups on (almost) every CU at once, which is when it happened
    q = np.ones(64, dtype=np.float32)  # ~15 KiB of code per row: the words lost were among the last-cleared slots (768..1023) of the third and fourth window
    n, stride, nblk = H // 8, api.huffman_seg_stride(W), (W // 8) * (H // 8)
    work = torch.zeros((n + 2,), dtype=torch.int64, device="cuda")
    for seed in (3, 4):
        img = synth.plane_u8_torch(W, H, "photo", seed=seed)
        lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
        rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
        ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda")
        api.fwd_u8_records(img, W, H, lv, rn, ct, lut=q)
        seg0 = torch.zeros((n * stride,), dtype=torch.uint8, device="cuda")
        nb0 = torch.zeros((n,), dtype=torch.int32, device="cuda")
        ff0 = torch.zeros((n,), dtype=torch.int32, device="cuda")
        api.fwd_u8_huffman_rows(img, W, H, seg0, nb0, lut=q, ff_counts=ff0)
        assert int(nb0.min().item()) > 2 * 4096  # every row needs at least three windows of 1024 words
        total = int(nb0.sum().item()) + int(ff0.sum().item()) + 2 * (n - 1)
        scan0 = torch.zeros((total,), dtype=torch.uint8, device="cuda")
        off0 = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
        api.jpeg_pack_rows(seg0, nb0, stride, n, scan0, off0, ff_counts=ff0)
        if seed == 3:  # the checker on a few rows (its plain C coder takes its time on rows of this size)
            lvh, rnh, cth, bpr = lv.cpu().numpy(), rn.cpu().numpy(), ct.cpu().numpy(), W // 8
            for r in (0, n // 2, n - 1):
                ws, wn, _ = O.huffman_rows(lvh[r * bpr:(r + 1) * bpr], rnh[r * bpr:(r + 1) * bpr], cth[r * bpr:(r + 1) * bpr], W, 8)
                assert int(nb0[r].item()) == int(wn[0]) and np.array_equal(seg0[r * stride:r * stride + int(wn[0])].cpu().numpy(), ws[: int(wn[0])]), r
        seg = torch.zeros_like(seg0)
        nb = torch.zeros_like(nb0)
        scan = torch.zeros_like(scan0)
        off = torch.zeros_like(off0)
        bad = torch.zeros((3,), dtype=torch.int64, device="cuda")
        for rep in range(250):  # every launch compared, on the device
            api.fwd_u8_jpeg_scan(img, W, H, seg, work, scan, off, lut=q)
            bad[0] += (scan != scan0).sum() + (off != off0).sum()
            if rep % 5 == 0:
                api.fwd_u8_huffman_rows(img, W, H, seg, nb, lut=q)
                bad[1] += (seg != seg0).sum() + (nb != nb0).sum()
                api.huffman_rows(lv, rn, ct, W, H, seg, nb)
                bad[2] += (seg != seg0).sum() + (nb != nb0).sum()
        assert bad.tolist() == [0, 0, 0], (seed, bad.tolist())


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("MDCT_SOAK_JPEG"), reason="opt-in soak: MDCT_SOAK_JPEG=<seconds> python -m pytest tests -m gpu -k soak_of_the_encoders")
def test_soak_of_the_encoders():
    """opt-in: tools/soak_jpeg_scan.py (random sizes / tables / contents; one-launch encoder vs fused kernel + packing vs the staged path) and
    tools/soak_determinism.py (every kernel at full occupancy, every launch compared with the first) for MDCT_SOAK_JPEG seconds / launches x 10"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    secs = os.environ["MDCT_SOAK_JPEG"]
    for cmd in ([sys.executable, os.path.join(root, "tools", "soak_jpeg_scan.py"), secs], [sys.executable, os.path.join(root, "tools", "soak_determinism.py"), str(10 * int(float(secs)))]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]


def test_pack_checker_equals_the_host_writer():
    """orc_jpeg_pack_rows == jfif.scan_bytes (numpy) on segments full of 0xFF bytes, empty rows, a capacity that cuts the scan short"""
    rng = np.random.default_rng(9)
    stride, n = 408, 13
    nb = rng.integers(0, stride - 8, n).astype(np.uint32)
    nb[3] = 0
    seg = rng.integers(0, 256, n * stride).astype(np.uint8)
    seg[rng.random(n * stride) < 0.3] = 0xFF
    out, off = O.jpeg_pack_rows(seg, nb, stride)
    want = jfif.scan_bytes(seg, nb, stride)
    assert int(off[n]) == len(want) and out[: len(want)].tobytes() == want
    assert np.array_equal(np.diff(off.astype(np.int64))[:-1], [len(jfif.stuff(seg[r * stride:r * stride + nb[r]])) + 2 for r in range(n - 1)])
    cut = int(off[7]) + 5  # rows 0..6 fit, row 7 does not
    out2, off2 = O.jpeg_pack_rows(seg, nb, stride, capacity=cut, fill=0x77)
    assert np.array_equal(off2, off) and out2[: int(off[7])].tobytes() == want[: int(off[7])] and (out2[int(off[7]):] == 0x77).all()
    out3, off3 = O.jpeg_pack_rows(seg, nb, stride, first_rst=5)
    assert out3[int(off[1]) - 2: int(off[1])].tolist() == [0xFF, 0xD5] and out3[int(off[4]) - 2: int(off[4])].tolist() == [0xFF, 0xD0]


@pytest.mark.gpu
def test_device_packed_scan_is_the_file_libjpeg_opens():
    """mdct_jpeg_pack_rows: byte-equal to the checker (random segments incl. all-0xFF and empty rows, capacity cut, first_rst), and a grey
    JPEG whose scan was stuffed and joined on the device decodes like the one stuffed on the host"""
    api.init(0)
    rng = np.random.default_rng(10)
    for (stride, n) in ((16, 1), (408, 13), (412, 21), (4104, 300), (213000, 40), (16, 20000)):  # the last: more rows than the one-launch form takes
        nb = rng.integers(0, stride - 8, n).astype(np.uint32)
        nb[0] = stride - 8
        if n > 3:
            nb[3] = 0
        seg = rng.integers(0, 256, n * stride).astype(np.uint8)
        seg[rng.random(n * stride) < 0.2] = 0xFF
        if n > 5:
            seg[5 * stride:6 * stride] = 0xFF
        want, woff = O.jpeg_pack_rows(seg, nb, stride, first_rst=n % 8, fill=0x77)
        out = torch.full((len(want) + 16,), 0x77, dtype=torch.uint8, device="cuda")
        off = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
        api.jpeg_pack_rows(_dev(seg), _dev(nb.view(np.int32)), stride, n, out, off, first_rst=n % 8, out_capacity=len(want))
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), woff), (stride, n)
        g = out.cpu().numpy()
        assert np.array_equal(g[: len(want)], want) and (g[len(want):] == 0x77).all(), (stride, n)
        # the counted form (one launch: every workgroup sums the lengths before its own row), into a scan that starts at any byte address
        ff = np.array([int((seg[r * stride:r * stride + nb[r]] == 0xFF).sum()) for r in range(n)], dtype=np.int32)
        for shift in (0, 1, 2, 3):
            out2 = torch.full((len(want) + 16,), 0x77, dtype=torch.uint8, device="cuda")
            off.zero_()
            api.jpeg_pack_rows(_dev(seg), _dev(nb.view(np.int32)), stride, n, out2[shift:], off, first_rst=n % 8, out_capacity=len(want), ff_counts=_dev(ff))
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), woff), (stride, n, shift)
            g = out2.cpu().numpy()
            assert (g[:shift] == 0x77).all() and np.array_equal(g[shift: shift + len(want)], want) and (g[shift + len(want):] == 0x77).all(), (stride, n, shift)
        if n > 8:  # a capacity that ends inside row 7: rows 0..6 arrive, nothing else is touched
            cut = int(woff[7]) + 3
            out.fill_(0x77)
            api.jpeg_pack_rows(_dev(seg), _dev(nb.view(np.int32)), stride, n, out, off, first_rst=n % 8, out_capacity=cut)
            g = out.cpu().numpy()
            assert int(off[n].item()) == int(woff[n]) and np.array_equal(g[: int(woff[7])], want[: int(woff[7])]) and (g[int(woff[7]):] == 0x77).all()
    W, H = 4096, 2160 - 2160 % 16
    img = synth.plane_u8_torch(W, H, "photo")
    coef, _, comp = _encode_gpu(img, W, H, K1_LUMA)
    d_seg, d_nb = _dev(comp["segments"]), _dev(comp["seg_bytes"].view(np.int32))
    scan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
    off = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
    api.jpeg_pack_rows(d_seg, d_nb, comp["seg_stride"], H // 8, scan, off)
    total = int(off[-1].item())
    assert total <= scan.numel()
    packed = dict(scan=scan[:total].cpu().numpy(), blocks_per_row=W // 8, qtable=K1_LUMA)
    a, b = jfif.write_jpeg([packed], W, H), jfif.write_jpeg([comp], W, H)
    assert a == b
    assert np.asarray(Image.open(io.BytesIO(a)).convert("L")).shape == (H, W)


C_EXAMPLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mdct_jpeg")


@pytest.mark.gpu
def test_plain_c_caller_writes_the_same_file(tmp_path):
    """tools/mdct_jpeg.c -- gcc, the C-ABI and the HIP runtime only -- writes byte for byte the JPEG the Python host writes
    from the same pixels, and libjpeg decodes it (the host side of the stages stays C)"""
    root = os.path.dirname(C_EXAMPLE[: -len("/mdct_jpeg")])
    r = subprocess.run(["make", "-s", "-C", root, "jpeg_example"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(C_EXAMPLE), r.stderr[-600:]
    W, H = 1024, 768
    pic = synth.plane_u8_np(W, H, "photo", seed=31)
    raw, out = tmp_path / "in.raw", tmp_path / "out.jpg"
    pic.tofile(raw)
    r = subprocess.run([C_EXAMPLE, str(out), str(raw), str(W), str(H)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    api.init(0)
    _, _, comp = _encode_gpu(_dev(pic), W, H, K1_LUMA, fused=True)
    want = jfif.write_jpeg([comp], W, H)
    got = out.read_bytes()
    assert got == want, (len(got), len(want))
    assert np.asarray(Image.open(io.BytesIO(got)).convert("L")).shape == (H, W)
