"""Every kernel of the library at full occupancy, launch after launch on the same input, every output compared on the device with the first
launch's (whose bytes the parity tests pin): a race that needs a whole chip's worth of resident workgroups to show -- the kind the small
differential cases (tests/test_gpu_parity.py::test_soak_random_differential) cannot provoke -- shows up as a non-zero count.
    python3 tools/soak_determinism.py [launches per kernel, default 200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M.init(0)
K1 = synth.JPEG_LUMA
lut2000 = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
ones = np.ones(64, dtype=np.float32)
failed = 0


def soak(name, outs, launch):
    """outs: list of output tensors; launch(): one launch writing them"""
    global failed
    for o in outs:  # the same fill before every launch, the first included: the SSE encq tier leaves half of every block pair untouched
        o.fill_(0x5A if o.dtype == torch.uint8 else 23)
    launch()
    torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    t0 = time.time()
    for i in range(N):
        for o in outs:
            o.fill_(0x5A if o.dtype == torch.uint8 else 23)
        launch()
        for o, f in zip(outs, first):
            bad += (o != f).sum()
    nbad = int(bad.item())
    failed += nbad != 0
    print(f"{'!! ' if nbad else ''}{name:58s} {N} launches, {nbad} differing elements, {time.time() - t0:.1f} s", flush=True)


for (W, H) in ((8192, 8192), (4104, 2056 - 2056 % 16), (2048, 7680)):
    print(f"--- {W} x {H}")
    u8 = synth.plane_u8_torch(W, H, "photo", seed=11)
    i16 = synth.plane_i16_torch(W, H, "photo", seed=12)
    nblk = (W // 8) * (H // 8)
    o8 = torch.empty((W * H,), dtype=torch.uint8, device="cuda")
    if W % 64 == 0:
        soak("q32 / AVX2 tier", [o8], lambda: M.fwd_quant_u8(u8, o8, lut2000, W, H, 0, H // 8))
    for lname, layout, prof, rows in (("stereo / SSE", M.LAYOUT_STEREO, M.PROFILE_REF_SSE, H // 16), ("stereo / scalar", M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, H // 16),
                                      ("encq / SSE", M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, H // 8), ("encq / scalar", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, H // 8)):
        if W % 16 == 0:
            soak(lname + " tier", [o8], lambda: M.fwd_quant_u8(u8, o8, lut8, W, H, 0, rows, layout=layout, profile=prof))
    o16 = torch.empty((H, W), dtype=torch.int16, device="cuda")
    for table, tn in ((None, ""), (K1, " + table")):
        soak("i16 round trip" + tn, [o16], lambda: M.roundtrip_i16(i16, o16, W, H, lut=table))
        soak("i16 forward" + tn, [o16], lambda: M.fwd_i16(i16, o16, W, H, lut=table))
        soak("i16 inverse" + tn, [o16], lambda: M.inv_i16(i16, o16, W, H, lut=table))
    soak("plane batch, 1 plane + table", [o16], lambda: M.roundtrip_i16_planes([(i16, o16, W, H, K1)]))
    if W == 8192:  # plane batches: three planes of different shapes (partial last tiles), own tables, kernel-argument and device-table forms
        shapes = [(7680, 4320), (3840, 2160), (3840, 2160)]
        bi = [synth.plane_i16_torch(w, h, "photo", seed=20 + k) for k, (w, h) in enumerate(shapes)]
        bo = [torch.empty_like(t) for t in bi]
        desc = [(a, b, w, h, l) for a, b, (w, h), l in zip(bi, bo, shapes, (K1, synth.JPEG_CHROMA, synth.JPEG_CHROMA))]
        for mode in ("roundtrip", "fwd", "inv"):
            soak(f"plane batch 4:2:0 frame, {mode}, kernel arguments", bo, lambda: M.i16_batch(mode, desc))
        dev_batch = M.Batch("roundtrip", desc)
        soak("plane batch 4:2:0 frame, roundtrip, device table", bo, lambda: dev_batch.run())
        del bi, bo, desc, dev_batch
    soak("u8 px -> i16 coef", [o16], lambda: M.fwd_u8_i16(u8, o16, W, H, lut=K1))
    p8 = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    soak("u8 -> u8 fused round trip + table (k_u8_batch, fast build)", [p8], lambda: M.roundtrip_u8(u8, p8, W, H, lut=K1))
    soak("u8 -> u8 fused round trip, wild table (general build)", [p8], lambda: M.roundtrip_u8(u8, p8, W, H, lut=np.full(64, 0.02, dtype=np.float32)))
    if W == 8192:
        shapes8 = [(7680, 4320), (3840, 2160), (3840, 2160)]
        ui = [synth.plane_u8_torch(w, h, "photo", seed=30 + k) for k, (w, h) in enumerate(shapes8)]
        uo = [torch.empty_like(t) for t in ui]
        udesc = [(a, b, w, h, l) for a, b, (w, h), l in zip(ui, uo, shapes8, (K1, synth.JPEG_CHROMA, synth.JPEG_CHROMA))]
        soak("8-bit 4:2:0 frame, kernel arguments", uo, lambda: M.roundtrip_u8_batch(udesc))
        ub = M.Batch("roundtrip_u8", udesc)
        soak("8-bit 4:2:0 frame, device table", uo, lambda: ub.run())
        uc = [torch.empty((h, w), dtype=torch.int16, device="cuda") for (w, h) in shapes8]
        hdesc = [(a, c, w, h, l) for a, c, (w, h), l in zip(ui, uc, shapes8, (K1, synth.JPEG_CHROMA, synth.JPEG_CHROMA))]
        soak("8-bit 4:2:0 frame -> int16 coefficients, one launch", uc, lambda: M.u8_i16_batch("fwd", hdesc))
        idesc = [(b, c, w, h, l) for b, c, (w, h), l in zip(uo, uc, shapes8, (K1, synth.JPEG_CHROMA, synth.JPEG_CHROMA))]
        soak("int16 coefficients -> 8-bit 4:2:0 frame, one launch", uo, lambda: M.u8_i16_batch("inv", idesc))
        qo = [torch.empty(w * h, dtype=torch.uint8, device="cuda") for (w, h) in shapes8]
        qdesc = [(a, o, w, h, lut2000) for a, o, (w, h) in zip(ui, qo, shapes8)]
        soak("8-bit 4:2:0 frame -> q32 product, one launch", qo, lambda: M.fwd_quant32_u8_batch(qdesc))
        del ui, uo, udesc, ub, uc, hdesc, idesc, qo, qdesc
    soak("i16 coef -> u8 px", [p8], lambda: M.inv_i16_u8(i16, p8, W, H, lut=K1))
    f32 = i16.float()
    of = torch.empty_like(f32)
    soak("f32 forward", [of.view(torch.int32)], lambda: M.fwd_f32(f32, of, W, H))
    soak("f32 inverse", [of.view(torch.int32)], lambda: M.inv_f32(f32, of, W, H))
    del f32, of
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    M.fwd_i16(i16, o16, W, H, lut=K1)
    coef = o16.clone()
    soak("zig-zag + run/level, i16", [lv, rn, ct], lambda: M.zigzag_rle_i16(coef, W, H, lv, rn, ct))
    soak("u8 px -> records, fused", [lv, rn, ct], lambda: M.fwd_u8_records(u8, W, H, lv, rn, ct, lut=K1))
    if W % 64 == 0:
        M.fwd_quant_u8(u8, o8, lut2000, W, H, 0, H // 8)
        q32b = o8.clone()
        soak("zig-zag + run/level, q32 bytes", [lv, rn, ct], lambda: M.zigzag_rle_q32(q32b, W, H, lv, rn, ct))
    stride = M.huffman_seg_stride(W)
    seg = torch.empty(((H // 8) * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
    ff = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
    for table, tn in ((K1, "Annex K.1"), (ones, "all-ones table (several ring windows)")):
        M.fwd_u8_records(u8, W, H, lv, rn, ct, lut=table)
        M.huffman_rows(lv, rn, ct, W, H, seg, nb)
        used = nb.clone()
        mask = (torch.arange(stride, device="cuda")[None, :] < used[:, None].long()).reshape(-1)  # only the bytes of the segments are defined
        segm = torch.empty_like(seg)
        def staged():
            M.huffman_rows(lv, rn, ct, W, H, seg, nb)
            torch.where(mask, seg, torch.zeros_like(seg), out=segm)
        soak("Huffman rows from records, " + tn, [segm, nb], staged)
        def fused():
            M.fwd_u8_huffman_rows(u8, W, H, seg, nb, lut=table, ff_counts=ff)
            torch.where(mask, seg, torch.zeros_like(seg), out=segm)
        soak("px -> Huffman rows, one kernel, " + tn, [segm, nb, ff], fused)
        M.fwd_u8_huffman_rows(u8, W, H, seg, nb, lut=table, ff_counts=ff)
        total = int(nb.sum().item()) + int(ff.sum().item()) + 2 * (H // 8 - 1)
        scan = torch.empty((total,), dtype=torch.uint8, device="cuda")
        off = torch.empty((H // 8 + 1,), dtype=torch.int64, device="cuda")
        nbc, ffc, segc = nb.clone(), ff.clone(), seg.clone()
        soak("scan packing, counted, " + tn, [scan, off], lambda: M.jpeg_pack_rows(segc, nbc, stride, H // 8, scan, off, ff_counts=ffc))
        soak("scan packing, uncounted, " + tn, [scan, off], lambda: M.jpeg_pack_rows(segc, nbc, stride, H // 8, scan, off))
        work = torch.zeros((H // 8 + 2,), dtype=torch.int64, device="cuda")
        soak("px -> finished scan, one launch, " + tn, [scan, off], lambda: M.fwd_u8_jpeg_scan(u8, W, H, seg, work, scan, off, lut=table))
        del mask, segm, scan, segc
    del seg, lv, rn, ct
    if W % 16 == 0 and H % 16 == 0:
        ycc = torch.stack([u8, u8.flip(0), u8.flip(1)], dim=-1).contiguous()
        y = torch.empty((H, W), dtype=torch.int16, device="cuda")
        cb = torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda")
        cr = torch.empty_like(cb)
        soak("4:2:0 split", [y, cb, cr], lambda: M.split420_u8(ycc, W, H, y, cb, cr))
        y8, cb8, cr8 = torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda"), torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda")
        soak("4:2:0 split into 8-bit planes", [y8, cb8, cr8], lambda: M.split420_u8_planes(ycc, W, H, y8, cb8, cr8))
        del y8, cb8, cr8
        del ycc, y, cb, cr
    torch.cuda.empty_cache()
print("soak ok" if not failed else f"!! {failed} kernels were not deterministic")
sys.exit(1 if failed else 0)
