"""Soak of the one-launch encoder (mdct_fwd_u8_jpeg_scan) against the two-launch path: random sizes, tables and contents,
one shared row_work over all calls, several launches in flight back to back before each comparison.
    python3 tools/soak_jpeg_scan.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
M.init(0)
rng = np.random.default_rng(2026)
K1 = synth.JPEG_LUMA
work = torch.zeros((8192 + 2,), dtype=torch.int64, device="cuda")
t0 = time.time()
it = 0
while time.time() - t0 < secs:
    W = int(rng.choice([8, 64, 200, 512, 1000, 2048, 2048, 2048, 4096, 8192]))  # 2048 = one 256-block chunk per row: dense rows need several ring windows
    H = 8 * int(rng.integers(1, 1 + min(8192, (1 << 24) // W) // 8))
    kind = str(rng.choice(["photo", "noise", "flat" if W >= 64 else "photo"]))
    q = [K1, np.ones(64, dtype=np.float32), (K1 * np.float32(rng.uniform(0.1, 4))).astype(np.float32), None][int(rng.integers(0, 4))]
    img = synth.plane_u8_torch(W, H, "photo" if kind == "flat" else kind, seed=int(rng.integers(1 << 30)))
    if kind == "flat":
        img = (img // 64) * 64
    n, stride = H // 8, M.huffman_seg_stride(W)
    seg = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((n,), dtype=torch.int32, device="cuda")
    ff = torch.zeros((n,), dtype=torch.int32, device="cuda")
    M.fwd_u8_huffman_rows(img, W, H, seg, nb, lut=q, ff_counts=ff)
    total = int(nb.sum().item()) + int(ff.sum().item()) + 2 * (n - 1)
    want = torch.zeros((total + 8,), dtype=torch.uint8, device="cuda")
    woff = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    M.jpeg_pack_rows(seg, nb, stride, n, want, woff, ff_counts=ff)
    if it % 4 == 0 and W * H <= 1 << 22:  # the staged coder too (3 B/px of records: smaller planes only)
        nblk = (W // 8) * (H // 8)
        lv = torch.zeros((nblk, 64), dtype=torch.int16, device="cuda")
        rn = torch.zeros((nblk, 64), dtype=torch.uint8, device="cuda")
        ct = torch.zeros((nblk,), dtype=torch.uint8, device="cuda")
        M.fwd_u8_records(img, W, H, lv, rn, ct, lut=q)
        seg_s = torch.empty_like(seg)
        nb_s = torch.zeros_like(nb)
        M.huffman_rows(lv, rn, ct, W, H, seg_s, nb_s)
        want_s = torch.zeros_like(want)
        woff_s = torch.zeros_like(woff)
        M.jpeg_pack_rows(seg_s, nb_s, stride, n, want_s, woff_s)  # the uncounted packing
        if not (torch.equal(nb_s, nb) and torch.equal(woff_s, woff) and torch.equal(want_s, want)):
            print(f"!! MISMATCH at iteration {it}: {W}x{H} {kind}: staged path vs fused kernel + counted packing")
            sys.exit(1)
    gots = [(torch.zeros((total + 8,), dtype=torch.uint8, device="cuda"), torch.zeros((n + 1,), dtype=torch.int64, device="cuda")) for _ in range(3)]
    seg_w = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    for got, off in gots:  # three launches back to back on the same work array
        M.fwd_u8_jpeg_scan(img, W, H, seg_w, work, got, off, lut=q, out_capacity=total)
    torch.cuda.synchronize()
    for k, (got, off) in enumerate(gots):
        if not (torch.equal(off, woff) and torch.equal(got, want)):
            print(f"!! MISMATCH at iteration {it}: {W}x{H} {kind} launch {k}")
            o, wo = off.cpu().numpy(), woff.cpu().numpy()
            bad = np.nonzero(o != wo)[0]
            print("   offsets differing:", len(bad), "first", bad[:5], o[bad[:5]], wo[bad[:5]])
            g, w = got.cpu().numpy(), want.cpu().numpy()
            badb = np.nonzero(g != w)[0]
            print("   bytes differing:", len(badb), "first", badb[:8], "last", badb[-3:] if len(badb) else None)
            if len(badb):
                rows_hit = np.unique(np.searchsorted(wo, badb, side="right") - 1)
                print("   rows hit:", rows_hit[:20], "of", n, "| row lengths there:", (wo[rows_hit[:5] + 1] - wo[rows_hit[:5]]))
                r0 = int(rows_hit[0])
                nbh = nb.cpu().numpy()
                sa = seg.cpu().numpy()[r0 * stride:r0 * stride + nbh[r0]]
                sb = seg_w.cpu().numpy()[r0 * stride:r0 * stride + nbh[r0]]
                ds = np.nonzero(sa != sb)[0]
                print("   the row's unstuffed segment, one-launch scratch vs two-launch: differing bytes", len(ds), ds[:6], sa[ds[:6]], sb[ds[:6]], "of", nbh[r0])
                print("   position in the stuffed row:", badb - wo[r0], "mod 4/16/64:", (badb - wo[r0]) % 4, (badb[0] + 0) % 16, "global addr mod 16 unknown; scan offset", badb)
                b0 = badb[0]
                print("   got ", g[b0 - 4:b0 + 12], "\n   want", w[b0 - 4:b0 + 12])
            sys.exit(1)
    it += 1
    if it % 50 == 0:
        print(f"{it} cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {it} random cases x 3 launches in {time.time() - t0:.0f} s, epoch {int(work[0].item())}")
