import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
K1 = synth.JPEG_LUMA
for (W, H) in ((8, 8 * 1024), (8, 8 * 16384), (8, 8 * 65535), (64, 8 * 1024), (64, 8 * 30000), (512, 8 * 1024), (512, 8 * 16000)):
    n = H // 8
    img = synth.plane_u8_torch(W, H, "photo", seed=3)
    stride = M.huffman_seg_stride(W)
    seg = torch.empty((n * stride,), dtype=torch.uint8, device="cuda")
    nb = torch.zeros((n,), dtype=torch.int32, device="cuda")
    ff = torch.zeros((n,), dtype=torch.int32, device="cuda")
    M.fwd_u8_huffman_rows(img, W, H, seg, nb, lut=K1, ff_counts=ff)
    total = int(nb.sum().item()) + int(ff.sum().item()) + 2 * (n - 1)
    want = torch.zeros((total,), dtype=torch.uint8, device="cuda")
    woff = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    M.jpeg_pack_rows(seg, nb, stride, n, want, woff, ff_counts=ff)
    work = torch.zeros((n + 2,), dtype=torch.int64, device="cuda")
    got = torch.zeros((total,), dtype=torch.uint8, device="cuda")
    off = torch.zeros((n + 1,), dtype=torch.int64, device="cuda")
    dt = 1e9
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        M.fwd_u8_jpeg_scan(img, W, H, seg, work, got, off, lut=K1)
        torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    M.fwd_u8_huffman_rows(img, W, H, seg, nb, lut=K1, ff_counts=ff)
    M.jpeg_pack_rows(seg, nb, stride, n, want, woff, ff_counts=ff)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"{W}x{H}: {n} rows, one call {dt*1e6:.0f} us = {dt*1e9/n:.1f} ns per row, two calls {dt2*1e6:.0f} us, equal: {torch.equal(got, want) and torch.equal(off, woff)}", flush=True)
