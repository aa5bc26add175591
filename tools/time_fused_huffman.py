import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
Q60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
u8 = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
st = M.huffman_seg_stride(W)
seg = [torch.empty(((H // 8) * st,), dtype=torch.uint8, device="cuda") for _ in range(2)]
nb = [torch.empty((H // 8,), dtype=torch.int32, device="cuda") for _ in range(2)]
ff = [torch.empty((H // 8,), dtype=torch.int32, device="cuda") for _ in range(2)]
t = M.Timer()
for name, q in (("q60", Q60), ("K.1", K1)):
    for i in range(200):
        M.fwd_u8_huffman_rows(u8[i % 4], W, H, seg[i % 2], nb[i % 2], lut=q, ff_counts=ff[i % 2])
    best = []
    for r in range(5):
        t.start()
        for i in range(40):
            M.fwd_u8_huffman_rows(u8[i % 4], W, H, seg[i % 2], nb[i % 2], lut=q, ff_counts=ff[i % 2])
        t.stop()
        best.append(t.elapsed_ms() / 40)
    best.sort()
    print(f"fused px -> Huffman rows, {name}: {best[2] * 1e3:.1f} us")
# the packing of the K.1 rows just written: counted (one launch unless MDCT_PACK_SCAN_KERNEL is set) and uncounted (three launches)
scan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
off = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
for name, kw in (("counted", True), ("uncounted", False)):
    f = (lambda i: M.jpeg_pack_rows(seg[i % 2], nb[i % 2], st, H // 8, scan, off, ff_counts=ff[i % 2])) if kw else (lambda i: M.jpeg_pack_rows(seg[i % 2], nb[i % 2], st, H // 8, scan, off))
    for i in range(100):
        f(i)
    best = []
    for r in range(5):
        t.start()
        for i in range(40):
            f(i)
        t.stop()
        best.append(t.elapsed_ms() / 40)
    best.sort()
    print(f"pack {name} (scan kernel forced: {os.environ.get('MDCT_PACK_SCAN_KERNEL') is not None}): {best[2] * 1e3:.1f} us, {int(off[-1].item())} bytes")
