import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
u8 = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
st = M.huffman_seg_stride(W)
seg = torch.empty(((H // 8) * st,), dtype=torch.uint8, device="cuda")
work = torch.zeros((H // 8 + 2,), dtype=torch.int64, device="cuda")
scan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
off = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
t = M.Timer()
f = lambda i: M.fwd_u8_jpeg_scan(u8[i % 4], W, H, seg, work, scan, off, lut=K1)
for i in range(200):
    f(i)
best = []
for r in range(5):
    t.start()
    for i in range(40):
        f(i)
    t.stop()
    best.append(t.elapsed_ms() / 40)
best.sort()
print(f"one-launch px -> scan, K.1: {best[2] * 1e3:.1f} us, {int(off[-1].item())} bytes")
