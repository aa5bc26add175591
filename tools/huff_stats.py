import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
src = synth.plane_i16_torch(W, H, "photo", seed=synth.SEED)
lut60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
q = torch.empty_like(src)
M.fwd_i16(src, q, W, H, lut=lut60)
nblk = (W // 8) * (H // 8)
lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
M.zigzag_rle_i16(q, W, H, lv, rn, ct)
torch.cuda.synchronize()
c = ct.cpu().numpy().astype(int)
print("pairs/block mean", c.mean(), "max", c.max(), "p50/p90/p99", np.percentile(c, [50, 90, 99]))
g = c.reshape(-1, 64)
print("mean over waves of max n", g.max(axis=1).mean(), " >24:", (c > 24).mean(), "waves with any>24", (g.max(axis=1) > 24).mean())
s = np.sort(c.reshape(-1, 256), axis=1).reshape(-1, 4, 64)
print("sorted within 256: mean of wave max", s.max(axis=2).mean(axis=0), "sum", s.max(axis=2).sum(axis=1).mean(), "unsorted sum", c.reshape(-1,4,64).max(axis=2).sum(axis=1).mean())
