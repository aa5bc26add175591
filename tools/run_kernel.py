"""Launch ONE engine kernel repeatedly on the bench workload (for rocprofv3 --pmc passes):
    python3 tools/run_kernel.py {roundtrip|fwd|inv|q32|copy} [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import simd_dct_amd as M
from simd_dct_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "roundtrip"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W = H = 8192
M.init(0)
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
u8s = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
u8d = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
for i in range(n):
    s = i % 4
    if which == "roundtrip":
        M.roundtrip_i16(srcs[s], dsts[s], W, H)
    elif which == "fwd":
        M.fwd_i16(srcs[s], dsts[s], W, H)
    elif which == "inv":
        M.inv_i16(srcs[s], dsts[s], W, H)
    elif which == "q32":
        M.fwd_quant_u8(u8s[s], u8d[s], lut, W, H, 0, H // 8)
    elif which == "copy":
        M.stream_copy(srcs[s], dsts[s], W * H * 2)
torch.cuda.synchronize()
print("done", which, n)
