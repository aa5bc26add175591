"""A/B: time the q32 u8 kernel (and friends) for whatever libmdct_hip.so is installed.
   python3 tools/time_q32.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
u8s = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
u8d = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
calls = {
    "q32": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 8) for i in range(4)],
    "stereo_sse": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 16, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SSE) for i in range(4)],
    "stereo_scalar": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 16, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SCALAR) for i in range(4)],
    "encq_scalar": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 8, layout=M.LAYOUT_BLOCK, profile=M.PROFILE_REF_SCALAR) for i in range(4)],
    "encq_sse": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 8, layout=M.LAYOUT_BLOCK_SSE, profile=M.PROFILE_REF_SSE) for i in range(4)],
}
t = M.Timer()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
res = {k: [] for k in calls}
for r in range(rounds + 1):
    for k, c in calls.items():
        t.start()
        for i in range(20):
            c[i % 4]()
        t.stop()
        ms = t.elapsed_ms() / 20
        if r:
            res[k].append(ms)
for k, v in res.items():
    v.sort()
    print(f"{k:12s} median {v[len(v)//2]*1e3:8.2f} us  min {v[0]*1e3:8.2f} us   {2*W*H/(v[len(v)//2]*1e-3)/1e9:7.1f} GB/s")
