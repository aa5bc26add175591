"""k_q32_batch (the reference's q32 product on the 8K 4:2:0 frame, one launch) at different waves per SIMD: builds with -DMDCT_Q32B_WAVES=n
selected through MDCT_LIB_PATH.   for n in 3 4 5 8: MDCT_LIB_PATH=build_variants/libmdct_q32b$n.so python3 tools/experiments/exp_q32_batch_waves.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
t = M.Timer()
NF = 6
shapes = [(7680, 4320), (3840, 2160), (3840, 2160)]
ql = [(M.QUANTIZE_BASE * np.float32(s)).astype(np.float32) for s in (2000, 1200, 1200)]
frames = [[synth.plane_u8_torch(w, h, "photo", seed=10 * k + i) for k, (w, h) in enumerate(shapes)] for i in range(NF)]
outs = [[torch.empty(w * h, dtype=torch.uint8, device="cuda") for (w, h) in shapes] for _ in range(NF)]
calls = [M.Batch("q32", [(a, o, w, h, l) for a, o, (w, h), l in zip(f, os_, shapes, ql)]).prepared() for f, os_ in zip(frames, outs)]
for i in range(200):
    calls[i % NF]()
r = []
for k in range(11):
    t.start()
    for i in range(60):
        calls[i % NF]()
    t.stop()
    r.append(t.elapsed_ms() / 60)
r.sort()
print(f"{os.environ.get('MDCT_LIB_PATH', 'product build'):44s} {r[5]*1e3:7.2f} us (min {r[0]*1e3:.2f})", flush=True)
