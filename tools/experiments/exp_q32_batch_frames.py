"""k_q32_batch: 1, 2, 4 frames (Y + Cb + Cr each) per launch -- the per-launch constant against the per-frame cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
t = M.Timer()
shapes = [(7680, 4320), (3840, 2160), (3840, 2160)]
ql = [(M.QUANTIZE_BASE * np.float32(s)).astype(np.float32) for s in (2000, 1200, 1200)]
NF = 8
frames = [[synth.plane_u8_torch(w, h, "photo", seed=10 * k + i) for k, (w, h) in enumerate(shapes)] for i in range(NF)]
outs = [[torch.empty(w * h, dtype=torch.uint8, device="cuda") for (w, h) in shapes] for _ in range(NF)]


def planes(i):
    return [(a, o, w, h, l) for a, o, (w, h), l in zip(frames[i], outs[i], shapes, ql)]


for per in (1, 2, 4):
    calls = [M.Batch("q32", [p for i in range(g * per, (g + 1) * per) for p in planes(i)]).prepared() for g in range(NF // per)]
    for i in range(200):
        calls[i % len(calls)]()
    r = []
    for k in range(11):
        t.start()
        for i in range(40):
            calls[i % len(calls)]()
        t.stop()
        r.append(t.elapsed_ms() / 40)
    r.sort()
    print(f"{per} frame(s) per launch: {r[5]*1e3:7.2f} us per launch = {r[5]*1e3/per:6.2f} per frame (min {r[0]*1e3/per:.2f})", flush=True)
# Y alone and a chroma plane alone through the same kernel
for name, idx in (("Y 7680x4320 alone", 0), ("Cb 3840x2160 alone", 1)):
    calls = [M.Batch("q32", [planes(i)[idx]]).prepared() for i in range(NF)]
    for i in range(200):
        calls[i % NF]()
    r = []
    for k in range(11):
        t.start()
        for i in range(40):
            calls[i % NF]()
        t.stop()
        r.append(t.elapsed_ms() / 40)
    r.sort()
    print(f"{name}: {r[5]*1e3:7.2f} us (min {r[0]*1e3:.2f})", flush=True)
