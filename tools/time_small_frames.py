"""1080p and 4K 4:2:0 frames through the plane-batch call: launches of a few thousand tiles (A/B of library builds with MDCT_LIB_PATH).
   python3 tools/time_small_frames.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
t = M.Timer()
def run(name, calls, px, reps=200):
    for i in range(600): calls[i % len(calls)]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps): calls[i % len(calls)]()
        t.stop(); r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{name:60s} {r[3]*1e3:8.2f} us  {4*px/(r[3]*1e-3)/8e12:6.3f} of 8 TB/s", flush=True)
def mk(w, h, s):
    a = synth.plane_i16_torch(w, h, "photo", seed=s); return a, torch.empty_like(a)
for (YW, YH, name) in ((1920, 1088, "1080p 4:2:0 frame"), (3840, 2160, "4K 4:2:0 frame")):
    fr = []
    for f in range(8):
        y = mk(YW, YH, f); cb = mk(YW // 2, YH // 2, 20 + f); cr = mk(YW // 2, YH // 2, 40 + f)
        fr.append([(y[0], y[1], YW, YH, synth.JPEG_LUMA), (cb[0], cb[1], YW // 2, YH // 2, synth.JPEG_CHROMA), (cr[0], cr[1], YW // 2, YH // 2, synth.JPEG_CHROMA)])
    px = YW * YH * 3 // 2
    run(name + " roundtrip, one call", [M.prepare_i16_batch("roundtrip", f) for f in fr], px)
    run(name + " fwd, one call", [M.prepare_i16_batch("fwd", f) for f in fr], px)
