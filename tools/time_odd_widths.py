"""Single planes whose rows are not whole 64-block tiles: the linear kernel (mdct_*_i16) against a plane batch of one (k_i16_batch, partial last tile).
   python3 tools/time_odd_widths.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
t = M.Timer()
def run(name, calls, px, reps=100):
    for i in range(400): calls[i % len(calls)]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps): calls[i % len(calls)]()
        t.stop(); r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{name:60s} {r[3]*1e3:8.2f} us  {4*px/(r[3]*1e-3)/8e12:6.3f} of 8 TB/s", flush=True)
for (W, H) in ((7688, 4320), (3848, 2160), (1928, 1088), (8200, 8192), (4104, 4096)):
    pl = [(synth.plane_i16_torch(W, H, "photo", seed=i),) for i in range(4)]
    pl = [(a, torch.empty_like(a)) for (a,) in pl]
    for mode in ("roundtrip", "fwd"):
        run(f"{W}x{H} {mode}: mdct_{mode}_i16 (linear kernel)", [M.prepare_plane_i16(mode, a, b, W, H) for a, b in pl], W * H)
        run(f"{W}x{H} {mode}: batch of one (tiles, partial last)", [M.prepare_i16_batch(mode, [(a, b, W, H, None)]) for a, b in pl], W * H)
    del pl
    torch.cuda.empty_cache()
