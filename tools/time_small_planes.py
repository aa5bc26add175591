"""Single-plane launches that fill the chip only a few times over: how long do they take, and does more occupancy (fewer generations of waves) help?
   MDCT_LIB_PATH=build_variants/lib_tw4.so python3 tools/time_small_planes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
t = M.Timer()
tag = os.path.basename(os.environ.get("MDCT_LIB_PATH", "product"))
def run(name, calls, px, reps=200):
    for i in range(600): calls[i % len(calls)]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps): calls[i % len(calls)]()
        t.stop(); r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{tag:14s} {name:40s} {r[3]*1e3:8.2f} us  {4*px/(r[3]*1e-3)/8e12:6.3f} of 8 TB/s", flush=True)
for (W, H) in ((2048, 2048), (4096, 4096), (7680, 4320), (8192, 8192)):
    n = 8 if W * H <= 4096 * 4096 else 4
    pl = [synth.plane_i16_torch(W, H, "photo", seed=i) for i in range(n)]
    pl = [(a, torch.empty_like(a)) for a in pl]
    for mode in ("fwd", "roundtrip"):
        run(f"{W}x{H} {mode}", [M.prepare_plane_i16(mode, a, b, W, H) for a, b in pl], W * H)
    del pl; torch.cuda.empty_cache()
