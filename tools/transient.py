"""From an idle chip: per-100-launch average of the round-trip kernel over 8000 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
calls = [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H) for i in range(4)]
torch.cuda.synchronize()
import time; time.sleep(2.0)
N, CH = 8000, 100
timers = [M.Timer() for _ in range(N // CH)]
for c in range(N // CH):
    timers[c].start()
    for i in range(CH):
        calls[i % 4]()
    timers[c].stop()
us = [t.elapsed_ms() / CH * 1e3 for t in timers]
for i in range(0, len(us), 10):
    print(f"launches {i*CH:5d}+: " + " ".join(f"{u:5.1f}" for u in us[i:i+10]))
