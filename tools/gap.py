"""How long an idle gap restarts the power-management transient?  precondition -> sync -> gap -> 100 timed launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
calls = [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H) for i in range(4)]
t = M.Timer()
for gap_ms in (0, 0, 0.2, 1, 5, 20, 100, 0):
    for i in range(1500): calls[i % 4]()
    torch.cuda.synchronize()
    if gap_ms: time.sleep(gap_ms / 1e3)
    t.start()
    for i in range(100): calls[i % 4]()
    t.stop()
    a = t.elapsed_ms() / 100 * 1e3
    t.start()
    for i in range(100): calls[i % 4]()
    t.stop()
    b = t.elapsed_ms() / 100 * 1e3
    print(f"gap {gap_ms:6.1f} ms: first 100 launches {a:5.1f} us, next 100 {b:5.1f} us")
