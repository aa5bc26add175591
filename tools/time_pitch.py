"""Does the row pitch matter?  8192^2 int16 round trip / forward with tight rows (16 KiB, a power of two) and padded ones.
   python3 tools/time_pitch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
W = H = 8192
t = M.Timer()
def run(name, calls, reps=200):
    for i in range(1200): calls[i % 4]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps): calls[i % 4]()
        t.stop(); r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{name:44s} median {r[3]*1e3:7.2f} us  min {r[0]*1e3:7.2f}  {4*W*H/(r[3]*1e-3)/8e12:6.3f} of 8 TB/s", flush=True)
for pad in (0, 64, 128, 512, 2048, 8):
    P = W + pad
    bufs = [(torch.zeros((H, P), dtype=torch.int16, device="cuda"), torch.zeros((H, P), dtype=torch.int16, device="cuda")) for _ in range(4)]
    for i, (a, b) in enumerate(bufs):
        a[:, :W] = synth.plane_i16_torch(W, H, "photo", seed=i)
    for mode in ("roundtrip", "fwd"):
        run(f"pitch {P} elements ({2*P} B) {mode}", [M.prepare_plane_i16(mode, a, b, W, H, pitch_in=P, pitch_out=P) for a, b in bufs])
    del bufs
    torch.cuda.empty_cache()
