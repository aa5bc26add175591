"""Does a kernel slow down under sustained back-to-back launches (clock/power management)?
   python3 tools/sustained.py  -> per-20-launch averages over 600 launches for copy / fwd / roundtrip / q32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
W = H = 8192
M.init(0)
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
u8s = [s.view(torch.uint8).reshape(-1)[: W * H] for s in srcs]
u8d = [d.view(torch.uint8).reshape(-1)[: W * H] for d in dsts]
kinds = {
    "copy": [M.prepare_stream_copy(srcs[i], dsts[i], W * H * 2) for i in range(4)],
    "fwd_i16": [M.prepare_plane_i16("fwd", srcs[i], dsts[i], W, H) for i in range(4)],
    "roundtrip": [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H) for i in range(4)],
    "q32": [M.prepare_fwd_quant_u8(u8s[i], u8d[i], lut, W, H, 0, H // 8) for i in range(4)],
}
N, CH = 600, 20
for name, calls in kinds.items():
    torch.cuda.synchronize()
    import time; time.sleep(0.5)
    timers = [M.Timer() for _ in range(N // CH)]
    for c in range(N // CH):
        timers[c].start()
        for i in range(CH):
            calls[i % 4]()
        timers[c].stop()
    us = [t.elapsed_ms() / CH * 1e3 for t in timers]
    print(f"{name:10s} " + " ".join(f"{u:5.1f}" for u in us))
