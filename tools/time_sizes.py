"""Fixed (head/tail) vs per-pixel cost: time the main kernels on 4096^2, 8192^2, 16384^2 planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
t = M.Timer()
for W in (2048, 4096, 8192, 16384, 32768):
    H = W
    n = 4 if W <= 8192 else 2
    i16 = [synth.plane_i16_torch(W, H, "photo", seed=i) for i in range(n)]
    o16 = [torch.empty_like(s) for s in i16]
    u8 = [(s & 0xFF).to(torch.uint8).reshape(-1) for s in i16]
    o8 = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(n)]
    cases = {
        "copy": ([M.prepare_stream_copy(i16[i], o16[i], W * H * 2) for i in range(n)], 4),
        "roundtrip": ([M.prepare_plane_i16("roundtrip", i16[i], o16[i], W, H) for i in range(n)], 4),
        "fwd": ([M.prepare_plane_i16("fwd", i16[i], o16[i], W, H) for i in range(n)], 4),
        "q32": ([M.prepare_fwd_quant_u8(u8[i], o8[i], lut, W, H, 0, H // 8) for i in range(n)], 2),
    }
    for name, (calls, bpp) in cases.items():
        for i in range(300): calls[i % n]()
        r = []
        for k in range(5):
            t.start()
            for i in range(30): calls[i % n]()
            t.stop(); r.append(t.elapsed_ms() / 30)
        r.sort(); ms = r[2]
        print(f"{W:6d}^2 {name:10s} {ms*1e3:9.2f} us   {ms*1e6/(W*H/1e6):8.3f} ns/Mpx... {bpp*W*H/(ms*1e-3)/1e9:8.1f} GB/s")
    del i16, o16, u8, o8
    torch.cuda.empty_cache()
