"""One plane through k_u8_i16 (mdct_fwd_u8_i16 / mdct_inv_i16_u8) against the same plane as a batch of one through k_u8_batch<FWD|INV>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
t = M.Timer()


def run(name, calls, reps=60):
    for i in range(120):
        calls[i % len(calls)]()
    r = []
    for k in range(9):
        t.start()
        for i in range(reps):
            calls[i % len(calls)]()
        t.stop()
        r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{name:70s} {r[4]*1e3:8.2f} us (min {r[0]*1e3:.2f})", flush=True)


for (w, h) in ((7680, 4320), (8192, 8192), (3840, 2160), (1920, 1080 // 8 * 8)):
    n = 6
    px = [synth.plane_u8_torch(w, h, "photo", seed=i) for i in range(n)]
    out = [torch.empty_like(p) for p in px]
    co = [torch.empty((h, w), dtype=torch.int16, device="cuda") for _ in range(n)]
    L = synth.JPEG_LUMA
    run(f"{w}x{h} fwd single call", [M.prepare_u8_i16("fwd", px[i], co[i], w, h, lut=L) for i in range(n)])
    run(f"{w}x{h} fwd batch of one (kernel arguments)", [M.prepare_u8_i16_batch("fwd", [(px[i], co[i], w, h, L)]) for i in range(n)])
    bs = [M.Batch("fwd_u8_i16", [(px[i], co[i], w, h, L)]) for i in range(n)]
    run(f"{w}x{h} fwd batch of one (device table)", [b.prepared() for b in bs])
    run(f"{w}x{h} inv single call", [M.prepare_u8_i16("inv", co[i], out[i], w, h, lut=L) for i in range(n)])
    run(f"{w}x{h} inv batch of one (kernel arguments)", [M.prepare_u8_i16_batch("inv", [(out[i], co[i], w, h, L)]) for i in range(n)])
    del px, out, co, bs
