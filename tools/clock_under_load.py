"""The shader clock the chip holds under each of the engine's VALU-heavy kernels (mdct_clock_probe on a second stream beside back-to-back
launches): what turns an instruction count and issue cycles into bench.py's valu_floor_ms.   python3 tools/clock_under_load.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
W = H = 8192
jl, jc = synth.JPEG_LUMA, synth.JPEG_CHROMA
side = torch.cuda.Stream()
probe = torch.zeros(16, dtype=torch.int64, device="cuda")


def clock(name, calls, us_per_launch):
    for i in range(400):  # settle the power state on this load
        calls[i % len(calls)]()
    torch.cuda.synchronize()
    n = 600
    for i in range(60):
        calls[i % len(calls)]()
    M.clock_probe(probe, int(us_per_launch * (n - 200) * 100 * 0.5), waves=8, stream=side)  # half of the remaining queue, in 10 ns ticks
    for i in range(60, n):
        calls[i % len(calls)]()
    torch.cuda.synchronize()
    p = probe.cpu().numpy().reshape(8, 2)
    ghz = (p[:, 0] / (p[:, 1] * 10.0))
    print(f"{name:44s} {ghz.mean():.3f} GHz  (per probe wave {ghz.min():.3f} .. {ghz.max():.3f})", flush=True)


u8 = [synth.plane_u8_torch(W, H, "photo", seed=i) for i in range(4)]
u8o = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
lut2000 = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
clock("k_q32_tile (8192^2)", [M.prepare_fwd_quant_u8(u8[i], u8o[i], lut2000, W, H, 0, H // 8) for i in range(4)], 27)
for nm, lay, prof, rows in (("stereo / SSE", M.LAYOUT_STEREO, M.PROFILE_REF_SSE, H // 16), ("stereo / scalar", M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, H // 16),
                            ("encq / SSE", M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, H // 8), ("encq / scalar", M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, H // 8)):
    clock("k_fwd_quant_u8 " + nm, [M.prepare_fwd_quant_u8(u8[i], u8o[i], lut8, W, H, 0, rows, layout=lay, profile=prof) for i in range(4)], 30)
frames = []
for f in range(6):
    pl = []
    for (w, h, so, tab) in synth.CONFIG3_PLANES:
        a = synth.plane_u8_torch(w, h, "photo", seed=synth.SEED + so + 10 * f)
        pl.append((a, torch.empty_like(a), w, h, jl if tab == "luma" else jc))
    frames.append(pl)
bs = [M.Batch("roundtrip_u8", f) for f in frames]
clock("k_u8_batch (8K 4:2:0 frame, u8 -> u8)", [b.prepared() for b in bs], 26)
i16 = [synth.plane_i16_torch(W, H, "photo", seed=i) for i in range(4)]
i16o = [torch.empty_like(t) for t in i16]
clock("k_i16_tile<ROUNDTRIP> (bench workload)", [M.prepare_plane_i16("roundtrip", i16[i], i16o[i], W, H) for i in range(4)], 44)
clock("k_stream_copy (8192^2 int16)", [M.prepare_stream_copy(i16[i], i16o[i], W * H * 2) for i in range(4)], 43)
