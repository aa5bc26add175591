"""The fused 8-bit round trip (BASELINE.json configs[2] as SURVEY.md 8(d) states it: u8 in, u8 out, 2 B/px) against what it replaces.
   python3 tools/time_u8_roundtrip.py          (A/B of library builds: MDCT_LIB_PATH)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
jl, jc = synth.JPEG_LUMA, synth.JPEG_CHROMA
t = M.Timer()


def run(name, calls, px, bpp=2, reps=60):
    for i in range(max(3 * len(calls), 90)):
        calls[i % len(calls)]()
    r = []
    for k in range(9):
        t.start()
        for i in range(reps):
            calls[i % len(calls)]()
        t.stop()
        r.append(t.elapsed_ms() / reps)
    r.sort()
    ms = r[len(r) // 2]
    print(f"{name:72s} {ms*1e3:8.2f} us  {px/(ms*1e-3)/1e12:6.3f} Tpx/s  {bpp*px/(ms*1e-3)/1e12:6.3f} TB/s = {bpp*px/(ms*1e-3)/8e12:5.3f} of 8 TB/s  (min {r[0]*1e3:.2f})", flush=True)


def mk(w, h, s):
    a = synth.plane_u8_torch(w, h, "photo", seed=s)
    return a, torch.empty_like(a)


NF = 6  # 99.5 MB per frame: rotate well past the 256 MB MALL
fpx = 7680 * 4320 + 2 * 3840 * 2160
frames = []
for i in range(NF):
    y, cb, cr = mk(7680, 4320, i), mk(3840, 2160, 10 + i), mk(3840, 2160, 20 + i)
    frames.append([(y[0], y[1], 7680, 4320, jl), (cb[0], cb[1], 3840, 2160, jc), (cr[0], cr[1], 3840, 2160, jc)])
run("8K 4:2:0 frame u8 -> u8, mdct_roundtrip_u8_batch (kernel arguments)", [M.prepare_u8_batch(f) for f in frames], fpx)
bs = [M.Batch("roundtrip_u8", f) for f in frames]
run("8K 4:2:0 frame u8 -> u8, mdct_batch_run (device table)", [b.prepared() for b in bs], fpx)
big = [mk(8192, 8192, 40 + i) for i in range(6)]
run("8192^2 u8 -> u8 round trip + table", [M.prepare_roundtrip_u8(a, b, 8192, 8192, lut=jl) for a, b in big], 8192 * 8192)
if "--quick" in sys.argv:
    sys.exit(0)
run("8K 4:2:0 frame u8 -> u8, no level shift", [M.prepare_u8_batch(f, level_shift=False) for f in frames], fpx)
wild = np.full(64, 0.02, dtype=np.float32)
run("8K 4:2:0 frame u8 -> u8, wild table (general build)", [M.prepare_u8_batch([(a, b, w, h, wild) for a, b, w, h, l in f]) for f in frames], fpx)
run("Y 7680x4320 alone", [M.prepare_roundtrip_u8(f[0][0], f[0][1], 7680, 4320, lut=jl) for f in frames], 7680 * 4320)
run("Cb 3840x2160 alone", [M.prepare_roundtrip_u8(f[1][0], f[1][1], 3840, 2160, lut=jc) for f in frames], 3840 * 2160)
coefs = [torch.empty((4320, 7680), dtype=torch.int16, device="cuda") for _ in range(2)]
run("Y alone: mdct_fwd_u8_i16 (the first of the two calls it fuses)", [M.prepare_u8_i16("fwd", f[0][0], coefs[i % 2], 7680, 4320, lut=jl) for i, f in enumerate(frames)], 7680 * 4320, bpp=3)
run("Y alone: mdct_inv_i16_u8 (the second)", [M.prepare_u8_i16("inv", coefs[i % 2], f[0][1], 7680, 4320, lut=jl) for i, f in enumerate(frames)], 7680 * 4320, bpp=3)
del coefs
fc = [[torch.empty((h, w), dtype=torch.int16, device="cuda") for (_, _, w, h, _) in f] for f in frames]
fwd = [M.Batch("fwd_u8_i16", [(a, c, w, h, l) for (a, b, w, h, l), c in zip(f, cs)]) for f, cs in zip(frames, fc)]
run("8K 4:2:0 frame u8 -> int16 coefficients, one launch (3 B/px)", [b.prepared() for b in fwd], fpx, bpp=3)
inv = [M.Batch("inv_i16_u8", [(b, c, w, h, l) for (a, b, w, h, l), c in zip(f, cs)]) for f, cs in zip(frames, fc)]
run("8K 4:2:0 frame int16 coefficients -> u8, one launch (3 B/px)", [b.prepared() for b in inv], fpx, bpp=3)
del fc, fwd, inv
ql = [(M.QUANTIZE_BASE * np.float32(s)).astype(np.float32) for s in (2000, 1200, 1200)]
qo = [[torch.empty(w * h, dtype=torch.uint8, device="cuda") for (_, _, w, h, _) in f] for f in frames]
qb = [M.Batch("q32", [(a, o, w, h, l) for (a, b, w, h, _), o, l in zip(f, os_, ql)]) for f, os_ in zip(frames, qo)]
run("8K 4:2:0 frame -> the reference's q32 product, one launch (2 B/px)", [b.prepared() for b in qb], fpx)
run("   ... kernel-argument form", [M.prepare_fwd_quant32_u8_batch([(a, o, w, h, l) for (a, b, w, h, _), o, l in zip(f, os_, ql)]) for f, os_ in zip(frames, qo)], fpx)
q3 = [[M.prepare_fwd_quant_u8(a, o, l, w, h, 0, h // 8) for (a, b, w, h, _), o, l in zip(f, os_, ql)] for f, os_ in zip(frames, qo)]
run("   ... as the three single-plane calls of the reference's caller", [(lambda cs: (lambda: [c() for c in cs]))(cs) for cs in q3], fpx)
del qo, qb, q3
run("8192^2 u8 stream copy of the same bytes", [M.prepare_stream_copy(a, b, 8192 * 8192) for a, b in big], 8192 * 8192)
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
outs = [torch.empty(8192 * 8192, dtype=torch.uint8, device="cuda") for _ in range(2)]
run("8192^2 q32 (the reference's tier, 2 B/px) for comparison", [M.prepare_fwd_quant_u8(a, outs[i % 2], lut, 8192, 8192, 0, 1024) for i, (a, b) in enumerate(big)], 8192 * 8192)
del big, outs
# the int16 form of the same frame (what round 4 measured as configs[2]: 4 B/px)
def mk16(w, h, s):
    a = synth.plane_i16_torch(w, h, "photo", seed=s)
    return a, torch.empty_like(a)
f16 = []
for i in range(4):
    y, cb, cr = mk16(7680, 4320, i), mk16(3840, 2160, 10 + i), mk16(3840, 2160, 20 + i)
    f16.append([(y[0], y[1], 7680, 4320, jl), (cb[0], cb[1], 3840, 2160, jc), (cr[0], cr[1], 3840, 2160, jc)])
run("8K 4:2:0 frame int16 -> int16 (round 4's form, 4 B/px)", [M.prepare_i16_batch("roundtrip", f) for f in f16], fpx, bpp=4)
big16 = [mk16(8192, 8192, 40 + i) for i in range(4)]
run("8192^2 int16 round trip (bench workload)", [M.prepare_plane_i16("roundtrip", a, b, 8192, 8192) for a, b in big16], 8192 * 8192, bpp=4)
run("8192^2 int16 round trip + table", [M.prepare_plane_i16("roundtrip", a, b, 8192, 8192, lut=jl) for a, b in big16], 8192 * 8192, bpp=4)
