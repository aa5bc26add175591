"""Plane batches against what they replace (BASELINE.json configs[2] and configs[3] on one GPU).
   python3 tools/time_batch.py [--planes 256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
NPL = int(sys.argv[sys.argv.index("--planes") + 1]) if "--planes" in sys.argv else 256
jl, jc = synth.JPEG_LUMA, synth.JPEG_CHROMA
t = M.Timer()


def run(name, calls, px, reps=40, bpp=4):
    for i in range(max(3 * len(calls), 60 if reps >= 40 else 3)):
        calls[i % len(calls)]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps):
            calls[i % len(calls)]()
        t.stop()
        r.append(t.elapsed_ms() / reps)
    r.sort()
    ms = r[len(r) // 2]
    print(f"{name:64s} {ms*1e3:9.2f} us  {bpp*px/(ms*1e-3)/1e12:6.3f} TB/s  {bpp*px/(ms*1e-3)/8e12:6.3f} of 8 TB/s   (min {r[0]*1e3:.2f})", flush=True)


def mk(w, h, s):
    a = synth.plane_i16_torch(w, h, "photo", seed=s)
    return a, torch.empty_like(a)


# ---- configs[2]: 8K 4:2:0 frame, fused round trip with per-plane tables; 4 frames rotated (199 MB each: the MALL holds 256 MB)
NF = 4
Ys = [mk(7680, 4320, i) for i in range(NF)]
Cbs = [mk(3840, 2160, 10 + i) for i in range(NF)]
Crs = [mk(3840, 2160, 20 + i) for i in range(NF)]
fpx = 7680 * 4320 + 2 * 3840 * 2160
frames = [[(Ys[i][0], Ys[i][1], 7680, 4320, jl), (Cbs[i][0], Cbs[i][1], 3840, 2160, jc), (Crs[i][0], Crs[i][1], 3840, 2160, jc)] for i in range(NF)]
run("4:2:0 frame  mdct_roundtrip_i16_planes", [M.prepare_roundtrip_i16_planes(f) for f in frames], fpx)
run("4:2:0 frame  mdct_roundtrip_i16_batch (kernel arguments)", [M.prepare_i16_batch("roundtrip", f) for f in frames], fpx)
bs = [M.Batch("roundtrip", f) for f in frames]
run("4:2:0 frame  mdct_batch_run (device table)", [b.prepared() for b in bs], fpx)
run("4:2:0 frame  fwd batch", [M.prepare_i16_batch("fwd", f) for f in frames], fpx)
run("4:2:0 frame  inv batch", [M.prepare_i16_batch("inv", f) for f in frames], fpx)
nolut = [[(a, b, w, h, None) for (a, b, w, h, l) in f] for f in frames]
run("4:2:0 frame  roundtrip batch, no tables", [M.prepare_i16_batch("roundtrip", f) for f in nolut], fpx)
run("Y 7680x4320 alone, mdct_roundtrip_i16 (+table)", [M.prepare_plane_i16("roundtrip", Ys[i][0], Ys[i][1], 7680, 4320, lut=jl) for i in range(NF)], 7680 * 4320)
run("Cb 3840x2160 alone, mdct_roundtrip_i16 (+table)", [M.prepare_plane_i16("roundtrip", Cbs[i][0], Cbs[i][1], 3840, 2160, lut=jc) for i in range(NF)], 3840 * 2160)
run("Cb 3840x2160 alone, batch of one", [M.prepare_i16_batch("roundtrip", [f[1]]) for f in frames], 3840 * 2160)
big = [mk(8192, 8192, 40 + i) for i in range(4)]
run("8192^2 mdct_roundtrip_i16 (k_i16_tile)", [M.prepare_plane_i16("roundtrip", a, b, 8192, 8192) for a, b in big], 8192 * 8192)
run("8192^2 roundtrip, batch of one", [M.prepare_i16_batch("roundtrip", [(a, b, 8192, 8192, None)]) for a, b in big], 8192 * 8192)
run("8192^2 mdct_roundtrip_i16 + table (tables in the kernel arguments)", [M.prepare_plane_i16("roundtrip", a, b, 8192, 8192, lut=jl) for a, b in big], 8192 * 8192)
run("8192^2 roundtrip + table, batch of one, kernel arguments", [M.prepare_i16_batch("roundtrip", [(a, b, 8192, 8192, jl)]) for a, b in big], 8192 * 8192)
b1 = [M.Batch("roundtrip", [(a, b, 8192, 8192, jl)]) for a, b in big]
run("8192^2 roundtrip + table, batch of one, device table", [x.prepared() for x in b1], 8192 * 8192)
b2 = [M.Batch("fwd", [(a, b, 8192, 8192, jl)]) for a, b in big]
run("8192^2 fwd + table, batch of one, device table", [x.prepared() for x in b2], 8192 * 8192)
run("8192^2 mdct_fwd_i16 + table (k_i16_tile)", [M.prepare_plane_i16("fwd", a, b, 8192, 8192, lut=jl) for a, b in big], 8192 * 8192)
del b1, b2
run("8192^2 mdct_fwd_i16 (k_i16_tile)", [M.prepare_plane_i16("fwd", a, b, 8192, 8192) for a, b in big], 8192 * 8192)
run("8192^2 fwd, batch of one", [M.prepare_i16_batch("fwd", [(a, b, 8192, 8192, None)]) for a, b in big], 8192 * 8192)
run("8192^2 stream copy", [M.prepare_stream_copy(a, b, 8192 * 8192 * 2) for a, b in big], 8192 * 8192)
del Ys, Cbs, Crs, frames, bs, nolut, big
torch.cuda.empty_cache()
if "--frame-only" in sys.argv:
    sys.exit(0)

# ---- configs[3] on one GPU: NPL independent 4096^2 planes, forward only
W = H = 4096
pl = [mk(W, H, 100 + p) for p in range(NPL)]
desc = [(a, b, W, H, None) for a, b in pl]
px = NPL * W * H
b = M.Batch("fwd", desc)
run(f"{NPL} x 4096^2 fwd, separately allocated: mdct_batch_run ({b.launches} launch)", [b.prepared()], px, reps=3)
run(f"{NPL} x 4096^2 fwd, separately allocated: mdct_fwd_i16_batch (kernel arguments)", [M.prepare_i16_batch("fwd", desc)], px, reps=3)
per = [M.prepare_plane_i16("fwd", a, o, W, H) for a, o in pl]
run(f"{NPL} x 4096^2 fwd, one mdct_fwd_i16 per plane", [lambda: [c() for c in per]], px, reps=3)
del pl, desc, b, per
torch.cuda.empty_cache()
tall_in = torch.empty((NPL * H, W), dtype=torch.int16, device="cuda")
for p in range(NPL):
    tall_in[p * H:(p + 1) * H] = synth.plane_i16_torch(W, H, "photo", seed=100 + p)
tall_out = torch.empty_like(tall_in)
run(f"{NPL} x 4096^2 fwd, stacked = one tall plane, one mdct_fwd_i16", [M.prepare_plane_i16("fwd", tall_in, tall_out, W, NPL * H)], px, reps=3)
run(f"{NPL} x 4096^2 stream copy of the same bytes", [M.prepare_stream_copy(tall_in, tall_out, NPL * H * W * 2)], px, reps=3)
