"""A/B in ONE process (boxes differ by more than the effect): the 8K 4:2:0 frame through k_u8_batch and k_q32_batch with the chroma planes'
rows tiled in pairs (12,150 waves, round 6) against one tile grid per row (12,420 waves, every 8th chroma wave half empty).
    python3 tools/experiments/exp_paired_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
t = M.Timer()
NF = 6
fpx = 7680 * 4320 + 2 * 3840 * 2160
jl, jc = synth.JPEG_LUMA, synth.JPEG_CHROMA


def mk(w, h, s):
    a = synth.plane_u8_torch(w, h, "photo", seed=s)
    return a, torch.empty_like(a)


frames = []
for i in range(NF):
    y, cb, cr = mk(7680, 4320, i), mk(3840, 2160, 10 + i), mk(3840, 2160, 20 + i)
    frames.append([(y[0], y[1], 7680, 4320, jl), (cb[0], cb[1], 3840, 2160, jc), (cr[0], cr[1], 3840, 2160, jc)])
ql = [(M.QUANTIZE_BASE * np.float32(s)).astype(np.float32) for s in (2000, 1200, 1200)]
qo = [[torch.empty(w * h, dtype=torch.uint8, device="cuda") for (_, _, w, h, _) in f] for f in frames]


def mk16(w, h, s):
    a = synth.plane_i16_torch(w, h, "photo", seed=s)
    return a, torch.empty_like(a)


f16 = []
for i in range(4):
    y, cb, cr = mk16(7680, 4320, i), mk16(3840, 2160, 10 + i), mk16(3840, 2160, 20 + i)
    f16.append([(y[0], y[1], 7680, 4320, jl), (cb[0], cb[1], 3840, 2160, jc), (cr[0], cr[1], 3840, 2160, jc)])


def build(paired):
    os.environ["MDCT_PAIRED_ROWS"] = "1" if paired else "0"
    u8 = [M.Batch("roundtrip_u8", f) for f in frames]
    q = [M.Batch("q32", [(a, o, w, h, l) for (a, b, w, h, _), o, l in zip(f, os_, ql)]) for f, os_ in zip(frames, qo)]
    i16 = [M.Batch("roundtrip", f) for f in f16]
    return u8, q, i16


variants = {"rows paired": build(True), "one grid per row": build(False)}
# same bytes either way
outs = {}
for name, (u8, q, i16) in variants.items():
    u8[0].run(); q[0].run(); i16[0].run()
    torch.cuda.synchronize()
    outs[name] = [x[1].clone() for x in frames[0]] + [o.clone() for o in qo[0]] + [x[1].clone() for x in f16[0]]
a, b = outs.values()
print("outputs equal:", all(torch.equal(x, y) for x, y in zip(a, b)), flush=True)


def med(calls, reps=100):
    for i in range(300):
        calls[i % len(calls)]()
    r = []
    for k in range(11):
        t.start()
        for i in range(reps):
            calls[i % len(calls)]()
        t.stop()
        r.append(t.elapsed_ms() / reps * 1e3)
    r.sort()
    return r[len(r) // 2], r[0]


for rnd in range(3):
    for name, (u8, q, i16) in variants.items():
        mu, lu = med([b.prepared() for b in u8])
        mq, lq = med([b.prepared() for b in q])
        mi, li = med([b.prepared() for b in i16])
        print(f"round {rnd}  {name:18s}  k_u8_batch frame {mu:6.2f} us (min {lu:6.2f})   k_q32_batch frame {mq:6.2f} us (min {lq:6.2f})   k_i16_batch frame (int16, 4 B/px) {mi:6.2f} us (min {li:6.2f})", flush=True)
