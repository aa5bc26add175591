"""Steady-state timing of every engine entry point on BASELINE.json-sized inputs.
   python3 tools/time_all.py   -> one line per kernel: us, algorithmic GB/s, % of 8 TB/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth

W = H = 8192
M.init(0)
NS = 4
i16 = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(NS)]
o16 = [torch.empty_like(s) for s in i16]
u8 = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + 50 + i).reshape(-1) for i in range(NS)]
o8 = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(NS)]
f32 = [s.to(torch.float32) for s in i16[:2]]
of32 = [torch.empty_like(s) for s in f32]
lut2000 = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
jpeg = (M.QUANTIZE_BASE * np.float32(100)).astype(np.float32)
# 4:2:0 frame (config 3)
Y = synth.plane_i16_torch(7680, 4320, "photo"); Cb = synth.plane_i16_torch(3840, 2160, "photo", seed=1); Cr = synth.plane_i16_torch(3840, 2160, "photo", seed=2)
oY, oCb, oCr = torch.empty_like(Y), torch.empty_like(Cb), torch.empty_like(Cr)
frame = [(Y, oY, 7680, 4320, jpeg), (Cb, oCb, 3840, 2160, jpeg), (Cr, oCr, 3840, 2160, jpeg)]
frame_px = 7680 * 4320 + 2 * 3840 * 2160

P = M.prepare_plane_i16
Q = M.prepare_fwd_quant_u8
cases = [
    ("stream copy (roofline)", 4, W * H, [M.prepare_stream_copy(i16[i], o16[i], W * H * 2) for i in range(NS)]),
    ("i16 roundtrip", 4, W * H, [P("roundtrip", i16[i], o16[i], W, H) for i in range(NS)]),
    ("i16 roundtrip + table", 4, W * H, [P("roundtrip", i16[i], o16[i], W, H, lut=jpeg) for i in range(NS)]),
    ("i16 fwd", 4, W * H, [P("fwd", i16[i], o16[i], W, H) for i in range(NS)]),
    ("i16 fwd + table", 4, W * H, [P("fwd", i16[i], o16[i], W, H, lut=jpeg) for i in range(NS)]),
    ("i16 inv", 4, W * H, [P("inv", i16[i], o16[i], W, H) for i in range(NS)]),
    ("i16 inv + table", 4, W * H, [P("inv", i16[i], o16[i], W, H, lut=jpeg) for i in range(NS)]),
    ("4:2:0 frame roundtrip+tables", 4, frame_px, [M.prepare_roundtrip_i16_planes(frame)]),
    ("f32 fwd", 8, W * H, [lambda i=i: M.fwd_f32(f32[i], of32[i], W, H) for i in range(2)]),
    ("f32 inv", 8, W * H, [lambda i=i: M.inv_f32(f32[i], of32[i], W, H) for i in range(2)]),
    ("u8 px -> i16 coef (+table)", 3, W * H, [M.prepare_u8_i16("fwd", u8[i], o16[i], W, H, lut=jpeg) for i in range(NS)]),
    ("i16 coef -> u8 px (+table)", 3, W * H, [M.prepare_u8_i16("inv", i16[i], o8[i], W, H, lut=jpeg) for i in range(NS)]),
    ("u8 q32 / AVX2 tier", 2, W * H, [Q(u8[i], o8[i], lut2000, W, H, 0, H // 8) for i in range(NS)]),
    ("u8 stereo / SSE tier", 2, W * H, [Q(u8[i], o8[i], lut8, W, H, 0, H // 16, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SSE) for i in range(NS)]),
    ("u8 stereo / scalar tier", 2, W * H, [Q(u8[i], o8[i], lut8, W, H, 0, H // 16, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SCALAR) for i in range(NS)]),
    ("u8 encq / SSE tier", 2, W * H, [Q(u8[i], o8[i], lut8, W, H, 0, H // 8, layout=M.LAYOUT_BLOCK_SSE, profile=M.PROFILE_REF_SSE) for i in range(NS)]),
    ("u8 encq / scalar tier", 2, W * H, [Q(u8[i], o8[i], lut8, W, H, 0, H // 8, layout=M.LAYOUT_BLOCK, profile=M.PROFILE_REF_SCALAR) for i in range(NS)]),
]
# stages either side of the transform: algorithmic bytes per pixel = what must cross HBM once
lut60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
qcoef = [torch.empty_like(s) for s in i16]
for i in range(NS):
    M.fwd_i16(i16[i], qcoef[i], W, H, lut=lut60)  # sparse, photo-like quantised coefficients
    M.fwd_quant_u8(u8[i], o8[i], lut2000, W, H, 0, H // 8)
q32b = [o.clone() for o in o8]
st8, bl8 = [], []
for i in range(2):  # the same pictures in the reference's other two intact layouts
    M.fwd_quant_u8(u8[i], o8[i], lut8, W, H, 0, H // 16, layout=M.LAYOUT_STEREO, profile=M.PROFILE_REF_SSE)
    st8.append(o8[i].clone())
    M.fwd_quant_u8(u8[i], o8[i], lut8, W, H, 0, H // 8, layout=M.LAYOUT_BLOCK, profile=M.PROFILE_REF_SCALAR)
    bl8.append(o8[i].clone())
nblk = (W // 8) * (H // 8)
lv = [torch.empty((nblk, 64), dtype=torch.int16, device="cuda") for _ in range(2)]
rn = [torch.empty((nblk, 64), dtype=torch.uint8, device="cuda") for _ in range(2)]
ct = [torch.empty((nblk,), dtype=torch.uint8, device="cuda") for _ in range(2)]
ycc = [torch.stack([u8[(i + k) % NS].reshape(H, W) for k in range(3)], dim=-1).contiguous() for i in range(2)]
sy = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in range(2)]
scb = [torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda") for _ in range(2)]
scr = [torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda") for _ in range(2)]
for i in range(2):
    M.zigzag_rle_i16(qcoef[i], W, H, lv[i], rn[i], ct[i])
# the same planes under the Annex K.1 luminance table: the sparser records of an ordinary-quality JPEG
K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
lvk = [torch.empty_like(lv[0]) for _ in range(2)]
rnk = [torch.empty_like(rn[0]) for _ in range(2)]
ctk = [torch.empty_like(ct[0]) for _ in range(2)]
for i in range(2):
    M.fwd_i16(i16[i], o16[i], W, H, lut=K1)
    M.zigzag_rle_i16(o16[i], W, H, lvk[i], rnk[i], ctk[i])
torch.cuda.synchronize()
print(f"records: {float(ct[0].float().mean()):.1f} pairs per block (quality-60 table), {float(ctk[0].float().mean()):.1f} (Annex K.1 table)")
Q60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
hstride = M.huffman_seg_stride(W)
hseg = [torch.empty(((H // 8) * hstride,), dtype=torch.uint8, device="cuda") for _ in range(2)]
hnb = [torch.empty((H // 8,), dtype=torch.int32, device="cuda") for _ in range(2)]
pscan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
poff = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
pwork = torch.zeros((H // 8 + 2,), dtype=torch.int64, device="cuda")  # the one-launch encoder's chain between the rows: zeroed once
pff = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
for i in range(2):
    M.huffman_rows(lv[i], rn[i], ct[i], W, H, hseg[i], hnb[i])
cases += [
    ("Huffman rows from records (3 B/px in)", 3.016, W * H, [lambda i=i: M.huffman_rows(lv[i % 2], rn[i % 2], ct[i % 2], W, H, hseg[i % 2], hnb[i % 2]) for i in range(2)]),
    ("Huffman rows, Annex K.1 records", 3.016, W * H, [lambda i=i: M.huffman_rows(lvk[i % 2], rnk[i % 2], ctk[i % 2], W, H, hseg[i % 2], hnb[i % 2]) for i in range(2)]),
    ("JPEG scan pack (stuffing + RSTm) of those rows", 0.45, W * H, [lambda i=i: M.jpeg_pack_rows(hseg[i % 2], hnb[i % 2], hstride, H // 8, pscan, poff) for i in range(2)]),
    ("zig-zag scan, i16 (2+2 B/px)", 4, W * H, [lambda i=i: M.zigzag_rle_i16(qcoef[i], W, H, lv[i % 2]) for i in range(NS)]),
    ("zig-zag + run/level, i16 (2+3)", 5.016, W * H, [lambda i=i: M.zigzag_rle_i16(qcoef[i], W, H, lv[i % 2], rn[i % 2], ct[i % 2]) for i in range(NS)]),
    ("zig-zag + run/level, q32 (1+3)", 4.016, W * H, [lambda i=i: M.zigzag_rle_q32(q32b[i], W, H, lv[i % 2], rn[i % 2], ct[i % 2]) for i in range(NS)]),
    ("zig-zag + run/level, stereo planes (1+3)", 4.016, W * H, [lambda i=i: M.zigzag_rle_u8(st8[i % 2], M.LAYOUT_STEREO, W, H, lv[i % 2], rn[i % 2], ct[i % 2]) for i in range(2)]),
    ("zig-zag + run/level, encq blocks (1+3)", 4.016, W * H, [lambda i=i: M.zigzag_rle_u8(bl8[i % 2], M.LAYOUT_BLOCK, W, H, lv[i % 2], rn[i % 2], ct[i % 2]) for i in range(2)]),
    ("u8 px -> records, fused (1+3 B/px)", 4.016, W * H, [lambda i=i: M.fwd_u8_records(u8[i], W, H, lv[i % 2], rn[i % 2], ct[i % 2], lut=K1) for i in range(NS)]),
    ("i16 plane -> records, fused (2+3 B/px)", 5.016, W * H, [lambda i=i: M.fwd_i16_records(i16[i], W, H, lv[i % 2], rn[i % 2], ct[i % 2], lut=K1) for i in range(NS)]),
    ("u8 px -> Huffman rows, ONE kernel, q60 table (1+0.2 B/px)", 1.22, W * H, [lambda i=i: M.fwd_u8_huffman_rows(u8[i], W, H, hseg[i % 2], hnb[i % 2], lut=Q60) for i in range(NS)]),
    ("u8 px -> Huffman rows, ONE kernel, Annex K.1 table", 1.15, W * H, [lambda i=i: M.fwd_u8_huffman_rows(u8[i], W, H, hseg[i % 2], hnb[i % 2], lut=K1) for i in range(NS)]),
    ("i16 plane -> Huffman rows, ONE kernel, Annex K.1 table", 2.15, W * H, [lambda i=i: M.fwd_i16_huffman_rows(i16[i], W, H, hseg[i % 2], hnb[i % 2], lut=K1) for i in range(NS)]),
    ("u8 px -> finished JPEG scan, ONE launch, Annex K.1 table", 1.15, W * H, [lambda i=i: M.fwd_u8_jpeg_scan(u8[i], W, H, hseg[i % 2], pwork, pscan, poff, lut=K1) for i in range(NS)]),
    ("  the same as two launches: fused kernel + counted pack", 1.15, W * H, [lambda i=i: (M.fwd_u8_huffman_rows(u8[i], W, H, hseg[i % 2], hnb[i % 2], lut=K1, ff_counts=pff), M.jpeg_pack_rows(hseg[i % 2], hnb[i % 2], hstride, H // 8, pscan, poff, ff_counts=pff)) for i in range(NS)]),
    ("4:2:0 split (3+3 B/px)", 6, W * H, [lambda i=i: M.split420_u8(ycc[i], W, H, sy[i], scb[i], scr[i]) for i in range(2)]),
    ("4:2:0 split into 8-bit planes (3+1.5 B/px)", 4.5, W * H, [lambda i=i: M.split420_u8_planes(ycc[i], W, H, sy[i].view(torch.uint8), scb[i].view(torch.uint8), scr[i].view(torch.uint8)) for i in range(2)]),
]
# config 4 on one GPU: 256 independent 4096x4096 int16 planes, forward only.  Blocks are
# independent, so a batch stacked in memory IS one tall plane: one launch, no per-plane drain.
NB = 256
one = synth.plane_i16_torch(4096, 4096, "photo", seed=77)
batch_in = one.repeat(NB, 1)            # [256*4096, 4096] int16, 8.6 GB
batch_out = torch.empty_like(batch_in)
cases.append(("config 4: 256 x 4096^2 i16 fwd", 4, NB * 4096 * 4096, [P("fwd", batch_in, batch_out, 4096, NB * 4096)]))
cases.append(("  same, one launch per plane", 4, 4096 * 4096, [P("fwd", batch_in[i * 4096:(i + 1) * 4096], batch_out[i * 4096:(i + 1) * 4096], 4096, 4096) for i in range(16)]))
t = M.Timer()
print(f"{'kernel':32s} {'us':>8s} {'Mpx/s':>10s} {'alg GB/s':>9s} {'% of 8 TB/s':>11s}")
for name, bpp, px, calls in cases:
    big = px > 1 << 30
    for i in range(3 if big else 300):
        calls[i % len(calls)]()
    best = []
    reps = 3 if big else 40
    for r in range(5):
        t.start()
        for i in range(reps):
            calls[i % len(calls)]()
        t.stop()
        best.append(t.elapsed_ms() / reps)
    best.sort()
    ms = best[len(best) // 2]
    gbps = bpp * px / (ms * 1e-3) / 1e9
    print(f"{name:32s} {ms*1e3:8.2f} {px/(ms*1e-3)/1e6:10.0f} {gbps:9.1f} {gbps/80:10.1f}%")
