"""Launch ONE engine kernel repeatedly on the bench workload (for rocprofv3 --pmc passes):
    python3 tools/run_kernel.py {roundtrip|fwd|inv|q32|stereo_sse|encq_sse|stereo_scalar|encq_scalar|scan_q32|u8_records|copy|huffman|huffman_k1|px_huffman|px_huffman_k1|
                                 frame420|batch<N>|f32|roundtrip_lut} [launches]
    frame420: BASELINE.json configs[2] in one call (k_i16_batch, round trip, Annex-K tables); batch256: configs[3] on one GPU, 256 separately
    allocated 4096^2 planes forward in one launch (device-table batch); f32: configs[4], k_f32_tile forward"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import simd_dct_amd as M
from simd_dct_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "roundtrip"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W = H = 8192
M.init(0)
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
lut = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
u8s = [synth.plane_u8_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
u8d = [torch.empty(W * H, dtype=torch.uint8, device="cuda") for _ in range(4)]
if which.startswith("huffman"):
    # records of quantised coefficients: "huffman" = the dense time_all.py case (22 pairs per block), "huffman_k1" = Annex K.1 table (5 pairs)
    q = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32) if which == "huffman" else synth.JPEG_LUMA
    nblk = (W // 8) * (H // 8)
    recs = []
    for s_ in range(2):
        M.fwd_i16(srcs[s_], dsts[s_], W, H, lut=q)
        lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
        rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
        ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
        M.zigzag_rle_i16(dsts[s_], W, H, lv, rn, ct)
        recs.append((lv, rn, ct))
    hstride = M.huffman_seg_stride(W)
    hseg = torch.empty(((H // 8) * hstride,), dtype=torch.uint8, device="cuda")
    hnb = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    print("pairs per block", float(recs[0][2].float().mean()))
K1 = synth.JPEG_LUMA
if which.startswith("px_huffman"):  # the fused pixels -> Huffman rows kernel: dense quality-60 table, or Annex K.1
    pq = K1 if which.endswith("_k1") else (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
    hstride = M.huffman_seg_stride(W)
    hseg = torch.empty(((H // 8) * hstride,), dtype=torch.uint8, device="cuda")
    hnb = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
    hff = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
U8 = {"stereo_sse": (M.LAYOUT_STEREO, M.PROFILE_REF_SSE, H // 16), "encq_sse": (M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, H // 8),
      "stereo_scalar": (M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, H // 16), "encq_scalar": (M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, H // 8)}
if which in ("scan_q32", "u8_records"):
    nblk = (W // 8) * (H // 8)
    lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
    rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
    ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
    q60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
    for s_ in range(4):
        M.fwd_quant_u8(u8s[s_], u8d[s_], lut, W, H, 0, H // 8)
if which == "frame420":
    frames = [[(synth.plane_i16_torch(w, h, "photo", seed=synth.SEED + so + 10 * f), None, w, h, synth.JPEG_LUMA if tab == "luma" else synth.JPEG_CHROMA) for (w, h, so, tab) in synth.CONFIG3_PLANES] for f in range(4)]
    frames = [[(a, torch.empty_like(a), w, h, l) for (a, _, w, h, l) in f] for f in frames]
if which.startswith("batch") and which[5:].isdigit():  # batch256 = configs[3] on one GPU; batch32 for counter passes (every torch kernel that
    del srcs, dsts, u8s, u8d                              # builds an input is slowed by the counters too: 256 planes take minutes under --pmc)
    nb = int(which[5:])
    pl = [synth.plane_i16_torch(4096, 4096, "photo", seed=synth.SEED + 100 + p) for p in range(nb)]
    b256 = M.Batch("fwd", [(a, torch.empty_like(a), 4096, 4096, None) for a in pl])
    print("planes ready", nb, flush=True)
if which == "f32":
    fsrc = [t.to(torch.float32) for t in srcs[:2]]
    fdst = [torch.empty_like(t) for t in fsrc]
torch.cuda.synchronize()
for i in range(n):
    s = i % 4
    if which == "roundtrip":
        M.roundtrip_i16(srcs[s], dsts[s], W, H)
    elif which == "fwd":
        M.fwd_i16(srcs[s], dsts[s], W, H)
    elif which == "inv":
        M.inv_i16(srcs[s], dsts[s], W, H)
    elif which == "q32":
        M.fwd_quant_u8(u8s[s], u8d[s], lut, W, H, 0, H // 8)
    elif which in U8:
        M.fwd_quant_u8(u8s[s], u8d[s], lut8, W, H, 0, U8[which][2], layout=U8[which][0], profile=U8[which][1])
    elif which == "scan_q32":
        M.zigzag_rle_q32(u8d[s], W, H, lv, rn, ct)
    elif which == "u8_records":
        M.fwd_u8_records(u8s[s], W, H, lv, rn, ct, lut=q60)
    elif which.startswith("px_huffman"):
        M.fwd_u8_huffman_rows(u8s[s], W, H, hseg, hnb, lut=pq, ff_counts=hff)
    elif which.startswith("huffman"):
        M.huffman_rows(*recs[i % 2], W, H, hseg, hnb)
    elif which == "frame420":
        M.roundtrip_i16_planes(frames[s])
    elif which.startswith("batch") and which[5:].isdigit():
        b256.run()
    elif which == "f32":
        M.fwd_f32(fsrc[i % 2], fdst[i % 2], W, H)
    elif which == "roundtrip_lut":
        M.roundtrip_i16(srcs[s], dsts[s], W, H, lut=K1)
    elif which == "copy":
        M.stream_copy(srcs[s], dsts[s], W * H * 2)
torch.cuda.synchronize()
print("done", which, n)
