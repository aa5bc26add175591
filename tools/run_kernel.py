"""Launch ONE engine kernel repeatedly on the bench workload (for rocprofv3 --pmc passes):
    python3 tools/run_kernel.py {roundtrip|fwd|inv|q32|stereo_sse|encq_sse|stereo_scalar|encq_scalar|scan_q32|u8_records|copy|huffman|huffman_k1|px_huffman|px_huffman_k1} [launches]"""
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
    q = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32) if which == "huffman" else np.array(
        [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
         18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
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
K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
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
    elif which == "copy":
        M.stream_copy(srcs[s], dsts[s], W * H * 2)
torch.cuda.synchronize()
print("done", which, n)
