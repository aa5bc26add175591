"""Example: a baseline JPEG written by the engine's stages (tools/ = not part of the product path).
    python3 tools/gpu_jpeg.py out.jpg [raw_grey_file X Y | synthetic X Y]
pixels -> mdct_fwd_u8_records (Annex K.1 table; = mdct_fwd_u8_i16 + mdct_zigzag_rle_i16 in one pass) -> mdct_huffman_rows
-> mdct_jpeg_pack_rows (stuffing + RSTm, one contiguous scan) -> simd_dct_amd.jfif.write_jpeg (the marker segments)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import jfif, synth

K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
out = sys.argv[1] if len(sys.argv) > 1 else "out.jpg"
src = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2160 - 2160 % 8)
M.init(0)
img = synth.plane_u8_torch(W, H, "photo") if src == "synthetic" else torch.from_numpy(np.fromfile(src, dtype=np.uint8)[: W * H].reshape(H, W)).cuda()
nblk = (W // 8) * (H // 8)
lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
stride = M.huffman_seg_stride(W)
seg = torch.empty(((H // 8) * stride,), dtype=torch.uint8, device="cuda")
nb = torch.empty((H // 8,), dtype=torch.int32, device="cuda")
scan = torch.empty((W * H // 2,), dtype=torch.uint8, device="cuda")
off = torch.zeros((H // 8 + 1,), dtype=torch.int64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    M.fwd_u8_records(img, W, H, lv, rn, ct, lut=K1)
    M.huffman_rows(lv, rn, ct, W, H, seg, nb)
    M.jpeg_pack_rows(seg, nb, stride, H // 8, scan, off)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
t0 = time.perf_counter()
total = int(off[-1].item())
assert total <= scan.numel()
data = jfif.write_jpeg([dict(scan=scan[:total].cpu().numpy(), blocks_per_row=W // 8, qtable=K1)], W, H)
host_ms = (time.perf_counter() - t0) * 1e3
open(out, "wb").write(data)
print(f"{W}x{H}: three device stages {dt * 1e6:.0f} us ({W * H / dt / 1e6:.0f} Mpx/s), file {len(data)} bytes ({8 * len(data) / (W * H):.2f} bit/px) -> {out}; copy back + header {host_ms:.1f} ms")
