"""Example: a baseline JPEG written by the engine's stages (tools/ = not part of the product path).
    python3 tools/gpu_jpeg.py out.jpg [synthetic | synthetic-color | raw_grey_file] [X Y]
Default (round 3), grey: ONE launch, pixels -> finished scan (mdct_fwd_u8_jpeg_scan).  Colour: per plane the fused pixels -> Huffman rows
kernel (mdct_fwd_i16_huffman_rows), then mdct_jpeg_pack_rows_counted -- three one-launch encoders side by side on three streams wait on each
other's rows and measure 2 % slower (MDCT_JPEG_ONE_LAUNCH=1 selects them anyway; MDCT_JPEG_TWO_LAUNCH=1 the two launches for grey too).
MDCT_JPEG_STAGED=1 selects the staged path of round 2:
grey:   pixels -> mdct_fwd_u8_records (Annex K.1 table; = mdct_fwd_u8_i16 + mdct_zigzag_rle_i16 in one pass) -> mdct_huffman_rows
        -> mdct_jpeg_pack_rows (stuffing + RSTm, one contiguous scan) -> simd_dct_amd.jfif.write_jpeg (the marker segments)
colour: interleaved 8-bit YCbCr -> mdct_split420_u8 -> per plane mdct_fwd_i16_records (K.1 / K.2) -> mdct_huffman_rows -> mdct_jpeg_pack_rows;
        three non-interleaved scans (Y at full resolution, Cb / Cr at half)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import jfif, synth

K1 = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
               18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
K2 = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.float32)
STAGED = os.environ.get("MDCT_JPEG_STAGED") == "1"
COLOUR = len(sys.argv) > 2 and sys.argv[2] == "synthetic-color"
TWO = os.environ.get("MDCT_JPEG_TWO_LAUNCH") == "1" or (COLOUR and os.environ.get("MDCT_JPEG_ONE_LAUNCH") != "1")
out = sys.argv[1] if len(sys.argv) > 1 else "out.jpg"
src = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 2160 - 2160 % 16)
M.init(0)


class Plane:
    """device buffers of one component: records, row segments, packed scan"""

    def __init__(self, w, h, qtable, chroma):
        self.w, self.h, self.q, self.chroma = w, h, qtable, chroma
        nblk = (w // 8) * (h // 8)
        if STAGED:  # the fused path needs no record arrays (3 bytes per pixel)
            self.lv = torch.empty((nblk, 64), dtype=torch.int16, device="cuda")
            self.rn = torch.empty((nblk, 64), dtype=torch.uint8, device="cuda")
            self.ct = torch.empty((nblk,), dtype=torch.uint8, device="cuda")
        self.stride = M.huffman_seg_stride(w)
        self.seg = torch.empty(((h // 8) * self.stride,), dtype=torch.uint8, device="cuda")
        self.nb = torch.empty((h // 8,), dtype=torch.int32, device="cuda")
        self.ff = None if STAGED else torch.empty((h // 8,), dtype=torch.int32, device="cuda")  # 0xFF bytes per row, counted by the fused kernel
        self.scan = torch.empty((w * h // 2,), dtype=torch.uint8, device="cuda")
        self.off = torch.zeros((h // 8 + 1,), dtype=torch.int64, device="cuda")
        self.work = torch.zeros((h // 8 + 2,), dtype=torch.int64, device="cuda")  # the one-launch form's chain between the rows: zeroed once

    def entropy(self, stream=None):
        M.huffman_rows(self.lv, self.rn, self.ct, self.w, self.h, self.seg, self.nb, chroma=self.chroma, stream=stream)
        self.pack(stream)

    def pack(self, stream=None):
        M.jpeg_pack_rows(self.seg, self.nb, self.stride, self.h // 8, self.scan, self.off, ff_counts=self.ff, stream=stream)

    def component(self):
        total = int(self.off[-1].item())
        assert total <= self.scan.numel()
        return dict(scan=self.scan[:total].cpu().numpy(), blocks_per_row=self.w // 8, qtable=self.q)


if src == "synthetic-color":
    ycc = torch.stack([synth.plane_u8_torch(W, H, "photo", seed=s) for s in (5, 6, 7)], dim=-1).contiguous()
    y = torch.empty((H, W), dtype=torch.int16, device="cuda")
    cb = torch.empty((H // 2, W // 2), dtype=torch.int16, device="cuda")
    cr = torch.empty_like(cb)
    planes = [Plane(W, H, K1, False), Plane(W // 2, H // 2, K2, True), Plane(W // 2, H // 2, K2, True)]

    side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def encode():
        # the three planes are independent after the split: Y on the current stream, Cb and Cr on two side streams
        # (fork / join by events, so the same code is capturable as one hipGraph with three parallel branches)
        cur = torch.cuda.current_stream()
        M.split420_u8(ycc, W, H, y, cb, cr)
        for st in side_streams:
            st.wait_stream(cur)
        for p, s, st in zip(planes, (y, cb, cr), (cur, side_streams[0], side_streams[1])):
            if STAGED:
                M.fwd_i16_records(s, p.w, p.h, p.lv, p.rn, p.ct, lut=p.q, stream=st.cuda_stream)
                p.entropy(stream=st.cuda_stream)
            elif TWO:
                M.fwd_i16_huffman_rows(s, p.w, p.h, p.seg, p.nb, lut=p.q, chroma=p.chroma, ff_counts=p.ff, stream=st.cuda_stream)
                p.pack(stream=st.cuda_stream)
            else:
                M.fwd_i16_jpeg_scan(s, p.w, p.h, p.seg, p.work, p.scan, p.off, lut=p.q, chroma=p.chroma, stream=st.cuda_stream)
        for st in side_streams:
            cur.wait_stream(st)
    what = "4:2:0 colour, 10 launches on 3 streams (staged)" if STAGED else ("4:2:0 colour, split + 3 x (fused kernel + pack) on 3 streams" if TWO else "4:2:0 colour, split + 3 x one launch on 3 streams")
else:
    img = synth.plane_u8_torch(W, H, "photo") if src == "synthetic" else torch.from_numpy(np.fromfile(src, dtype=np.uint8)[: W * H].reshape(H, W)).cuda()
    planes = [Plane(W, H, K1, False)]

    def encode():
        p = planes[0]
        if STAGED:
            M.fwd_u8_records(img, W, H, p.lv, p.rn, p.ct, lut=K1)
            p.entropy()
        elif TWO:
            M.fwd_u8_huffman_rows(img, W, H, p.seg, p.nb, lut=K1, ff_counts=p.ff)
            p.pack()
        else:
            M.fwd_u8_jpeg_scan(img, W, H, p.seg, p.work, p.scan, p.off, lut=K1)
    what = "grey, 3 stages" if STAGED else ("grey, fused kernel + pack" if TWO else "grey, one launch")

for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    encode()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
# the same launches captured once as a hipGraph and replayed (what a per-frame encoder loop would do)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    encode()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    encode()
for rep in range(20):
    g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
nrep = 200
for rep in range(nrep):
    g.replay()
torch.cuda.synchronize(); dt_graph = (time.perf_counter() - t0) / nrep
t0 = time.perf_counter()
data = jfif.write_jpeg([p.component() for p in planes], W, H)
host_ms = (time.perf_counter() - t0) * 1e3
open(out, "wb").write(data)
print(f"{W}x{H} {what}: device {dt * 1e6:.0f} us ({W * H / dt / 1e6:.0f} Mpx/s), as a replayed hipGraph {dt_graph * 1e6:.0f} us per frame ({1 / dt_graph:.0f} frames/s), file {len(data)} bytes ({8 * len(data) / (W * H):.2f} bit/px) -> {out}; copy back + header {host_ms:.1f} ms")
