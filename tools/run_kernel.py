"""Launch engine kernels on the bench workloads -- the target of the rocprofv3 --pmc passes (tools/profile_round.sh):

    python3 tools/run_kernel.py <name>[,<name>...] | all  [launches = 6]

Every input is built on the HOST (numpy, simd_dct_amd.synth) and uploaded, or replicated on the device with the library's own stream
copy: no torch kernel runs, so a counter pass costs what the engine's launches cost (round 4 could not finish the 256-plane batch under
--pmc because torch's input builders ran under the counters too).  torch is used for allocation (torch.empty) and copies only.
One process can run all kernels in turn (`all`): a counter pass is then ONE rocprofv3 run; tools/pmc_round.py tells the kernels apart
by name and grid size.

    roundtrip roundtrip_lut fwd inv copy        8192^2 int16 (k_i16_tile<...>, k_stream_copy)
    q32 stereo_sse stereo_scalar encq_sse encq_scalar   the reference's five behaviours, 8192^2 (k_q32_tile, k_fwd_quant_u8<...>)
    f32                                          configs[4]: k_f32_tile forward
    frame420 frame420_u8                         configs[2]: the 8K 4:2:0 frame as int16 planes (k_i16_batch) / as 8-bit planes (k_u8_batch), one launch
    frame420_u8_fwd frame420_u8_inv              the two halves of the 8-bit frame: pixels -> int16 coefficients / back, one launch each (k_u8_batch<1|2>)
    frame420_q32                                 the same frame as the reference's q32 product, one launch (k_q32_batch)
    batch256                                     configs[3] on one GPU: 256 separately allocated 4096^2 planes, forward, ONE launch (17.2 GB)
    u8_i16_fwd u8_i16_inv                        one plane: k_u8_i16_fwd / a batch of one through k_u8_batch<U8_INV>
    scan_i16 scan_q32 u8_records split420 split420_u8_planes huffman px_huffman jpeg_scan   the stages either side (8192^2)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import simd_dct_amd as M
from simd_dct_amd import synth

ALL = ["copy", "roundtrip", "roundtrip_lut", "fwd", "inv", "q32", "stereo_sse", "stereo_scalar", "encq_sse", "encq_scalar", "f32", "frame420", "frame420_u8", "frame420_u8_fwd", "frame420_u8_inv", "frame420_q32", "u8_i16_fwd", "u8_i16_inv",
       "scan_i16", "scan_q32", "u8_records", "split420", "split420_u8_planes", "huffman", "px_huffman", "jpeg_scan", "batch256"]
which = sys.argv[1] if len(sys.argv) > 1 else "roundtrip"
names = ALL if which == "all" else which.split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = H = 8192
M.init(0)


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()  # host -> device copy, no kernel


def empty(shape, dtype):
    return torch.empty(shape, dtype=dtype, device="cuda")


def clone_on_device(t):
    """a second buffer with the same bytes, made by the library's stream copy (bytes % 16 == 0)"""
    o = torch.empty_like(t)
    M.stream_copy(t, o, t.numel() * t.element_size())
    return o


K1, K2 = synth.JPEG_LUMA, synth.JPEG_CHROMA
lut2000 = (M.QUANTIZE_BASE * np.float32(2000)).astype(np.float32)
lut8 = (M.QUANTIZE_BASE * np.float32(8)).astype(np.float32)
q60 = (M.QUANTIZE_BASE * np.float32(60)).astype(np.float32)
_cache = {}


def i16_planes():
    if "i16" not in _cache:
        a = up(synth.plane_i16_np(W, H, "photo"))
        _cache["i16"] = ([a, clone_on_device(a)], [empty((H, W), torch.int16) for _ in range(2)])
    return _cache["i16"]


def u8_planes():
    if "u8" not in _cache:
        a = up(synth.plane_u8_np(W, H, "photo"))
        _cache["u8"] = ([a, clone_on_device(a)], [empty((W * H,), torch.uint8) for _ in range(2)])
    return _cache["u8"]


def drop(*keys):
    for k in keys:
        _cache.pop(k, None)
    torch.cuda.empty_cache()


def run(name):
    U8 = {"stereo_sse": (M.LAYOUT_STEREO, M.PROFILE_REF_SSE, H // 16), "encq_sse": (M.LAYOUT_BLOCK_SSE, M.PROFILE_REF_SSE, H // 8),
          "stereo_scalar": (M.LAYOUT_STEREO, M.PROFILE_REF_SCALAR, H // 16), "encq_scalar": (M.LAYOUT_BLOCK, M.PROFILE_REF_SCALAR, H // 8)}
    if name in ("roundtrip", "roundtrip_lut", "fwd", "inv", "copy"):
        s, d = i16_planes()
        for i in range(n):
            if name == "roundtrip":
                M.roundtrip_i16(s[i % 2], d[i % 2], W, H)
            elif name == "roundtrip_lut":
                M.roundtrip_i16(s[i % 2], d[i % 2], W, H, lut=K1)
            elif name == "fwd":
                M.fwd_i16(s[i % 2], d[i % 2], W, H)
            elif name == "inv":
                M.inv_i16(s[i % 2], d[i % 2], W, H)
            else:
                M.stream_copy(s[i % 2], d[i % 2], W * H * 2)
    elif name == "q32" or name in U8:
        s, d = u8_planes()
        for i in range(n):
            if name == "q32":
                M.fwd_quant_u8(s[i % 2], d[i % 2], lut2000, W, H, 0, H // 8)
            else:
                M.fwd_quant_u8(s[i % 2], d[i % 2], lut8, W, H, 0, U8[name][2], layout=U8[name][0], profile=U8[name][1])
    elif name == "f32":
        drop("i16", "u8")
        a = up(synth.plane_i16_np(W, H, "photo").astype(np.float32))
        o = empty((H, W), torch.float32)
        for i in range(n):
            M.fwd_f32(a, o, W, H)
    elif name == "frame420_q32":
        drop("i16", "u8")
        frames = []
        for f in range(2):
            pl = []
            for (w, h, so, tab) in synth.CONFIG3_PLANES:
                a = up(synth.plane_u8_np(w, h, "photo", seed=synth.SEED + so + 10 * f))
                pl.append((a, empty((w * h,), torch.uint8), w, h, lut2000 if tab == "luma" else (M.QUANTIZE_BASE * np.float32(1200)).astype(np.float32)))
            frames.append(pl)
        b = [M.Batch("q32", f) for f in frames]
        for i in range(n):
            b[i % 2].run()
        torch.cuda.synchronize()
    elif name in ("frame420_u8_fwd", "frame420_u8_inv"):
        drop("i16", "u8")
        frames = []
        for f in range(2):
            pl = []
            for (w, h, so, tab) in synth.CONFIG3_PLANES:
                a = up(synth.plane_u8_np(w, h, "photo", seed=synth.SEED + so + 10 * f))
                pl.append((a, empty((h, w), torch.int16), w, h, K1 if tab == "luma" else K2))
            frames.append(pl)
        fb = [M.Batch("fwd_u8_i16", f) for f in frames]
        for b in fb:
            b.run()  # (the inverse reads real coefficients)
        b = fb if name.endswith("fwd") else [M.Batch("inv_i16_u8", [(torch.empty_like(p), c, w, h, l) for (p, c, w, h, l) in f]) for f in frames]
        for i in range(n):
            b[i % 2].run()
        torch.cuda.synchronize()
    elif name in ("frame420", "frame420_u8"):
        drop("i16", "u8")
        u8 = name.endswith("_u8")
        frames = []
        for f in range(2):
            pl = []
            for (w, h, so, tab) in synth.CONFIG3_PLANES:
                a = up((synth.plane_u8_np if u8 else synth.plane_i16_np)(w, h, "photo", seed=synth.SEED + so + 10 * f))
                pl.append((a, torch.empty_like(a), w, h, K1 if tab == "luma" else K2))
            frames.append(pl)
        b = [M.Batch("roundtrip_u8" if u8 else "roundtrip", f) for f in frames]
        for i in range(n):
            b[i % 2].run()
        torch.cuda.synchronize()
    elif name == "batch256":
        drop("i16", "u8")
        base = [up(synth.plane_i16_np(4096, 4096, "photo", seed=synth.SEED + 100 + p)) for p in range(4)]  # four uploaded planes ...
        pl = [base[p] if p < 4 else empty((4096, 4096), torch.int16) for p in range(256)]
        for p in range(4, 256):                                                                              # ... replicated by the library's stream copy
            M.stream_copy(base[p % 4], pl[p], 4096 * 4096 * 2)
        out = [empty((4096, 4096), torch.int16) for _ in range(256)]
        b = M.Batch("fwd", [(a, o, 4096, 4096, None) for a, o in zip(pl, out)])
        assert b.launches == 1
        print("256 separately allocated planes ready", flush=True)
        for i in range(max(2, n // 2)):
            b.run()
        torch.cuda.synchronize()
    elif name in ("u8_i16_fwd", "u8_i16_inv"):
        s, d = u8_planes()
        coef = empty((H, W), torch.int16)
        M.fwd_u8_i16(s[0].view(H, W), coef, W, H, lut=K1)
        for i in range(n):
            if name.endswith("fwd"):
                M.fwd_u8_i16(s[i % 2].view(H, W), coef, W, H, lut=K1)
            else:
                M.inv_i16_u8(coef, d[i % 2].view(H, W), W, H, lut=K1)
    elif name in ("scan_i16", "scan_q32", "u8_records", "huffman", "px_huffman", "jpeg_scan"):
        nblk = (W // 8) * (H // 8)
        lv, rn, ct = empty((nblk, 64), torch.int16), empty((nblk, 64), torch.uint8), empty((nblk,), torch.uint8)
        hstride = M.huffman_seg_stride(W)
        if name in ("scan_i16", "huffman"):
            s, d = i16_planes()
            M.fwd_i16(s[0], d[0], W, H, lut=q60)  # dense quantised coefficients (quality-60 table: ~22 pairs per block)
            M.zigzag_rle_i16(d[0], W, H, lv, rn, ct)
            if name == "scan_i16":
                for i in range(n):
                    M.zigzag_rle_i16(d[0], W, H, lv, rn, ct)
            else:
                hseg, hnb = empty(((H // 8) * hstride,), torch.uint8), empty((H // 8,), torch.int32)
                for i in range(n):
                    M.huffman_rows(lv, rn, ct, W, H, hseg, hnb)
        elif name == "scan_q32":
            s, d = u8_planes()
            M.fwd_quant_u8(s[0], d[0], lut2000, W, H, 0, H // 8)
            for i in range(n):
                M.zigzag_rle_q32(d[0], W, H, lv, rn, ct)
        elif name == "u8_records":
            s, d = u8_planes()
            for i in range(n):
                M.fwd_u8_records(s[i % 2].view(H, W), W, H, lv, rn, ct, lut=q60)
        else:
            s, d = u8_planes()
            hseg, hnb, hff = empty(((H // 8) * hstride,), torch.uint8), empty((H // 8,), torch.int32), empty((H // 8,), torch.int32)
            if name == "px_huffman":
                for i in range(n):
                    M.fwd_u8_huffman_rows(s[i % 2].view(H, W), W, H, hseg, hnb, lut=q60, ff_counts=hff)
            else:
                work = up(np.zeros(H // 8 + 2, dtype=np.int64))
                scan, off = empty((W * H // 2,), torch.uint8), empty((H // 8 + 1,), torch.int64)
                for i in range(n):
                    M.fwd_u8_jpeg_scan(s[i % 2].view(H, W), W, H, hseg, work, scan, off, lut=K1)
    elif name in ("split420", "split420_u8_planes"):
        drop("i16", "u8")
        rng = np.random.default_rng(1)
        ycc = up(rng.integers(0, 256, size=(H, W * 3), dtype=np.uint8))
        dt = torch.int16 if name == "split420" else torch.uint8
        y, cb, cr = empty((H, W), dt), empty((H // 2, W // 2), dt), empty((H // 2, W // 2), dt)
        for i in range(n):
            (M.split420_u8 if name == "split420" else M.split420_u8_planes)(ycc, W, H, y, cb, cr)
    else:
        raise SystemExit(f"unknown kernel name {name!r}; one of {ALL}")
    torch.cuda.synchronize()
    print("done", name, flush=True)


for nm in names:
    run(nm)
