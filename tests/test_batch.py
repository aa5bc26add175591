"""Plane batches (mdct_*_i16_batch, mdct_batch_*): any number of separately allocated planes per call.

CPU: the host half (csrc/batch_plan.h) under ASan + UBSan -- every tile index of a laid-out launch walked through the
kernel's own index arithmetic -- and the status codes that are decided before a device is touched.
GPU (-m gpu): the batch entry points against the oracle and against the single-plane entry points on the same planes;
BASELINE.json configs[2] (8K 4:2:0 frame, one call) and configs[3] (256 separately allocated 4096x4096 planes, forward)
at full size.  The reference's only batching affordance is the caller-side row range, simd_dct.cpp:2243-2261."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

import __graft_entry__ as G
from simd_dct_amd import _lib, api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_layout_and_index_arithmetic_under_sanitizers(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "batch_plan"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "simd_dct_amd", "csrc"), os.path.join(ROOT, "tests", "batch_plan_driver.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr + r.stdout).lower() and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "batch plan ok" in r.stdout, r.stdout + r.stderr


def test_kernel_and_host_share_the_layout_header():
    """the kernels' plane lookup IS the arithmetic the CPU test walked: one header, no second copy of the descriptor, and the batch kernels
    call batch_plan.h's batch_plane_of / magic_apply instead of restating them"""
    k = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_kernels.h")).read()
    assert '#include "batch_plan.h"' in k and "struct BatchDesc" not in k
    hdr = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "batch_plan.h")).read()
    code = "\n".join(l.split("//")[0] for l in hdr.splitlines())
    assert "#include <hip" not in code and "hipError" not in code  # builds with plain g++ (the sanitizer test above does)
    assert "MDCT_HD uint32_t batch_plane_of(" in hdr and "MDCT_HD uint32_t magic_apply(" in hdr
    kern = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_kernels.hip")).read()
    assert "batch_plane_of(w, n, uniform" in kern and kern.count("magic_apply(") >= 1
    assert "magic_quot" not in kern and "__umulhi" not in kern  # no device-side restatement of the division or of the search
    # k_i16_batch, k_u8_batch and k_q32_batch: every batch kernel finds its tile through it, in the form that understands paired rows (batch_pos, the same header)
    assert kern.count("batch_tile<true>(blockIdx.x)") == 3 and "batch_pos(lt, t.d[8]" in kern
    assert "MDCT_HD BatchPos batch_pos(" in hdr
    api_src = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_api.hip")).read()
    assert "mdct::batch_layout(" in api_src


def test_batch_status_codes_without_device():
    G.build_hip()
    lib = _lib.load()
    b = np.zeros(64 * 16, dtype=np.int16)
    ok = (b, b, 64, 16, None)
    for mode in ("fwd", "inv", "roundtrip"):
        assert api.i16_batch(mode, [ok, (b, None, 64, 16, None)], check=False) == 1  # null plane pointer: nothing launched
        assert api.i16_batch(mode, [ok, (b, b, 60, 16, None)], check=False) == 2  # not a multiple of 8x8
        assert api.i16_batch(mode, [ok, (b[1:], b, 64, 8, None)], check=False) == 1  # rows not 16-byte aligned
        assert api.i16_batch(mode, [(b, b, 64, 16, None, 32, 64)], check=False) == 1  # pitch below the width
        bad = np.ones(64, dtype=np.float32)
        bad[5] = 0.0
        assert api.i16_batch(mode, [ok, (b, b, 64, 16, bad)], check=False) == 1  # zero table entry
        assert "table" in api.last_error()
    assert lib.mdct_fwd_i16_batch(None, 2, None) == 1
    assert lib.mdct_fwd_i16_batch(None, -1, None) == 1
    h = ctypes.c_void_p()
    arr, _keep = api._plane_array([ok])
    assert lib.mdct_batch_create(ctypes.byref(h), 7, arr, 1) == 1 and not h  # unknown mode
    assert lib.mdct_batch_create(None, 0, arr, 1) == 1
    assert lib.mdct_batch_run(None, None) == 1
    assert lib.mdct_batch_destroy(None) == 0 and lib.mdct_batch_launches(None) == 0


# ----------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    api.init(0)
    return torch


def _lut(scale):
    return (api.QUANTIZE_BASE * np.float32(scale)).astype(np.float32)


CANARY = -21846


def _planes(torch, shapes, luts, pad=0, seed0=0):
    """separately allocated planes; pad > 0: pitched rows whose padding must survive"""
    import oracle as O  # noqa: F401

    srcs, d_in, d_out = [], [], []
    for i, (w, h) in enumerate(shapes):
        s = synth.plane_i16_np(w, h, "photo", seed=synth.SEED + seed0 + i, bits=12 if i % 2 else 8)
        if pad:
            full = np.full((h, w + pad), 1234, dtype=np.int16)
            full[:, :w] = s
            s = full
        srcs.append(s)
        d_in.append(torch.from_numpy(s).cuda())
        d_out.append(torch.full((h, w + pad + (8 if pad else 0)), CANARY, dtype=torch.int16, device="cuda"))  # (the two pitches differ)
    desc = [(a, b, w, h, l, w + pad, w + pad + (8 if pad else 0)) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)]
    return srcs, d_in, d_out, desc


def _check_against_oracle(mode, srcs, d_out, shapes, luts, pad, tag):
    import oracle as O

    for i, (s, o, (w, h), l) in enumerate(zip(srcs, d_out, shapes, luts)):
        got = o.cpu().numpy()
        want = O.i16(mode, np.ascontiguousarray(s[:, :w]), w, h, lut=l)
        assert np.array_equal(got[:, :w], want), (tag, mode, i, w, h)
        if pad:
            assert (got[:, w:] == CANARY).all(), (tag, mode, i, "padding written")


@gpu
@pytest.mark.parametrize("mode", ["fwd", "inv", "roundtrip"])
def test_batch_mixed_shapes_match_oracle(cuda, mode):
    """widths that are not multiples of 512 px (partial last tile), one-block planes, shared / distinct / no tables, pitched rows:
    the no-allocation call and the device-table batch give the oracle's planes and leave everything else alone"""
    torch = cuda
    shapes = [(1920, 64), (8, 8), (72, 24), (520, 16), (256, 24), (200, 40), (3840, 16), (768, 40)]  # the last two and (256, 24): rows ending in half a tile, tiled in pairs
    l30, l60 = _lut(30), _lut(60)
    for luts, pad in (([None] * 8, 0), ([l30, l60, l60, None, l30, _lut(10), None, l60], 24), ([l30] * 8, 8)):
        for form in ("args", "device"):
            srcs, d_in, d_out, desc = _planes(torch, shapes, luts, pad)
            if form == "args":
                api.i16_batch(mode, desc)
            else:
                b = api.Batch(mode, desc)
                assert b.launches == 1
                b.run()
                b.run()  # repeatable
                b.close()
            torch.cuda.synchronize()
            _check_against_oracle(mode, srcs, d_out, shapes, luts, pad, form)


@gpu
def test_batch_more_shapes_than_the_compare_chain_and_more_planes_than_one_argument_block(cuda):
    """> 8 different shapes: binary search over the descriptors; 130 planes with 5 tables: several launches of the
    no-allocation call (a chunk ends where tables + descriptors would outgrow the argument block), one of the device batch"""
    torch = cuda
    rng = np.random.default_rng(11)
    shapes = [(8 * int(rng.integers(1, 160)), 8 * int(rng.integers(1, 6))) for _ in range(130)]
    tabs = [_lut(s) for s in (8, 20, 50, 90, 140)]
    luts = [tabs[i % 5] if i % 7 else None for i in range(130)]
    for mode in ("fwd", "roundtrip"):
        for form in ("args", "device"):
            srcs, d_in, d_out, desc = _planes(torch, shapes, luts, 8, seed0=500)
            if form == "args":
                api.i16_batch(mode, desc)
            else:
                b = api.Batch(mode, desc)
                assert b.launches == 1
                b.run()
            torch.cuda.synchronize()
            _check_against_oracle(mode, srcs, d_out, shapes, luts, 8, form)


@gpu
def test_batch_of_equal_planes_equals_the_single_plane_calls(cuda):
    """equal shapes take the division-free plane index; 96 separately allocated 520 x 72 planes (65 blocks per row: the second
    tile of every row holds one block), all three modes, against mdct_*_i16 plane by plane"""
    torch = cuda
    shapes = [(520, 72)] * 96
    lut = _lut(25)
    for mode, single in (("fwd", api.fwd_i16), ("inv", api.inv_i16), ("roundtrip", api.roundtrip_i16)):
        for luts in ([None] * 96, [lut] * 96):
            srcs, d_in, d_out, desc = _planes(torch, shapes, luts, 0, seed0=900)
            b = api.Batch(mode, desc)
            b.run()
            for i in range(96):
                want = torch.empty_like(d_in[i])
                single(d_in[i], want, 520, 72, lut=luts[i])
                assert torch.equal(want, d_out[i]), (mode, i)
            d_out2 = [torch.full_like(t, CANARY) for t in d_out]
            api.i16_batch(mode, [(a, o, w, h, l) for (a, _, w, h, l, _, _), o in zip(desc, d_out2)])
            for a, c in zip(d_out, d_out2):
                assert torch.equal(a, c), mode


@gpu
def test_batch_is_graph_capturable(cuda):
    torch = cuda
    shapes = [(1920, 32), (960, 16), (960, 16)]
    luts = [_lut(30), _lut(60), _lut(60)]
    srcs, d_in, d_out, desc = _planes(torch, shapes, luts)
    b = api.Batch("roundtrip", desc)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.run(stream=s)  # warm
        api.i16_batch("roundtrip", desc, stream=s)
    s.synchronize()
    for o in d_out:
        o.fill_(CANARY)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        b.run(stream=s)
    g.replay()
    torch.cuda.synchronize()
    _check_against_oracle("roundtrip", srcs, d_out, shapes, luts, 0, "graph/device")
    for o in d_out:
        o.fill_(CANARY)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        api.i16_batch("roundtrip", desc, stream=s)
    g2.replay()
    torch.cuda.synchronize()
    _check_against_oracle("roundtrip", srcs, d_out, shapes, luts, 0, "graph/args")


JPEG_LUMA, JPEG_CHROMA = synth.JPEG_LUMA, synth.JPEG_CHROMA  # ITU-T T.81 Annex K.1


@gpu
def test_config3_frame_as_a_batch_equals_the_planes_call_and_the_oracle(cuda):
    """configs[2] at full size: Y 7680x4320 + Cb/Cr 3840x2160 (480 blocks per row = 7.5 tiles) with the Annex-K tables, fused
    round trip, ONE launch: the no-allocation call, the device batch and mdct_roundtrip_i16_planes agree byte for byte, each plane
    equals the single-plane entry point, and the chroma plane the threaded oracle"""
    import oracle as O

    torch = cuda
    shapes = [(7680, 4320), (3840, 2160), (3840, 2160)]
    luts = [JPEG_LUMA, JPEG_CHROMA, JPEG_CHROMA]
    d_in = [synth.plane_i16_torch(w, h, "photo", seed=synth.SEED + i) for i, (w, h) in enumerate(shapes)]
    outs = {}
    for form in ("args", "device", "planes"):
        d_out = [torch.full_like(t, CANARY) for t in d_in]
        desc = [(a, b, w, h, l) for a, b, (w, h), l in zip(d_in, d_out, shapes, luts)]
        if form == "args":
            api.i16_batch("roundtrip", desc)
        elif form == "device":
            b = api.Batch("roundtrip", desc)
            assert b.launches == 1
            b.run()
        else:
            api.roundtrip_i16_planes(desc)
        outs[form] = d_out
    for i, ((w, h), l) in enumerate(zip(shapes, luts)):
        single = torch.empty_like(d_in[i])
        api.roundtrip_i16(d_in[i], single, w, h, lut=l)
        for form in outs:
            assert torch.equal(single, outs[form][i]), (form, i)
    assert np.array_equal(outs["args"][1].cpu().numpy(), O.i16_par("roundtrip", d_in[1].cpu().numpy(), 3840, 2160, lut=JPEG_CHROMA))
    # forward / inverse batches of the same frame compose to the fused call
    coef = [torch.empty_like(t) for t in d_in]
    api.i16_batch("fwd", [(a, c, w, h, l) for a, c, (w, h), l in zip(d_in, coef, shapes, luts)])
    back = [torch.empty_like(t) for t in d_in]
    api.i16_batch("inv", [(c, b, w, h, l) for c, b, (w, h), l in zip(coef, back, shapes, luts)])
    for i in range(3):
        assert torch.equal(back[i], outs["args"][i]), i


@gpu
def test_config4_256_separately_allocated_planes_forward_in_one_call(cuda):
    """configs[3] on one GPU as the config words it: 256 INDEPENDENT 4096x4096 int16 planes (separate allocations, 8 GiB in,
    8 GiB out), forward only.  One launch of the device batch, six of the no-allocation call; both equal the stacked single
    launch (whose planes tests/test_gpu_parity.py compares with the oracle one by one), and two planes are checked against
    the oracle here as well."""
    import oracle as O

    torch = cuda
    W = H = 4096
    n = 256
    d_in = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + 100 + p) for p in range(n)]
    d_out = [torch.full((H, W), CANARY, dtype=torch.int16, device="cuda") for _ in range(n)]
    desc = [(a, b, W, H, None) for a, b in zip(d_in, d_out)]
    b = api.Batch("fwd", desc)
    assert b.launches == 1
    b.run()
    torch.cuda.synchronize()
    for p in (0, 255):
        assert np.array_equal(d_out[p].cpu().numpy(), O.i16_par("fwd", d_in[p].cpu().numpy(), W, H)), p
    want = torch.empty((H, W), dtype=torch.int16, device="cuda")
    for p in range(n):
        api.fwd_i16(d_in[p], want, W, H)
        assert torch.equal(want, d_out[p]), p
        d_out[p].fill_(CANARY)
    api.i16_batch("fwd", desc)
    for p in range(n):
        api.fwd_i16(d_in[p], want, W, H)
        assert torch.equal(want, d_out[p]), p


@gpu
def test_table_cache_first_sight_under_capture_and_beyond_its_capacity(cuda):
    """Tables are parked in device memory on first sight (mdct_api.hip: table cache; tests/test_table_cache.py) -- except inside a stream
    capture, where they travel in the kernel arguments; beyond the cache's 256 slots the least recently used table is evicted.  Same
    bytes either way: a table never seen before used first under capture (single-plane call and batch), then 300 more distinct tables."""
    import oracle as O

    torch = cuda
    W, H = 1024, 64
    src = synth.plane_i16_np(W, H, "photo", seed=77, bits=12)
    d = torch.from_numpy(src).cuda()
    rng = np.random.default_rng(2026)
    fresh = (rng.uniform(9.0, 90.0, 64)).astype(np.float32)
    out1 = torch.full_like(d, CANARY)
    out2 = torch.full_like(d, CANARY)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        api.roundtrip_i16(d, out1, W, H, lut=fresh, stream=s)
        api.i16_batch("roundtrip", [(d, out2, W, H, fresh)], stream=s)
    g.replay()
    torch.cuda.synchronize()
    want = O.i16("roundtrip", src, W, H, lut=fresh)
    assert np.array_equal(out1.cpu().numpy(), want) and np.array_equal(out2.cpu().numpy(), want)
    for k in range(300):
        q = (rng.uniform(8.5, 200.0, 64)).astype(np.float32)
        mode = ("fwd", "inv", "roundtrip")[k % 3]
        out = torch.full_like(d, CANARY)
        if k % 2:
            {"fwd": api.fwd_i16, "inv": api.inv_i16, "roundtrip": api.roundtrip_i16}[mode](d, out, W, H, lut=q)
        else:
            api.i16_batch(mode, [(d, out, W, H, q)])
        if k % 25 == 0 or k > 290:
            assert np.array_equal(out.cpu().numpy(), O.i16(mode, src, W, H, lut=q)), (k, mode)


@gpu
def test_two_host_threads_park_new_tables_and_run_batches_concurrently(cuda):
    """the table cache is shared by all host threads of a device: two threads, each on its own stream, keep introducing tables nobody has
    seen (first-sight uploads under the cache's lock) while launching single-plane calls and plane batches with them; every result is
    the oracle's"""
    import threading

    import oracle as O

    torch = cuda
    W, H = 1024, 64
    src = synth.plane_i16_np(W, H, "photo", seed=123, bits=12)
    errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            api.init(0)
            rng = np.random.default_rng(1000 + tid)
            s = torch.cuda.Stream()
            d = torch.from_numpy(src).cuda()
            for k in range(40):
                q = rng.uniform(9.0, 120.0, 64).astype(np.float32)
                mode = ("fwd", "roundtrip", "inv")[k % 3]
                out = torch.full((H, W), CANARY, dtype=torch.int16, device="cuda")
                out2 = torch.full((H, W), CANARY, dtype=torch.int16, device="cuda")
                with torch.cuda.stream(s):
                    {"fwd": api.fwd_i16, "inv": api.inv_i16, "roundtrip": api.roundtrip_i16}[mode](d, out, W, H, lut=q, stream=s)
                    api.i16_batch(mode, [(d, out2, W, H, q), (d, out, W // 2, H, None)] if k % 5 == 0 else [(d, out2, W, H, q)], stream=s)
                s.synchronize()
                want = O.i16(mode, src, W, H, lut=q)
                if not np.array_equal(out2.cpu().numpy(), want):
                    errors.append((tid, k, mode, "batch"))
                if k % 5 and not np.array_equal(out.cpu().numpy(), want):
                    errors.append((tid, k, mode, "plane"))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]
