"""The table cache of the int16 / 8-bit entry points (csrc/mdct_api.hip, contract in include/mdct.h "Tables"): quantiser tables are parked
in device memory by a one-wave upload kernel on the CALLER'S stream -- no host block, no synchronisation --, 256 per device, least
recently used evicted, never touched by a capturing stream.  (Round 4's cache did a blocking 512-byte hipMemcpy on first sight, never
evicted and fell back to the kernel arguments for good after 256 tables: VERDICT r4 weak #8.)

CPU: the source holds no blocking copy on a launch path.  GPU: 1,000 distinct tables back to back (all device-resident, nothing blocks,
results exact and reproducible after eviction), three streams sharing and evicting each other's tables, a captured launch that keeps
working from its arguments whatever the cache does later."""
import os
import re
import threading
import time

import numpy as np
import pytest

from simd_dct_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANARY = -21846


def test_no_blocking_copy_on_a_launch_path():
    src = open(os.path.join(ROOT, "simd_dct_amd", "csrc", "mdct_api.hip")).read()
    cache = src[src.index("// ---- table cache"):src.index("int run_i16(")]
    code = "\n".join(l.split("//")[0] for l in cache.splitlines())
    assert "hipMemcpy" not in code and "hipDeviceSynchronize" not in code
    assert len(re.findall(r"hipStreamSynchronize", code)) == 1  # the event-record error path only
    assert "launch_park_table" in code and "hipStreamWaitEvent" in code
    hdr = open(os.path.join(ROOT, "include", "mdct.h")).read()
    assert "blocking 512-byte copy" not in hdr and "never evicted" not in hdr


gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    torch.cuda.set_device(0)
    api.init(0)
    return torch


def _delta(a, b):
    return {k: b[k] - a[k] for k in a}


@gpu
def test_thousand_distinct_tables_back_to_back(cuda):
    """every launch reads a device-resident table (none falls back to the kernel arguments), the cache evicts, and the same table met again
    after its eviction gives the same bytes; a sample against the oracle"""
    import oracle as O

    torch = cuda
    W, H = 512, 64
    src = synth.plane_i16_np(W, H, "photo", seed=31, bits=12)
    d = torch.from_numpy(src).cuda()
    rng = np.random.default_rng(5)
    luts = [rng.uniform(8.5, 150.0, 64).astype(np.float32) for _ in range(1000)]
    modes = [("fwd", api.fwd_i16), ("inv", api.inv_i16), ("roundtrip", api.roundtrip_i16)]
    first = torch.full((1000, H, W), CANARY, dtype=torch.int16, device="cuda")
    s0 = api.table_cache_stats()
    for k, q in enumerate(luts):
        modes[k % 3][1](d, first[k], W, H, lut=q)
    s1 = api.table_cache_stats()
    dl = _delta(s0, s1)
    assert dl["uploads"] == 1000 and dl["hits"] == 0 and dl["from_arguments"] == 0, dl  # 1000 launches, 1000 device-resident tables
    assert dl["evictions"] >= 1000 - 256, dl
    again = torch.full((1000, H, W), CANARY, dtype=torch.int16, device="cuda")
    for k in reversed(range(1000)):  # the last 256 are still resident (hits), the rest are uploaded again
        if k % 2:
            modes[k % 3][1](d, again[k], W, H, lut=luts[k])
        else:
            api.i16_batch(modes[k % 3][0], [(d, again[k], W, H, luts[k])])
    torch.cuda.synchronize()
    s2 = api.table_cache_stats()
    dl = _delta(s1, s2)
    assert dl["from_arguments"] == 0 and dl["hits"] + dl["uploads"] == 1000 and dl["hits"] >= 128, dl
    assert torch.equal(first, again)
    for k in (0, 1, 2, 255, 256, 257, 600, 997, 998, 999):
        assert np.array_equal(first[k].cpu().numpy(), O.i16(modes[k % 3][0], src, W, H, lut=luts[k])), k


@gpu
def test_first_sight_does_not_block_the_host(cuda):
    """400 launches of the 8192^2 round trip (~45 us each on the GPU), each with a table nobody has seen: the host finishes enqueueing
    while most of them are still queued -- a blocking first-sight copy would hold the host to the GPU's pace"""
    torch = cuda
    W = H = 8192
    n = 400
    src = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(2)]
    dst = [torch.empty_like(t) for t in src]
    rng = np.random.default_rng(17)
    luts = [rng.uniform(9.0, 99.0, 64).astype(np.float32) for _ in range(n)]
    calls = [api.prepare_plane_i16("roundtrip", src[k % 2], dst[k % 2], W, H, lut=luts[k]) for k in range(n)]
    for k in range(20):  # warm: code objects, clocks (20 of the tables are resident afterwards)
        calls[k]()
    torch.cuda.synchronize()
    s0 = api.table_cache_stats()
    done = torch.cuda.Event()
    t0 = time.perf_counter()
    for k in range(20, n):
        calls[k]()
    host_s = time.perf_counter() - t0
    done.record()
    still_running = not done.query()
    torch.cuda.synchronize()
    gpu_s = time.perf_counter() - t0
    dl = _delta(s0, api.table_cache_stats())
    assert dl["uploads"] == n - 20 and dl["from_arguments"] == 0, dl
    assert still_running and host_s < 0.6 * gpu_s, (host_s, gpu_s)


@gpu
def test_three_streams_share_and_evict_tables(cuda):
    """three host threads, each on its own stream, cycle through the same 300 tables (more than the cache holds) in different orders: a
    stream meets tables another stream is uploading (waits on the device) and evicts tables another stream's launches still read
    (the upload queues behind them); every output equals the one a device-resident batch -- which owns its tables -- computed"""
    torch = cuda
    W, H = 1024, 32
    src = synth.plane_i16_np(W, H, "photo", seed=88, bits=12)
    d = torch.from_numpy(src).cuda()
    rng = np.random.default_rng(23)
    luts = [rng.uniform(8.5, 120.0, 64).astype(np.float32) for _ in range(300)]
    want = torch.empty((300, H, W), dtype=torch.int16, device="cuda")
    for k, q in enumerate(luts):
        b = api.Batch("roundtrip", [(d, want[k], W, H, q)])
        b.run()
        torch.cuda.synchronize()
        b.close()
    errors = []
    s0 = api.table_cache_stats()

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            api.init(0)
            s = torch.cuda.Stream()
            order = np.random.default_rng(100 + tid).permutation(300) if tid else np.arange(300)
            outs = torch.full((300, H, W), CANARY, dtype=torch.int16, device="cuda")
            s.wait_stream(torch.cuda.default_stream())
            for rep in range(2):
                for k in order:
                    api.roundtrip_i16(d, outs[k], W, H, lut=luts[k], stream=s)
            s.synchronize()
            bad = [int(k) for k in range(300) if not torch.equal(outs[k], want[k])]
            if bad:
                errors.append((tid, bad[:8]))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    dl = _delta(s0, api.table_cache_stats())
    assert dl["from_arguments"] == 0 and dl["hits"] + dl["uploads"] == 3 * 600 and dl["evictions"] > 0, dl


@gpu
def test_captured_launch_never_depends_on_the_cache(cuda):
    """a table that IS parked, used under capture: the captured launch carries the table in its arguments (the cache counts it), and its
    replay is right after 300 other tables have gone through the slot it would have pointed at"""
    import oracle as O

    torch = cuda
    W, H = 1024, 64
    src = synth.plane_i16_np(W, H, "photo", seed=9, bits=12)
    d = torch.from_numpy(src).cuda()
    rng = np.random.default_rng(41)
    q = rng.uniform(9.0, 90.0, 64).astype(np.float32)
    out = torch.full_like(d, CANARY)
    api.roundtrip_i16(d, out, W, H, lut=q)  # parks q
    torch.cuda.synchronize()
    want = O.i16("roundtrip", src, W, H, lut=q)
    assert np.array_equal(out.cpu().numpy(), want)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    s0 = api.table_cache_stats()
    g = torch.cuda.CUDAGraph()
    out_g = torch.full_like(d, CANARY)
    with torch.cuda.graph(g, stream=s):
        api.roundtrip_i16(d, out_g, W, H, lut=q, stream=s)
    dl = _delta(s0, api.table_cache_stats())
    assert dl["from_arguments"] == 1 and dl["hits"] == 0 and dl["uploads"] == 0, dl
    for k in range(300):  # evict everything that was resident
        api.fwd_i16(d, out, W, H, lut=rng.uniform(8.5, 200.0, 64).astype(np.float32))
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out_g.cpu().numpy(), want)


@gpu
def test_seventh_stream_reading_a_many_reader_table_is_fenced_before_eviction(cuda):
    """A table read by more streams than a slot remembers (kSlotStreams = 4) is fenced against the cache-wide stream list, so EVERY stream
    that launches a read must be on that list -- also one whose reads only ever met such slots (round 5 returned early for them: ADVICE r5).
    Stream 6 queues 25 ms of other work, then a launch that reads the shared table; stream 0 then pushes 300 new tables through the cache,
    evicting the shared one: its upload must queue behind stream 6's read, or that read sees somebody else's multipliers."""
    torch = cuda
    W, H = 1024, 64
    src = synth.plane_i16_np(W, H, "photo", seed=5, bits=12)
    d = torch.from_numpy(src).cuda()
    rng = np.random.default_rng(77)
    shared = rng.uniform(9.0, 80.0, 64).astype(np.float32)
    want = torch.empty_like(d)
    b = api.Batch("roundtrip", [(d, want, W, H, shared)])  # owns its table: independent of the cache
    b.run()
    torch.cuda.synchronize()
    b.close()
    big = synth.plane_i16_torch(8192, 8192, "photo", seed=1)
    big_out = torch.empty_like(big)
    streams = [torch.cuda.Stream() for _ in range(7)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    s0 = api.table_cache_stats()
    outs = [torch.full_like(d, CANARY) for _ in range(7)]
    for i in range(1, 6):  # five reader streams: one more than the slot remembers
        api.roundtrip_i16(d, outs[i], W, H, lut=shared, stream=streams[i].cuda_stream)
    blocker = api.prepare_plane_i16("roundtrip", big, big_out, 8192, 8192, stream=streams[6].cuda_stream)
    for _ in range(600):  # ~25 ms of queue in front of stream 6's read
        blocker()
    api.roundtrip_i16(d, outs[6], W, H, lut=shared, stream=streams[6].cuda_stream)
    scratch = torch.empty_like(d)
    for k in range(300):  # stream 0 evicts everything, the shared table included
        api.fwd_i16(d, scratch, W, H, lut=rng.uniform(8.5, 200.0, 64).astype(np.float32), stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
    dl = _delta(s0, api.table_cache_stats())
    assert dl["uploads"] == 301 and dl["from_arguments"] == 0 and dl["unfenceable"] == 0, dl
    for i in range(1, 7):
        assert torch.equal(outs[i], want), i
