"""bench_cpu.py -- the `cpu_baseline` leg of bench.py: the CHECKER timed on the GPU box's host cores.

Two figures (BASELINE.md 3 / SURVEY.md 8d), both on a bounded sample so that the default bench run stays short:
  * `value`: oracle/dct_oracle.c's orc_roundtrip_i16 (a scalar C port of the bench workload, kind "port") on all host threads;
  * `reference_q32`: the REAL reference's q32 / AVX2 tier (oracle/_ref, kind "reference"; the restatement orc_q32_avx when
    oracle/_ref did not travel) on ONE pinned core and on ALL host threads pinned to distinct CPUs over disjoint
    startY/endY ranges -- timed natively by oracle/time_mt.c, 3 warm-ups + 40 runs, median and inter-quartile range
    (min-time kept for continuity with print_perf_info, main.cpp:34-80).
This is the only file besides tests/ and __graft_entry__.smoke() that touches oracle/; it is a reported baseline, never the product.
"""
import ctypes
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
W = H = 8192


def host_share():
    """what this process may really use: the affinity mask and the cgroup CPU quota"""
    cpus = sorted(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = round(int(q) / int(period), 2)
    except Exception:
        pass
    threads = len(cpus) if quota is None else min(len(cpus), max(1, int(quota)))
    return cpus, quota, max(1, min(threads, 64))


def _quartiles(samples):
    q = statistics.quantiles(samples, n=4) if len(samples) >= 4 else [min(samples), statistics.median(samples), max(samples)]
    return q[0], q[1], q[2]


def summarise(seconds, px):
    """Mpx/s statistics of a list of per-run wall times"""
    rates = sorted(px / s / 1e6 for s in seconds)
    q1, med, q3 = _quartiles(rates)
    return {"median": round(med, 1), "iqr": round(q3 - q1, 1), "min_time": round(rates[-1], 1), "runs": len(rates)}


def reference_q32_baseline(runs=40, warmups=3):
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from simd_dct_amd import synth
    from simd_dct_amd.api import QUANTIZE_BASE

    cpus, quota, threads = host_share()
    lib = O.oracle()
    ref = O.reference()
    if ref is not None:
        fn, which = ctypes.cast(ref.ref_call_tier, ctypes.c_void_p), O.REF_FUNCS["q32_avx"][1]
    else:
        fn, which = ctypes.cast(lib.orc_q32_avx, ctypes.c_void_p), -1
    img = np.ascontiguousarray(synth.plane_u8_np(W, H, "photo").reshape(-1))
    lut = np.ascontiguousarray((QUANTIZE_BASE * np.float32(2000)).astype(np.float32))
    dst = np.zeros(W * H, dtype=np.uint8)
    pin = (ctypes.c_int * len(cpus))(*cpus)

    def timed(nthreads):
        sec = (ctypes.c_double * runs)()
        rc = lib.orc_time_q32_mt(fn, which, img.ctypes.data, dst.ctypes.data, lut.ctypes.data_as(O.f32p), W, H, nthreads, pin, len(cpus), warmups, runs, sec)
        if rc < 0:
            return {"error": f"orc_time_q32_mt returned {rc}"}
        out = summarise(list(sec), W * H)
        out["threads"], out["pinned"] = nthreads, rc
        return out

    try:
        cpu_model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    except OSError:
        cpu_model = "?"
    return {"kind": "reference" if ref is not None else "port", "unit": "Mpixels/s", "cpu": cpu_model, "affinity_cpus": len(cpus), "cgroup_cpu_quota": quota,
            "one_pinned_core": timed(1), "all_host_threads": timed(threads)}


def cpu_baseline(budget_s=12.0):
    """oracle/ (the CPU port) on a bounded sample of the bench workload, all host threads; + the reference's q32 product"""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from simd_dct_amd import synth

    _, _, threads = host_share()
    rows = 64  # 8192 x 64 px stripe = 0.5 Mpx per call
    src = synth.plane_i16_np(W, rows, "photo")
    bufs = [(src.copy(), np.empty_like(src)) for _ in range(threads)]
    O.i16("roundtrip", src, W, rows, out=bufs[0][1])  # warm (loads the checker)
    counts = [0] * threads
    deadline = time.perf_counter() + budget_s

    def work(i):
        a, b = bufs[i]
        while time.perf_counter() < deadline:  # bounded by time, whatever the host's core share is
            O.i16("roundtrip", a, W, rows, out=b)
            counts[i] += 1

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    px = sum(counts) * W * rows
    assert np.array_equal(bufs[0][0], bufs[0][1])
    return {"value": round(px / dt / 1e6, 1), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": f"orc_roundtrip_i16 (scalar C, -O2 -ffp-contract=off), {threads} threads, {px / 1e6:.0f} Mpx of {W}x{rows} int16 stripes in {dt:.1f} s",
            "reference_q32": reference_q32_baseline()}


if __name__ == "__main__":
    import json

    sys.path.insert(0, ROOT)
    print(json.dumps(cpu_baseline(float(sys.argv[1]) if len(sys.argv) > 1 else 12.0), indent=1))
