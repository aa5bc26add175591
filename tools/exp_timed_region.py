"""Where the time of a SHORT timed region goes (bench.py at the driver's --steps 20): K launches of the bench workload after a device
synchronisation, clocked by the host (t0 .. stop event seen by polling) and by the stream's events, for several K and for two ways of
going idle before t0 (torch.cuda.synchronize(), or polling an event).   python3 tools/exp_timed_region.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simd_dct_amd as M
from simd_dct_amd import synth

M.init(0)
W = H = 8192
srcs = [synth.plane_i16_torch(W, H, "photo", seed=synth.SEED + i) for i in range(4)]
dsts = [torch.empty_like(s) for s in srcs]
steps = [M.prepare_plane_i16("roundtrip", srcs[i], dsts[i], W, H) for i in range(4)]
lib = M.api._lib.load()
st = M.api._stream()
tm, pre = M.Timer(), M.Timer()
for i in range(1500):
    steps[i % 4]()
torch.cuda.synchronize()
for idle in ("torch.cuda.synchronize()", "event polled", "synchronize x2 (round 4's barrier)"):
    for K in (1, 2, 5, 10, 20, 50, 200, 2000):
        walls, evs, tails = [], [], []
        for rep in range(7):
            for i in range(300):
                steps[i % 4]()
            if idle.startswith("event"):
                lib.mdct_timer_stop(pre._t, st)
                pre.wait_spin()
            else:
                torch.cuda.synchronize()
                if "x2" in idle:
                    torch.cuda.synchronize()
            lib.mdct_timer_start(tm._t, st)
            t0 = time.perf_counter()
            for i in range(K):
                steps[i % 4]()
            lib.mdct_timer_stop(tm._t, st)
            tm.wait_spin()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            walls.append((t1 - t0) * 1e6)
            tails.append((t2 - t1) * 1e6)
            evs.append(tm.elapsed_ms() * 1e3)
        walls.sort(); evs.sort(); tails.sort()
        print(f"idle by {idle:34s} K={K:5d}  host clock {walls[3]/K:8.2f} us/step (total {walls[3]:9.1f})   events {evs[3]/K:8.2f} us/step (total {evs[3]:9.1f})   "
              f"host - events {walls[3]-evs[3]:7.1f} us   synchronize() after the poll {tails[3]:6.1f} us", flush=True)
