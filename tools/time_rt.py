"""The bench workload alone (8192^2 int16 fused round trip, 4 rotating plane pairs), steady state; A/B of library builds:
   MDCT_LIB_PATH=build_variants/x.so python3 tools/time_rt.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
W = H = 8192
jpeg = (M.QUANTIZE_BASE * np.float32(100)).astype(np.float32)
pl = [synth.plane_i16_torch(W, H, "photo", seed=i) for i in range(4)]
pl = [(a, torch.empty_like(a)) for a in pl]
t = M.Timer()
def run(name, calls, reps=200):
    for i in range(1200): calls[i % 4]()
    r = []
    for k in range(7):
        t.start()
        for i in range(reps): calls[i % 4]()
        t.stop(); r.append(t.elapsed_ms() / reps)
    r.sort()
    print(f"{os.path.basename(os.environ.get('MDCT_LIB_PATH', 'default')):18s} {name:28s} median {r[3]*1e3:7.2f} us  min {r[0]*1e3:7.2f}  {4*W*H/(r[3]*1e-3)/8e12:6.3f} of 8 TB/s", flush=True)
run("roundtrip", [M.prepare_plane_i16("roundtrip", a, b, W, H) for a, b in pl])
ok = all(torch.equal(a, b) for a, b in pl)
run("roundtrip + table", [M.prepare_plane_i16("roundtrip", a, b, W, H, lut=jpeg) for a, b in pl])
run("stream copy", [M.prepare_stream_copy(a, b, W * H * 2) for a, b in pl])
print("bit-exact round trip:", ok)
