import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simd_dct_amd as M
from simd_dct_amd import synth
M.init(0)
jpeg = (M.QUANTIZE_BASE * np.float32(100)).astype(np.float32)
def mk(w,h,s): 
    a = synth.plane_i16_torch(w,h,"photo",seed=s); return a, torch.empty_like(a)
t = M.Timer()
def run(name, calls, px):
    for i in range(300): calls[i % len(calls)]()
    r=[]
    for k in range(5):
        t.start()
        for i in range(40): calls[i % len(calls)]()
        t.stop(); r.append(t.elapsed_ms()/40)
    r.sort(); ms=r[2]
    print(f"{name:44s} {ms*1e3:8.2f} us  {4*px/(ms*1e-3)/1e9:8.1f} GB/s")
big=[mk(8192,8192,i) for i in range(4)]
run("single-plane API 8192^2 +table", [M.prepare_plane_i16("roundtrip",a,b,8192,8192,lut=jpeg) for a,b in big], 8192*8192)
run("planes API, 1 x 8192^2 +table", [M.prepare_roundtrip_i16_planes([(a,b,8192,8192,jpeg)]) for a,b in big], 8192*8192)
run("planes API, 1 x 8192^2 no table", [M.prepare_roundtrip_i16_planes([(a,b,8192,8192,None)]) for a,b in big], 8192*8192)
Ys=[mk(7680,4320,i) for i in range(4)]; Cbs=[mk(3840,2160,10+i) for i in range(4)]; Crs=[mk(3840,2160,20+i) for i in range(4)]
fpx=7680*4320+2*3840*2160
run("planes API, 4:2:0 frame +tables (1 launch)", [M.prepare_roundtrip_i16_planes([(Ys[i][0],Ys[i][1],7680,4320,jpeg),(Cbs[i][0],Cbs[i][1],3840,2160,jpeg),(Crs[i][0],Crs[i][1],3840,2160,jpeg)]) for i in range(4)], fpx)
def three(i):
    c=[M.prepare_plane_i16("roundtrip",Ys[i][0],Ys[i][1],7680,4320,lut=jpeg),M.prepare_plane_i16("roundtrip",Cbs[i][0],Cbs[i][1],3840,2160,lut=jpeg),M.prepare_plane_i16("roundtrip",Crs[i][0],Crs[i][1],3840,2160,lut=jpeg)]
    return lambda: [x() for x in c]
run("single-plane API x3 (3 launches)", [three(i) for i in range(4)], fpx)
run("single-plane API Y only 7680x4320", [M.prepare_plane_i16("roundtrip",Ys[i][0],Ys[i][1],7680,4320,lut=jpeg) for i in range(4)], 7680*4320)
