"""Analyse gpurun_out/q32_timeline.bin (tools/experiments/exp_u8_r3 timeline): per-wave phase stamps of the q32 kernel.
Phases: load = entry -> rows arrived; compute = -> transform done + bytes staged; store = -> stores issued; ack = -> acknowledged.
Prints the distribution of each phase, the kernel span, and per SIMD how many of its resident waves are in the compute
phase over time (the VALU has work only while that number is > 0; it saturates at about 3)."""
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/q32_timeline.bin"
raw = np.fromfile(path, dtype=np.uint32).reshape(-1, 16)
hw, xcc, wt0 = raw[:, 0], raw[:, 1], raw[:, 2]
rt = raw[:, 4:10].astype(np.int64)[:, :5]
ck = raw[:, 10:16].astype(np.int64)[:, :5]
n = len(raw)
t0 = rt[:, 0].min()
rt = rt - t0  # 10 ns ticks
span = (rt[:, 4].max()) / 100.0
print(f"{n} waves; kernel span (first entry -> last store ack) {span:.2f} us")
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
x = xcc & 15
key = ((((x * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
uk = np.unique(key)
print(f"distinct SIMDs seen: {len(uk)} (expected 1024); waves per SIMD min/mean/max {np.bincount(np.searchsorted(uk, key)).min()}/{n / len(uk):.1f}/{np.bincount(np.searchsorted(uk, key)).max()}")
names = ["load (entry->rows arrived)", "compute+stage", "read back + store issue", "store ack"]
for i, nm in enumerate(names):
    d_rt = (rt[:, i + 1] - rt[:, i]) / 100.0
    d_ck = (ck[:, i + 1] - ck[:, i]) & 0xFFFFFFFF
    print(f"{nm:28s} us: p10 {np.percentile(d_rt, 10):6.2f}  p50 {np.percentile(d_rt, 50):6.2f}  p90 {np.percentile(d_rt, 90):6.2f}  mean {d_rt.mean():6.2f} | shader cycles p50 {int(np.percentile(d_ck, 50))}")
life = (rt[:, 4] - rt[:, 0]) / 100.0
print(f"wave life                    us: p10 {np.percentile(life, 10):6.2f}  p50 {np.percentile(life, 50):6.2f}  p90 {np.percentile(life, 90):6.2f}")
clk = ((ck[:, 4] - ck[:, 0]) & 0xFFFFFFFF) / np.maximum(1, rt[:, 4] - rt[:, 0]) * 100.0
print(f"shader clock over wave lives: median {np.median(clk):.0f} MHz")
# entry-time histogram: generations
print("entry times (us) deciles:", np.round(np.percentile(rt[:, 0], np.arange(0, 101, 10)) / 100.0, 2))
# per SIMD: number of waves in the compute phase over time, sampled every 0.1 us
T = int(rt[:, 4].max()) + 1
grid = np.arange(0, T, 10)
occ_hist = np.zeros(9)
res_hist = np.zeros(9)
load_hist = np.zeros(9)
for k in uk:
    m = key == k
    c0, c1 = rt[m, 1], rt[m, 2]
    incomp = ((grid[None, :] >= c0[:, None]) & (grid[None, :] < c1[:, None])).sum(0)
    resident = ((grid[None, :] >= rt[m, 0][:, None]) & (grid[None, :] < rt[m, 4][:, None])).sum(0)
    loading = ((grid[None, :] >= rt[m, 0][:, None]) & (grid[None, :] < rt[m, 1][:, None])).sum(0)
    occ_hist += np.bincount(np.minimum(incomp, 8), minlength=9)
    res_hist += np.bincount(np.minimum(resident, 8), minlength=9)
    load_hist += np.bincount(np.minimum(loading, 8), minlength=9)
for nm, h in (("waves in compute phase", occ_hist), ("waves resident", res_hist), ("waves waiting for rows", load_hist)):
    print(f"fraction of SIMD-time with k {nm:24s} k=0..8:", np.round(h / h.sum(), 3))
# a few SIMD timelines
for k in uk[:3]:
    m = np.where(key == k)[0]
    m = m[np.argsort(rt[m, 0])]
    print(f"SIMD key {k}:")
    for i in m:
        print("   wave", int(wt0[i]) >> 6, " ".join(f"{v / 100.0:7.2f}" for v in rt[i]))
