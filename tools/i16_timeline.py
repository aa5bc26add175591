"""Analyse gpurun_out/i16_timeline.bin (tools/experiments/exp_u8_r3 timeline_i16): entry / stores issued / stores acknowledged per wave of the
fused int16 round trip.  Prints the kernel span, per-SIMD wave counts and finish times, the life of a wave, how many waves
are resident per SIMD over time and the rate at which waves complete over the kernel (fill, steady state, drain)."""
import sys

import numpy as np

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/i16_timeline.bin"
raw = np.fromfile(path, dtype=np.uint32).reshape(-1, 16)
hw, xcc = raw[:, 0], raw[:, 1]
rt = raw[:, 4:9].astype(np.int64)
ck = raw[:, 10:15].astype(np.int64)
rt -= rt[:, 0].min()
n = len(raw)
print(f"{n} waves; kernel span {rt[:, 4].max() / 100:.2f} us; last entry {rt[:, 0].max() / 100:.2f} us")
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; x = xcc & 15
key = ((((x * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
uk = np.unique(key)
cnt = np.array([(key == k).sum() for k in uk])
fin = np.array([rt[key == k, 4].max() for k in uk]) / 100
print(f"SIMDs {len(uk)}; waves per SIMD min/mean/max {cnt.min()}/{cnt.mean():.1f}/{cnt.max()}; per-SIMD finish p10 {np.percentile(fin, 10):.2f} p50 {np.percentile(fin, 50):.2f} p90 {np.percentile(fin, 90):.2f} max {fin.max():.2f}")
life = (rt[:, 3] - rt[:, 0]) / 100
ack = (rt[:, 4] - rt[:, 3]) / 100
print(f"entry -> stores issued us: p10 {np.percentile(life, 10):.2f} p50 {np.percentile(life, 50):.2f} p90 {np.percentile(life, 90):.2f}; store ack p50 {np.percentile(ack, 50):.2f}")
clk = ((ck[:, 4] - ck[:, 0]) & 0xFFFFFFFF) / np.maximum(1, rt[:, 4] - rt[:, 0]) * 100.0
print(f"shader clock over wave lives: median {np.median(clk):.0f} MHz")
# lives by entry time (first generation vs later)
first = rt[:, 0] < 100
print(f"first-generation waves ({first.sum()}): life p50 {np.percentile(life[first], 50):.2f} p90 {np.percentile(life[first], 90):.2f}; later waves: p50 {np.percentile(life[~first], 50):.2f}")
# completions per microsecond
T = int(rt[:, 4].max() / 100) + 1
done = np.bincount((rt[:, 3] // 100).astype(int), minlength=T)
ent = np.bincount((rt[:, 0] // 100).astype(int), minlength=T)
print("us     :", " ".join(f"{i:4d}" for i in range(T)))
print("entered:", " ".join(f"{v:4d}" for v in ent[:T]))
print("stored :", " ".join(f"{v:4d}" for v in done[:T]))
print(f"steady-state completions per us (middle half): {np.mean(done[T // 4: 3 * T // 4]):.0f} waves = {np.mean(done[T // 4: 3 * T // 4]) * 16384 / 1e6:.2f} TB/s of 16 KiB tiles")
res_hist = np.zeros(9)
grid = np.arange(0, int(rt[:, 4].max()) + 1, 10)
for k in uk[::8]:
    m = key == k
    resident = ((grid[None, :] >= rt[m, 0][:, None]) & (grid[None, :] < rt[m, 4][:, None])).sum(0)
    res_hist += np.bincount(np.minimum(resident, 8), minlength=9)
print("fraction of SIMD-time with k waves resident k=0..8:", np.round(res_hist / res_hist.sum(), 3))
k = uk[0]
m = np.where(key == k)[0]; m = m[np.argsort(rt[m, 0])]
print("one SIMD: (entry, stores issued, ack) us")
for i in m:
    print("   ", " ".join(f"{rt[i, j] / 100:7.2f}" for j in (0, 3, 4)))
