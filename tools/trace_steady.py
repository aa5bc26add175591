"""Steady-state duration per kernel from a rocprofv3 --kernel-trace run of bench.py:
    python3 tools/trace_steady.py <trace dir>
bench.py times every kernel after untimed pre-conditioning launches, back to back: for each kernel name the LAST run of
consecutive dispatches is what matters; the LONGEST such run of a kernel is its pre-conditioning + warm-up + timed region; the statistics below are over the last
`n_timed` dispatches of that run (2000 for the headline kernel, 500 for the u8 products, 100-200 for the others), so the
from-idle power transient and the verification launches are outside."""
import csv
import glob
import os
import statistics as st
import sys

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive runs of one kernel name
runs = []
for r in rows:
    k = r["Kernel_Name"]
    if runs and runs[-1][0] == k:
        runs[-1][1].append(r)
    else:
        runs.append([k, [r]])
last_run = {}
for k, rs in runs:  # the longest back-to-back run of each kernel = its pre-conditioning + warm-up + timed region
    if "mdct::" in k and len(rs) >= 8 and len(rs) > len(last_run.get(k, [])):
        last_run[k] = rs
print(f"{'kernel':88s} {'n':>5s} {'mean ns':>9s} {'median':>8s} {'min':>8s} {'gap':>6s}")
for k, rs in sorted(last_run.items()):
    n = 2000 if len(rs) >= 2200 else (500 if len(rs) >= 1500 else (max(40, len(rs) * 2 // 5) if len(rs) >= 100 else max(3, len(rs) * 2 // 3)))
    sel = rs[-n:]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    gaps = [int(sel[i + 1]["Start_Timestamp"]) - int(sel[i]["End_Timestamp"]) for i in range(len(sel) - 1)]
    print(f"{k[:88]:88s} {n:5d} {st.mean(d):9.0f} {st.median(d):8.0f} {min(d):8d} {st.mean(gaps):6.0f}")
