"""Per-kernel statistics from a rocprofv3 --kernel-trace CSV, optionally for the LAST n dispatches
of one kernel only (= bench.py's timed region, which ends the run when --no-extras is given).
   python3 tools/trace_stats.py <kernel_trace.csv> [substring] [last_n]"""
import csv, statistics as st, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "mdct::"
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sel = [r for r in rows if sub in r["Kernel_Name"]]
if last:
    sel = sel[-last:]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
gaps = [int(sel[i + 1]["Start_Timestamp"]) - int(sel[i]["End_Timestamp"]) for i in range(len(sel) - 1)]
print(f"kernel filter '{sub}', dispatches {len(d)}" + (f" (last {last})" if last else ""))
print(f"duration ns: mean {st.mean(d):.0f}  median {st.median(d):.0f}  min {min(d)}  max {max(d)}  stdev {st.pstdev(d):.0f}")
if gaps:
    print(f"gap between consecutive dispatches ns: mean {st.mean(gaps):.0f}  median {st.median(gaps):.0f}")
    print(f"mean duration + mean gap = {st.mean(d) + st.mean(gaps):.0f} ns per step")
