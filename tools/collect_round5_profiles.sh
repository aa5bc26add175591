# copies what tools/profile_round5.sh (and the bench runs beside it) left under gpurun_out/ into profiles/ under the names profiles/README.md lists
set -e
O=gpurun_out/r05prof
cp $O/r05_pmc_raw.txt profiles/r05_pmc_raw.txt
cp $O/traffic.json profiles/traffic.json
cp $O/r05_valubench2.log profiles/r05_valubench2.log
cp $O/bench_under_trace.log profiles/r05_b_bench_under_kernel_trace.log
cp $O/trace_steady.txt profiles/r05_b_steady_state_from_kernel_trace.txt
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/r05_b_kernel_stats_bench_with_extras.csv
[ -f gpurun_out/r05_e_bench_default.log ] && cp gpurun_out/r05_e_bench_default.log profiles/r05_e_bench_default.log
[ -f gpurun_out/r05_e_bench_driver_form_steps20.log ] && cp gpurun_out/r05_e_bench_driver_form_steps20.log profiles/r05_e_bench_driver_form_steps20.log
python3 - <<'PY'
import json
ns = cyc = None
for line in open("profiles/r05_valubench2.log"):
    if line.startswith("VALU_ISSUE_COSTS_NS"):
        ns = json.loads(line.split(" ", 1)[1])
    if line.startswith("VALU_ISSUE_COSTS_CYCLES"):
        cyc = json.loads(line.split(" ", 1)[1])
if ns and cyc:
    old = json.load(open("profiles/valu_issue_costs.json"))
    new = dict(old, ns={k: ns[k] for k in ("plain", "packed", "other")}, cycles={k: cyc[k] for k in ("plain", "packed", "other")})
    json.dump(new, open("profiles/valu_issue_costs.json", "w"), indent=1)
    print("issue cycles", new["cycles"], "(were", old["cycles"], ")")
PY
