# ROUND=6 bash tools/collect_round_profiles.sh -- copies what tools/profile_round.sh (and the bench runs beside it) left under gpurun_out/ into profiles/ under the names profiles/README.md lists
set -e
ROUND=${ROUND:-6}
TAG=$(printf 'r%02d' $ROUND)
O=gpurun_out/${TAG}prof
cp $O/${TAG}_pmc_raw.txt profiles/${TAG}_pmc_raw.txt
cp $O/traffic.json profiles/traffic.json
cp $O/${TAG}_valubench2.log profiles/${TAG}_valubench2.log
cp $O/bench_under_trace.log profiles/${TAG}_b_bench_under_kernel_trace.log
cp $O/trace_steady.txt profiles/${TAG}_b_steady_state_from_kernel_trace.txt
f=$(ls $O/trace/*kernel_stats.csv $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/${TAG}_b_kernel_stats_bench_with_extras.csv
[ -f gpurun_out/${TAG}_e_bench_default.log ] && cp gpurun_out/${TAG}_e_bench_default.log profiles/${TAG}_e_bench_default.log
[ -f gpurun_out/${TAG}_e_bench_driver_form_steps20.log ] && cp gpurun_out/${TAG}_e_bench_driver_form_steps20.log profiles/${TAG}_e_bench_driver_form_steps20.log
python3 - <<PY
import json
ns = cyc = None
for line in open("profiles/${TAG}_valubench2.log"):
    if line.startswith("VALU_ISSUE_COSTS_NS"):
        ns = json.loads(line.split(" ", 1)[1])
    if line.startswith("VALU_ISSUE_COSTS_CYCLES"):
        cyc = json.loads(line.split(" ", 1)[1])
if ns and cyc:
    old = json.load(open("profiles/valu_issue_costs.json"))
    new = dict(old, ns={k: ns[k] for k in ("plain", "packed", "other")}, cycles={k: cyc[k] for k in ("plain", "packed", "other")})
    new["source"] = old["source"].replace("r05_valubench2.log", "${TAG}_valubench2.log").replace("round 5", "round ${ROUND}").replace("profile_round5.sh", "profile_round.sh")
    json.dump(new, open("profiles/valu_issue_costs.json", "w"), indent=1)
    print("issue cycles", new["cycles"], "(were", old["cycles"], ")")
PY
