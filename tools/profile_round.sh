# the round's profiles, on the GPU box:   gpurun -- 'ROUND=6 bash tools/profile_round.sh'   (the round's final build must be in place)
#   1. kernel-trace + stats of the default bench.py (the program itself after `--`, no wrappers)
#   2. four separate --pmc passes (FETCH_SIZE / WRITE_SIZE / two SQ sets: they do not fit one pass, and --pmc is never combined
#      with a trace) of tools/run_kernel.py all: EVERY kernel a bench block names, the 256-plane batch at its real size
#   3. tools/valubench2: issue cost per instruction class (the constants of bench.py's vector-issue floor)
#   4. tools/pmc_round.py: gpurun_out/rNNprof/{rNN_pmc_raw.txt, traffic.json}  -> tools/collect_round_profiles.sh copies them to profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-6}
TAG=$(printf 'r%02d' $ROUND)
O=$R/gpurun_out/${TAG}prof
mkdir -p $O
N=${LAUNCHES:-6}
if [ -z "$SKIP_TRACE" ]; then
  rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_under_trace.log 2> $O/bench_under_trace.err
  echo "trace rc=$?"
fi
if [ -z "$SKIP_PMC" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_$c -o run --output-format csv -- python3 $R/tools/run_kernel.py ${KERNELS:-all} $N > $O/pmc_$c.log 2>&1
    echo "pmc $c rc=$?"
  done
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_SQ -o run --output-format csv -- python3 $R/tools/run_kernel.py ${KERNELS:-all} $N > $O/pmc_SQ.log 2>&1
  echo "pmc SQ rc=$?"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d $O/pmc_SQ2 -o run --output-format csv -- python3 $R/tools/run_kernel.py ${KERNELS:-all} $N > $O/pmc_SQ2.log 2>&1
  echo "pmc SQ2 rc=$?"
fi
cd $R
[ -x tools/valubench2 ] && tools/valubench2 > $O/${TAG}_valubench2.log 2>&1
python3 tools/pmc_round.py $O $O $ROUND > $O/pmc_summary.txt 2>&1
[ -z "$SKIP_TRACE" ] && python3 tools/trace_steady.py $O/trace > $O/trace_steady.txt 2>&1
cat $O/pmc_summary.txt
[ -z "$SKIP_TRACE" ] && tail -40 $O/trace_steady.txt
grep VALU_ISSUE $O/${TAG}_valubench2.log
