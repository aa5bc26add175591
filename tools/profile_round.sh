cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_under_trace.log 2> $O/bench_under_trace.err
echo "trace rc=$?"
for k in q32 roundtrip; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_${k}_$c -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 12 > $O/pmc_${k}_$c.log 2>&1
    echo "pmc $k $c rc=$?"
  done
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_${k}_SQ -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 12 > $O/pmc_${k}_SQ.log 2>&1
  echo "pmc $k SQ rc=$?"
done
cd $R
for d in $O/pmc_*/; do echo "== $d"; python3 tools/pmc_agg.py $d mdct; done > $O/pmc_summary.txt 2>&1
ls $O/trace | head; 
python3 - <<'PY'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/r02prof/trace/*kernel_stats.csv'):
    print(open(f).read()[:3000])
PY
cat $O/pmc_summary.txt
tail -c 1500 $O/bench_under_trace.log
