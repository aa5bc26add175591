# round-4 profiles (run on the GPU box: gpurun -- 'bash tools/profile_round4.sh'):
#   kernel-trace + stats of the default bench (steady-state extraction per kernel by tools/trace_steady.py)
#   separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) per kernel through tools/run_kernel.py -- never combined with a trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04prof
mkdir -p $O
if [ -z "$SKIP_TRACE" ]; then
rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_under_trace.log 2> $O/bench_under_trace.err
echo "trace rc=$?"
fi
for k in ${KERNELS:-frame420 batch32 f32 roundtrip roundtrip_lut stereo_scalar encq_scalar stereo_sse q32}; do
  n=12
  [ $k = batch32 ] && n=4
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $O/pmc_${k}_$c -o run --output-format csv -- python3 $R/tools/run_kernel.py $k $n > $O/pmc_${k}_$c.log 2>&1
    echo "pmc $k $c rc=$?"
  done
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_${k}_SQ -o run --output-format csv -- python3 $R/tools/run_kernel.py $k $n > $O/pmc_${k}_SQ.log 2>&1
  echo "pmc $k SQ rc=$?"
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d $O/pmc_${k}_SQ2 -o run --output-format csv -- python3 $R/tools/run_kernel.py $k $n > $O/pmc_${k}_SQ2.log 2>&1
  echo "pmc $k SQ2 rc=$?"
done
cd $R
for d in $O/pmc_*/; do echo "== $(basename $d)"; python3 tools/pmc_agg.py $d mdct; done > $O/pmc_summary.txt 2>&1
python3 tools/trace_steady.py $O/trace > $O/trace_steady.txt 2>&1
cat $O/trace_steady.txt
tail -c 1500 $O/pmc_summary.txt
