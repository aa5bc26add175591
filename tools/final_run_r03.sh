set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_tests.log 2>&1 || { tail -20 gpurun_out/r03_final_tests.log; exit 1; }
tail -2 gpurun_out/r03_final_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r03_e_bench_default.log 2> gpurun_out/r03_e_bench_default.err
timeout -k 10 400 python tools/time_all.py > gpurun_out/time_all.log 2>&1
timeout -k 10 500 bash tools/jpeg_runs.sh > gpurun_out/jpeg_runs.log 2>&1
grep -c identical gpurun_out/jpeg_runs.log
cd /tmp && export TMPDIR=/tmp
for k in px_huffman px_huffman_k1; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fused_$k -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_kernel.py $k 12 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fused_$k.log 2>&1
done
cd $GRAFT_REPO_ROOT
for k in px_huffman px_huffman_k1; do echo "== $k"; python3 tools/pmc_agg.py gpurun_out/pmc_fused_$k/ k_px_huffman; done > gpurun_out/pmc_fused_summary.txt 2>&1
cat gpurun_out/pmc_fused_summary.txt
