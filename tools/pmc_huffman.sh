# counters of k_huffman_rows (separate --pmc passes, no trace domains): bash tools/pmc_huffman.sh [traffic|sq]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/huffpmc
mkdir -p $O
what=${1:-traffic}
for k in huffman huffman_k1; do
  if [ $what = traffic ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c -d $O/${k}_$c -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 8 > $O/${k}_$c.log 2>&1 || exit 1
    done
  else
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/${k}_A -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 8 > $O/${k}_A.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d $O/${k}_B -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 8 > $O/${k}_B.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $O/${k}_C -o run --output-format csv -- python3 $R/tools/run_kernel.py $k 8 > $O/${k}_C.log 2>&1 || exit 1
  fi
done
cd $R
for d in $O/*/; do echo "== $d"; python3 tools/pmc_agg.py $d huffman; done > $O/summary_$what.txt 2>&1
cat $O/summary_$what.txt
