set -u
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
K=signal_sums_kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 scripts/run_signals_once.py 1000000 cluster 3 > $O/run.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-200
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 scripts/run_signals_once.py 1000000 cluster 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p2 -- python3 scripts/run_signals_once.py 1000000 cluster 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p3 -- python3 scripts/run_signals_once.py 1000000 cluster 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/p4 -- python3 scripts/run_signals_once.py 1000000 cluster 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/p5 -- python3 scripts/run_signals_once.py 1000000 cluster 3 > /dev/null 2>&1
python3 scripts/pmc_kernel.py $K $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 | tee $O/pmc_signal_sums.txt
rm -rf $O/stats $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
