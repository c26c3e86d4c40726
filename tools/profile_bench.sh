#!/bin/bash
# On the GPU box: (1) kernel-trace stats of a bench.py run, (2) the PMC traffic passes of the same command (separate --pmc
# passes: FETCH_SIZE and WRITE_SIZE cannot share one).  bench.py's own copy-floor kernel (calib_copy_kernel<true>: one level-8
# cell array read once and written once, known bytes) is in the same traces and calibrates the counters.
# Outputs under gpurun_out/prof_bench and gpurun_out/pmc_bench; post-process with tools/pmc_traffic.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_bench; P=$R/gpurun_out/pmc_bench; mkdir -p $O $P
# HYTEG_BENCH_SMALL_RING=0: without the extra regions on the Infinity-Cache-sized ring, so that every launch of the apply kernel in the
# trace is an HBM-regime launch and the average of the stats file is the figure bench.py reports
export HYTEG_BENCH_SMALL_RING=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 500 --warmup 50 --regions 5 --no-cpu-baseline > $O/bench.log 2>&1 || echo "stats pass failed"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $C | tr " " "_" | cut -c1-24)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $P/bench_$N -- python3 $R/bench.py --steps 100 --warmup 20 --regions 2 --no-cpu-baseline > $P/bench_$N.log 2>&1 || echo "pmc pass $N failed"
done
find $O -name "*kernel_stats.csv" | head -3
