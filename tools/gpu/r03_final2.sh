#!/bin/bash
# end of round 3, after the P2 class-rows kernel: kernel table with the CPU rows, rocprof stats of the P2 apply probe, then the whole suite + bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final2; mkdir -p $O
timeout -k 10 900 python tools/bench_kernels.py --level 8 --cpu > $O/kernel_table.txt 2>&1; tail -3 $O/kernel_table.txt | cut -c1-150
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/gpu/scratch/p2_class_rows_probe.py > $O/prof_run.log 2>&1 ) || echo "profile failed"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/p2_class_rows_kernel_stats.csv && head -6 $O/p2_class_rows_kernel_stats.csv | cut -c1-220
rm -rf $O/prof
bash $R/tools/gpu/full.sh final2
