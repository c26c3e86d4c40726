#!/bin/bash
# On the GPU box: kernel-trace stats of the default bench line, then the PMC traffic passes.  Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_bench; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench.log 2>&1 || echo "stats pass failed"
bash $R/hyteg_amd/csrc/exp/pmc_bench.sh > $R/gpurun_out/pmc_bench.log 2>&1
find $O -name "*kernel_stats.csv" | head -3
