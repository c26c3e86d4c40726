#!/bin/bash
# rocprofv3 kernel stats of Taylor-Hood V(3,3) cycles on cube_24el (final build of round 3): where a cycle's time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_th_profile; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/gpu/scratch/th_cycle_probe.py > $O/run.log 2>&1 || echo "profile failed"
grep -v "^W2026\|^E2026\|^I2026" $O/run.log | tail -6
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/th_cycle_kernel_stats.csv; rm -rf $O/prof; head -16 $O/th_cycle_kernel_stats.csv | cut -c1-190
