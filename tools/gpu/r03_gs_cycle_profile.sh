#!/bin/bash
# rocprofv3 kernel stats of V(3,3) Gauss-Seidel cycles on regular_octahedron_8el, levels 0..6 (BASELINE config 3's shape): where the
# 3.8 ms of a cycle go.  usage: r03_gs_cycle_profile.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_gs_cycle; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/hyteg_amd/csrc/exp/vcycle_probe.py regular_octahedron_8el 6 0 gs > $O/run.log 2>&1 || echo "profile failed"
tail -2 $O/run.log
f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -25 $O/kernel_stats.csv | cut -c1-200
