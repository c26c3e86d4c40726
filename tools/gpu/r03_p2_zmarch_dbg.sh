#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_p2_zmarch; mkdir -p $O
for d in 4 8 11; do echo "debug $d (1 loads off, 2 stores off, 4 return after the tile load, 8 no FMAs)"; HYTEG_HIP_P2_ZM_DEBUG=$d HYTEG_HIP_P2_ZMARCH_LZ=1 timeout -k 10 200 python tools/gpu/scratch/p2_zmarch_probe.py 2>&1 | grep "level 7 z" | tee -a $O/probe_dbg.txt; done
