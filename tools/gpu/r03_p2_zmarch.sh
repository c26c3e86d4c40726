#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_p2_zmarch; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_p2_zmarch.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for lz in 1 2; do HYTEG_HIP_P2_ZMARCH_LZ=$lz timeout -k 10 200 python tools/gpu/scratch/p2_zmarch_probe.py 2>&1 | grep level | tee -a $O/probe.txt; done
