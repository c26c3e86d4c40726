#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_p2_batch; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_p2_class_rows.py tests/test_gpu_p2_sor_shared.py tests/test_gpu_p2_gmg.py tests/test_gpu_taylor_hood.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/bench_kernels.py --level 8 > $O/kernel_table.txt 2>&1; grep -i "Taylor\|P2 Gauss" $O/kernel_table.txt | cut -c1-200
