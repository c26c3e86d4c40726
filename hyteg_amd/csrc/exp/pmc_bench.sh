#!/bin/bash
# HBM traffic of the apply kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass), plus a
# calibration pass on copy kernels of known size with the same access width (8 B per lane).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench; mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr " " "_" | cut -c1-24)
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/bench_$N -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_$N.log 2>&1 || echo "bench pass $N failed"
  timeout -k 10 100 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/calib_$N -- $R/hyteg_amd/csrc/exp/copy_calib 20 > $O/calib_$N.log 2>&1 || echo "calib pass $N failed"
done
ls $O
