set -e
for l in 5 6; do timeout -k 10 300 python tools/bench_kernels.py --level $l --only "SOR" 2>&1 | grep "^SOR" | grep -v "dataflow\|shell"; done
timeout -k 10 300 python tools/vcycle_breakdown.py --smoother jacobi --max 8 2>&1 | grep -v amdgpu.ids | tail -9
HYTEG_HIP_SOR_PIPELINE=0 timeout -k 10 300 python tools/vcycle_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -7
