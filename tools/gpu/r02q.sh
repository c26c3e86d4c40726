set -e
for lvl in 8 6; do
for v in "HYTEG_HIP_RESTRICT_SKIP=0" "HYTEG_HIP_RESTRICT_SKIP=1" "HYTEG_HIP_RESTRICT_SKIP=2" "HYTEG_HIP_RESTRICT_SKIP=3"; do
echo "== level $lvl $v (1: no edge points, 2: no face rows)"
env $v timeout -k 10 300 python tools/bench_kernels.py --level $lvl --only "restrict (fine" 2>&1 | grep -v "^{\|amdgpu.ids"
done
done
