set -e
for rep in 1 2; do
for v in 0 1 2 4 8; do
echo -n "stagger $v: "
HYTEG_HIP_APPLY_STAGGER=$v timeout -k 10 300 python tools/bench_kernels.py --level 8 --only "apply Replace" --reps 400 2>&1 | grep "^apply Replace  "
done
done
