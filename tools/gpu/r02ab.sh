set -e
cp hyteg_amd/lib/libhyteg_hip.so /tmp/libhyteg_hip_orig.so
for rep in 1 2; do
for lz in 8 4 2; do
cp gpurun_variants/libhyteg_hip_lz$lz.so hyteg_amd/lib/libhyteg_hip.so
echo -n "LZ=$lz level 8: "; timeout -k 10 300 python tools/bench_kernels.py --level 8 --only "olongate Repl" 2>&1 | grep "^prolongate"
echo -n "LZ=$lz level 7: "; timeout -k 10 300 python tools/bench_kernels.py --level 7 --only "olongate Repl" 2>&1 | grep "^prolongate"
done
done
cp gpurun_variants/libhyteg_hip_lz4.so hyteg_amd/lib/libhyteg_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_host.py -m gpu -x -q -k "prolong or transfer or gmg" 2>&1 | tail -1
cp /tmp/libhyteg_hip_orig.so hyteg_amd/lib/libhyteg_hip.so
