set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02d; mkdir -p $O
cd hyteg_amd/csrc/exp
./apply_trace_time 8 400 9 > $O/trace_time_l8.txt 2>&1
./apply_trace_time 7 400 9 > $O/trace_time_l7.txt 2>&1
./apply_trace 8 100 9 > $O/trace_l8.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_levels.py -x -q -m gpu > $O/pytest_apply.txt 2>&1
HYTEG_HIP_APPLY_DECODE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2>&1
HYTEG_HIP_APPLY_DECODE=0 python bench.py --no-cpu-baseline > $O/bench_default.json 2>&1
tail -2 $O/pytest_apply.txt; cat $O/trace_time_l8.txt $O/trace_time_l7.txt; head -12 $O/trace_l8.txt
