set -e
mkdir -p gpurun_out/r02a
cd hyteg_amd/csrc/exp
./apply_trace_time 8 300 9 > $GRAFT_REPO_ROOT/gpurun_out/r02a/trace_time_l8.txt 2>&1
./apply_trace 8 100 9 > $GRAFT_REPO_ROOT/gpurun_out/r02a/trace_l8.txt 2>&1
./apply_trace_time 7 300 9 > $GRAFT_REPO_ROOT/gpurun_out/r02a/trace_time_l7.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_levels.py -x -q -m gpu > gpurun_out/r02a/pytest_apply.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver_args.json 2> gpurun_out/r02a/bench_driver_args.err
HYTEG_HIP_APPLY_DECODE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_driver_args_table.json 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2>&1
HYTEG_HIP_APPLY_DECODE=0 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default_table.json 2>&1
tail -3 gpurun_out/r02a/pytest_apply.txt
cat gpurun_out/r02a/trace_time_l8.txt
