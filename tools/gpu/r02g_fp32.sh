set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fp32.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_fp32.txt 2>&1 || { tail -60 $O/pytest_fp32.txt; exit 1; }
tail -5 $O/pytest_fp32.txt
python bench.py --no-cpu-baseline > $O/bench_default.json 2>&1; tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e9,1), round(r['launch_us'],2), round(r['frac'],3), r['kernel'])"
