set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02p2p; mkdir -p $O
timeout -k 10 300 python hyteg_amd/csrc/exp/p2p_probe.py > $O/p2p_probe.txt 2>&1 || { tail -30 $O/p2p_probe.txt; exit 1; }
grep -v amdgpu.ids $O/p2p_probe.txt
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_dist.txt 2>&1 || { tail -60 $O/pytest_dist.txt; exit 1; }
tail -3 $O/pytest_dist.txt
