set -e
timeout -k 10 900 python -m pytest tests/test_gpu_large_levels.py -m gpu -x -q -k "level_9_to_10 or level9" 2>&1 | tail -15
