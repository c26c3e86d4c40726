#!/bin/bash
# builds the developer micro-benchmark next to this script (binary is git-ignored)
set -e
D="$(cd "$(dirname "$0")" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -DHYTEG_HIP_BUILDING -o "$D/apply_bench" "$D/apply_bench.hip" "$D/../runtime.hip"
