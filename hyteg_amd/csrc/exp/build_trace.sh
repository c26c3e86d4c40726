#!/bin/bash
# builds the round-2 apply harness next to this script: apply_trace_time (timing only) and apply_trace (per-wave timestamps)
set -e
D="$(cd "$(dirname "$0")" && pwd)"
F="--offload-arch=gfx950 -O3 -std=c++20 -DHYTEG_HIP_BUILDING"
/opt/rocm/bin/hipcc $F -o "$D/apply_trace_time" "$D/apply_trace.hip" "$D/../runtime.hip"
/opt/rocm/bin/hipcc $F -DZM_DO_TRACE -o "$D/apply_trace" "$D/apply_trace.hip" "$D/../runtime.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -o "$D/icache_probe" "$D/icache_probe.hip"
/opt/rocm/bin/hipcc $F -o "$D/sor_trace" "$D/sor_trace.hip" "$D/../runtime.hip" "$D/../p1_sor_dataflow.hip"
/opt/rocm/bin/hipcc $F -DSOR_LOOKAHEAD -o "$D/sor_trace_lookahead" "$D/sor_trace.hip" "$D/../runtime.hip" "$D/../p1_sor_dataflow.hip"
/opt/rocm/bin/hipcc $F -DSOR_LOOKAHEAD -DSOR_STEPS -o "$D/sor_trace_lookahead_steps" "$D/sor_trace.hip" "$D/../runtime.hip" "$D/../p1_sor_dataflow.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -o "$D/lat_probe" "$D/lat_probe.hip"
