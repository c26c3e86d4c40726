#!/usr/bin/env python3
"""Developer probe: does the relative placement of the arrays of a three-stream kernel (fused Jacobi: src, rhs -> dst) matter
on MI355X (HBM channel / bank conflicts between streams that advance in lock step)?  Arrays are carved out of one big
allocation at chosen byte distances.  Usage: python hyteg_amd/csrc/exp/stream_offset_probe.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from hyteg_amd import capi, host  # noqa: E402

L = 8
n = capi.cell_size(L)
st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh")
stream = torch.cuda.current_stream()
sh = stream.cuda_stream
st.set_stream(sh)
w = list(host.P1ConstantOperator(st, 2, L).stencils(0, L)[0])
nbytes = n * 8
base = (nbytes + 4095) // 4096 * 4096  # array size rounded to 4 KiB
nsets = 6  # rotate over more than the Infinity Cache


def run(pad):
    stride = base + pad
    buf = torch.rand((3 * nsets * stride) // 8 + 1024, dtype=torch.float64, device="cuda")
    p0 = buf.data_ptr()
    ptr = lambda s, k: p0 + (3 * s + k) * stride  # noqa: E731
    for name, fn in (("apply  ", lambda s: capi.p1_apply_cell(ptr(s, 0), ptr(s, 1), L, w, 0, sh)),
                     ("jacobi ", lambda s: capi.p1_jacobi_cell(ptr(s, 0), ptr(s, 2), ptr(s, 1), L, w, 0.66, None, sh))):
        for k in range(5):
            fn(k % nsets)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(300):
            fn(k % nsets)
        e1.record(stream)
        torch.cuda.synchronize()
        print(f"distance between arrays = 4 KiB-rounded size + {pad:8d} B: {name} {e0.elapsed_time(e1) * 1e3 / 300:6.2f} us", flush=True)
    del buf


import os
PADS = [int(x) for x in os.environ.get('PADS', '0,256,4096,65536,1048576,2109440').split(',')]
for pad in PADS:
    run(pad)
