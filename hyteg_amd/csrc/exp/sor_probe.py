#!/usr/bin/env python3
"""Times the forms of the macro-cell SOR sweep per level (development probe):  python sor_probe.py [levels...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi  # noqa: E402

capi.lib()
levels = [int(a) for a in sys.argv[1:]] or [5, 6, 7, 8]
w = [-0.1] * 15
w[7] = 2.0
for L in levels:
    capi.prepare_level(L)
    n = capi.cell_size(L)
    u = torch.rand(n, dtype=torch.float64, device="cuda")
    b = torch.rand(n, dtype=torch.float64, device="cuda")
    for name in ("SOR_BLOCKS", "SOR_DATAFLOW"):
        capi.set_sor_algorithm(getattr(capi, name))
        for bw in (False, True):
            for _ in range(3):
                capi.p1_sor_cell(u.data_ptr(), b.data_ptr(), L, w, 1.0, bw)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                capi.p1_sor_cell(u.data_ptr(), b.data_ptr(), L, w, 1.0, bw)
            torch.cuda.synchronize()
            print(f"level {L} {name:13s} {'backward' if bw else 'forward ':8s} {(time.perf_counter() - t0) / reps * 1e6:9.1f} us", flush=True)
