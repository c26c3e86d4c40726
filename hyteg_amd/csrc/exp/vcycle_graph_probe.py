#!/usr/bin/env python3
"""V(3,3) cycle time with and without launch graphs (development probe): python vcycle_graph_probe.py mesh lo hi [gs]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi, host  # noqa: E402

mesh, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sm = host.GAUSS_SEIDEL if len(sys.argv) > 4 and sys.argv[4] == "gs" else host.JACOBI
capi.lib(), host.lib()
for graphs in (False, True):
    st = host.Storage.from_gmsh(ROOT / f"hyteg_amd/data/meshes/{mesh}.msh")
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    A = host.P1ConstantOperator(st, lo, hi)
    A.compute_inverse_diagonal()
    x, b = host.P1Function(st, "x", lo, hi), host.P1Function(st, "b", lo, hi)
    rng = np.random.default_rng(0)
    for c in range(st.n_local_cells):
        x.upload_cell(c, hi, rng.random(capi.cell_size(hi)))
    x.sync_shared(hi, host.All)
    x.interpolate(0.0, hi, host.DirichletBoundary)
    gmg = host.Solver.gmg(st, lo, hi, smoother=sm, relax=2.0 / 3.0, pre=3, post=3, cg_max_iter=50, cg_tol=1e-10)
    gmg.set_use_graphs(graphs)
    for _ in range(3):
        gmg.solve(A, x, b, hi)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        gmg.solve(A, x, b, hi)
    torch.cuda.synchronize()
    print(f"{mesh} L{lo}-{hi} {'GS' if sm == host.GAUSS_SEIDEL else 'Jacobi'} graphs={graphs}: "
          f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per V(3,3) cycle (replayed {gmg.replayed_cycles})", flush=True)
