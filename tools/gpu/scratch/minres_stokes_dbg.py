import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from hyteg_amd import host
import test_gpu_minres as T
MESHES = ROOT / "hyteg_amd" / "data" / "meshes"
level = 2
for mesh in ("cube_24el", "regular_octahedron_8el"):
    for prec in ("pressure", "identity"):
        for its in (10, 50, 200, 1000, 3000):
            st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
            L, u, f, r, exact = T._stokes_problem(host, st, level, level)
            flag = host.Inner | host.NeumannBoundary
            def residual():
                L.apply(u, r, level, flag)
                r.assign([1.0, -1.0], [f, r], level, flag)
                return np.sqrt(r.dot(r, level, flag))
            r0 = residual()
            mr = host.StokesSolver.minres(st, level, level, its, 1e-15, prec)
            mr.solve(L, u, f, level)
            print(mesh, prec, "max_iter", its, "its", mr.minres_iterations, "r0", r0, "r", residual(), flush=True)
            st.close()
