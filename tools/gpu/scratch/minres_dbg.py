import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from hyteg_amd import host
from hostutil import MultiCellOracle, upload
MESHES = ROOT / "hyteg_amd" / "data" / "meshes"
for mesh in ("cube_6el", "tet_1el", "regular_octahedron_8el"):
    min_level, max_level = 2, 4
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    mo = MultiCellOracle(st)
    L = host.P1ConstantOperator(st, min_level, max_level)
    L.compute_inverse_diagonal()
    for which in ("cg", "minres0", "minres10"):
        u, f, ex, err, tmp, r = (host.P1Function(st, n, min_level, max_level) for n in ("u", "f", "u_exact", "err", "tmp", "r"))
        for fn in (u, f, ex, err, r):
            fn.interpolate(0.0, max_level, host.All)
        upload(ex, mo.interpolate(lambda x, y, z: x * x - y * y, max_level), max_level)
        u.assign([1.0], [ex], max_level, host.DirichletBoundary)
        L.apply(ex, r, max_level, host.Inner)
        res_exact = np.sqrt(r.dot(r, max_level, host.Inner))
        if which == "cg":
            s = host.Solver.cg(st, min_level, max_level, 2000, 1e-14)
        else:
            s = host.Solver.minres(st, min_level, max_level, 1000, 1e-8, 0 if which == "minres0" else 10)
        s.solve(L, u, f, max_level)
        L.apply(u, r, max_level, host.Inner)
        res = np.sqrt(r.dot(r, max_level, host.Inner))
        err.assign([1.0, -1.0], [u, ex], max_level, host.All)
        tmp.interpolate(1.0, max_level, host.All)
        l2 = np.sqrt(err.dot(err, max_level, host.All) / tmp.dot(tmp, max_level, host.All))
        print(mesh, which, "residual of exact", res_exact, "residual after", res, "l2 err", l2, flush=True)
