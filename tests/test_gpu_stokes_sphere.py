"""BASELINE config 5's workload: the P1-P1 Stokes operator and its Uzawa multigrid cycle on a thick spherical shell
(apps/stokesSphere/StokesSphere.cpp with its parameter file: ntan 2, layers 1 / 2 / 3, levels 2-3, V(2,2) with increment 2,
Uzawa( 0.3 ), pressure-preconditioned MINRES( 10 ) on the coarsest level, the plume right-hand side) -- on the mesh of the
repository's own generator (hyteg_amd/meshgen.py), with the Gauss-Seidel velocity smoother of the app and with the
mixed-precision ("fp32") Jacobi smoother config 5 names."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
MESH = ROOT / "hyteg_amd" / "data" / "meshes" / "spherical_shell_ntan2_3layers.msh"


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi, host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    host.lib()
    return torch, capi, host, po


def _rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def test_stokes_operator_on_the_shell_is_the_composition_of_the_oracle_blocks(env):
    """all 120 macro-cells: velocity rows lapl + divT, pressure row div + pspg against the oracle applied cell by cell"""
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, download, upload

    level = 2
    st = host.Storage.from_gmsh(MESH)
    assert st.n_local_cells == 120
    mo = MultiCellOracle(st)
    L = host.P1P1StokesOperator(st, level, level)
    src, dst = host.P1StokesFunction(st, "src", level, level), host.P1StokesFunction(st, "dst", level, level)
    fields = [lambda x, y, z: np.sin(x + y) + z * z, lambda x, y, z: x * y - np.cos(z), lambda x, y, z: x + 0.5 * y * z,
              lambda x, y, z: np.sin(x * y) + z]
    src_h = [mo.interpolate(f, level) for f in fields]
    for k in range(4):
        upload(src.components[k], src_h[k], level)
        dst.components[k].interpolate(0.0, level, host.All)
    flag = host.Inner | host.NeumannBoundary
    L.apply(src, dst, level, flag)
    zeros = lambda: [np.zeros(po.cell_size(level)) for _ in src_h[0]]  # noqa: E731
    for k in range(3):
        ref = mo.apply(src_h[k], zeros(), level, flag, po.FORM_LAPLACE)
        add = mo.apply(src_h[3], zeros(), level, flag, po.FORM_DIVT_X + k)
        got = download(dst.components[k], level)
        for c, (g, r_, a) in enumerate(zip(got, ref, add)):
            sel = ((st.mask(c, flag) >> po.slot_of_points(level)) & 1).astype(bool)
            assert np.all(g[~sel] == 0.0)
            assert _rel(g[sel], (r_ + a)[sel]) < 1e-12
    ref = mo.apply(src_h[3], zeros(), level, host.All, po.FORM_PSPG)
    for k in range(3):
        part = mo.apply(src_h[k], zeros(), level, host.All, po.FORM_DIV_X + k)
        ref = [r_ + p_ for r_, p_ in zip(ref, part)]
    for g, r_ in zip(download(dst.p, level), ref):
        assert _rel(g, r_) < 1e-12
    for o in (src, dst, L, st):
        o.close()


@pytest.mark.parametrize("velocity_smoother", ["gauss_seidel", "jacobi_fp32"])
def test_stokes_sphere_uzawa_cycles_reduce_the_residual(env, velocity_smoother):
    """StokesSphere.cpp:159-196, 224-300 with StokesSphere.prm: f.uvw = the plume (x, y, z) ( 0.5 - |x - source| ) inside the source
    ball, zero Dirichlet velocity on both spheres; the residual (the app's discrete L2 norm, :262-264) decreases in every cycle.
    jacobi_fp32: the velocity smoother of the Uzawa smoother is the mixed-precision Jacobi smoother (float sweeps on the cell
    interiors) in place of the app's Gauss-Seidel"""
    torch, capi, host, po = env
    from hostutil import MultiCellOracle, upload

    min_level, max_level = 2, 3
    st = host.Storage.from_gmsh(MESH)
    mo = MultiCellOracle(st)
    L = host.P1P1StokesOperator(st, min_level, max_level)
    u, f, r = (host.P1StokesFunction(st, n, min_level, max_level) for n in ("u", "f", "r"))
    for fn in (u, f, r):
        for lvl in (min_level, max_level):
            for k in range(4):
                fn.components[k].interpolate(0.0, lvl, host.All)
    src, radius = np.array([2.0, 0.0, 0.0]), 0.5

    def plume(k):
        def fn(x, y, z):
            d = np.sqrt((x - src[0]) ** 2 + (y - src[1]) ** 2 + (z - src[2]) ** 2)
            return np.where(d < radius, (x, y, z)[k] * (radius - d), 0.0)

        return fn

    for k in range(3):
        upload(f.components[k], mo.interpolate(plume(k), max_level), max_level)
    one = host.P1Function(st, "one", max_level, max_level)
    one.interpolate(1.0, max_level, host.All)
    ndofs = 4.0 * one.dot(one, max_level, host.All)
    flag = host.Inner | host.NeumannBoundary

    def residual():
        L.apply(u, r, max_level, flag)
        r.assign([1.0, -1.0], [f, r], max_level, flag)
        return np.sqrt(r.dot(r, max_level, host.Inner)) / ndofs

    vs = {"gauss_seidel": host.GAUSS_SEIDEL, "jacobi_fp32": host.JACOBI_FP32}[velocity_smoother]
    smoother = host.StokesSolver.uzawa(st, min_level, max_level, 0.3, velocity_iterations=2, velocity_smoother=vs, velocity_relax=2.0 / 3.0)
    gmg = host.StokesSolver.gmg(st, smoother, min_level, max_level, pre=2, post=2, increment=2, project_mean_after_restriction=True,
                                coarse="minres", coarse_max_iter=10, coarse_rel_tol=1e-16)
    res = [residual()]
    assert res[0] > 0.0
    for _ in range(4):
        gmg.solve(L, u, f, max_level)
        res.append(residual())
    assert all(res[i + 1] < res[i] for i in range(4)), res
    assert res[-1] < 0.2 * res[0], res
    for o in (gmg, smoother, u, f, r, one, L, st):
        o.close()
