"""P2-P1 Taylor-Hood Stokes (BASELINE config 5's "P2-P1 Stokes block operator"; hyteg_amd/host/taylorhood.hpp):
 * the mixed blocks div (P2 -> P1) and divT (P1 -> P2) -- the P2 apply kernel with padded element matrices -- against the oracle's
   micro-cell loop with the same matrices (which tests/test_oracle_taylor_hood.py pins to the reference's FEniCS forms);
 * div and divT are adjoint on a mesh of several macro-cells (the shared DoFs' shares travel through the additive exchange);
 * the composite operator: a divergence-free linear velocity with a constant pressure is in the kernel at all inner DoFs;
 * the reference's known answer tests/hyteg/convergence/P2P1Stokes3DUzawaConvergenceTest.cpp (cube_24el, levels 2-3, three
   V(3,3) cycles of Uzawa(0.4) over Gauss-Seidel, colliding flow): its three bounds, with pressure-preconditioned MINRES on the
   coarsest level where the reference uses PETSc's LU."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

MICRO = [[(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], [(1, 0, 0), (1, 1, 0), (0, 1, 0), (1, 0, 1)], [(1, 0, 0), (0, 1, 0), (1, 0, 1), (0, 0, 1)],
         [(1, 1, 0), (1, 1, 1), (0, 1, 1), (1, 0, 1)], [(1, 0, 1), (0, 1, 1), (0, 0, 1), (0, 1, 0)], [(0, 1, 0), (1, 1, 0), (1, 0, 1), (0, 1, 1)]]


def _cell_matrices(host, co, level, which, k):
    """the six padded element matrices of a macro-cell's micro-cell types (celldof::allCellTypes order)"""
    c = np.asarray(co, dtype=np.float64).reshape(4, 3)
    h = 1.0 / (1 << level)
    out = []
    for verts in MICRO:
        t = np.array([c[0] + h * (v[0] * (c[1] - c[0]) + v[1] * (c[2] - c[0]) + v[2] * (c[3] - c[0])) for v in verts])
        out.append(host.taylor_hood_form_element_matrix(which, k, t.reshape(12)))
    return np.array(out)


def _env():
    import torch

    import hostutil as hu
    from hyteg_amd import host
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    return host, hu, po


@pytest.mark.parametrize("level", [2, 3])
def test_div_and_divt_blocks_against_the_oracle(level):
    host, hu, po = _env()
    st = host.Storage.from_gmsh(hu.MESHES / "tet_1el.msh")
    gid, co, nnc = st.local_cell(0)
    A = host.TaylorHoodStokesOperator(st, level, level)
    x, y = host.TaylorHoodFunction(st, "x", level, level), host.TaylorHoodFunction(st, "y", level, level)
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    rng = np.random.default_rng(3)
    U = [(rng.standard_normal(nv), rng.standard_normal(ne)) for _ in range(3)]
    P = rng.standard_normal(nv)
    for k in range(3):
        x.velocity[k].upload(level, U[k][0], U[k][1])
    x.pressure.upload_cell(0, level, P)
    # div: pressure rows
    A.apply_div(x, y, level, host.All)
    want = np.zeros(nv)
    for k in range(3):
        em = _cell_matrices(host, co, level, 0, k)
        ov, oe = po.p2_elementwise_apply_cell(np.zeros(nv), np.zeros(ne), U[k][0], U[k][1], level, em, 1.0, 0, 0x7FFF)
        want += ov
    got = y.pressure.download_cell(0, level)
    assert np.abs(got - want).max() < 1e-13 * max(1.0, np.abs(want).max())
    # divT: velocity rows from the pressure
    A.apply_divt(x, y, level, host.All)
    for k in range(3):
        em = _cell_matrices(host, co, level, 1, k)
        ov, oe = po.p2_elementwise_apply_cell(np.zeros(nv), np.zeros(ne), P, np.zeros(ne), level, em, 1.0, 0, 0x7FFF)
        gv, ge = y.velocity[k].download(level)
        assert np.abs(gv - ov).max() < 1e-13 * max(1.0, np.abs(ov).max())
        assert np.abs(ge - oe).max() < 1e-13 * max(1.0, np.abs(oe).max())
    for o in (x, y, A, st):
        o.close()


def _fields(hu, po, st, level, fn):
    out = []
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        out.append((fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))))
    return out


def test_div_and_divt_are_adjoint_on_several_macro_cells():
    host, hu, po = _env()
    level = 3
    st = host.Storage.from_gmsh(hu.MESHES / "cube_6el.msh")
    A = host.TaylorHoodStokesOperator(st, level, level)
    x, y = host.TaylorHoodFunction(st, "x", level, level), host.TaylorHoodFunction(st, "y", level, level)
    fu = [lambda p: np.sin(3 * p[:, 0] + p[:, 1]) * p[:, 2], lambda p: p[:, 0] ** 2 - p[:, 1] * p[:, 2], lambda p: np.cos(2 * p[:, 1]) + p[:, 0]]
    fp = lambda p: np.sin(2 * p[:, 0]) * p[:, 1] + p[:, 2] ** 2  # noqa: E731
    for k in range(3):
        for c, (v, e) in enumerate(_fields(hu, po, st, level, fu[k])):
            x.velocity[k].upload(level, v, e, c)
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        x.pressure.upload_cell(c, level, fp(hu.cell_points(co, level)))
    A.apply_div(x, y, level, host.All)     # y.p = B u
    A.apply_divt(x, y, level, host.All)    # y.uvw = B^T p
    lhs = y.pressure.dot(x.pressure, level, host.All)
    rhs = sum(y.velocity[k].dot(x.velocity[k], level, host.All) for k in range(3))
    assert abs(lhs - rhs) < 1e-12 * max(abs(lhs), 1.0), (lhs, rhs)
    for o in (x, y, A, st):
        o.close()


def test_linear_divergence_free_flow_with_constant_pressure_is_in_the_kernel():
    host, hu, po = _env()
    level = 3
    st = host.Storage.from_gmsh(hu.MESHES / "cube_6el.msh")
    A = host.TaylorHoodStokesOperator(st, level, level)
    x, y = host.TaylorHoodFunction(st, "x", level, level), host.TaylorHoodFunction(st, "y", level, level)
    fu = [lambda p: 2.0 * p[:, 1] - p[:, 2], lambda p: p[:, 0] + 0.5 * p[:, 2], lambda p: 3.0 * p[:, 0] - p[:, 1]]
    for k in range(3):
        for c, (v, e) in enumerate(_fields(hu, po, st, level, fu[k])):
            x.velocity[k].upload(level, v, e, c)
    x.pressure.interpolate(1.7, level, host.All)
    y.interpolate(0.0, level, host.All)
    A.apply(x, y, level, host.Inner)
    # momentum rows at the inner velocity DoFs: A u = 0 (linear), B^T const = 0 (its test functions vanish on the boundary);
    # continuity rows: div u = 0 against every pressure test function
    for k in range(3):
        assert abs(y.velocity[k].dot(y.velocity[k], level, host.Inner)) < 1e-22
    assert abs(y.pressure.dot(y.pressure, level, host.All)) < 1e-22
    for o in (x, y, A, st):
        o.close()


def test_p2p1_stokes_3d_uzawa_convergence():
    """P2P1Stokes3DUzawaConvergenceTest.cpp:56-207"""
    host, hu, po = _env()
    lo, hi = 2, 3
    st = host.Storage.from_gmsh(hu.MESHES / "cube_24el.msh")
    L = host.TaylorHoodStokesOperator(st, lo, hi)
    u, f, r, exact, err = (host.TaylorHoodFunction(st, n_, lo, hi) for n_ in ("u", "f", "r", "exact", "err"))
    flow = [lambda p: 20.0 * p[:, 0] * p[:, 1] ** 3, lambda p: 5.0 * p[:, 0] ** 4 - 5.0 * p[:, 1] ** 4, lambda p: 0.0 * p[:, 0]]
    pres = lambda p: 60.0 * p[:, 0] ** 2 * p[:, 1] - 20.0 * p[:, 1] ** 3  # noqa: E731
    tmp = host.TaylorHoodFunction(st, "tmp", lo, hi)
    for k in range(3):
        for c, (v, e) in enumerate(_fields(hu, po, st, hi, flow[k])):
            exact.velocity[k].upload(hi, v, e, c)
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        exact.pressure.upload_cell(c, hi, pres(hu.cell_points(co, hi)))
    # u = the flow on the Dirichlet boundary, zero inside
    u.interpolate(0.0, hi, host.All)
    u.assign([1.0], [exact], hi, host.DirichletBoundary)
    u.pressure.interpolate(0.0, hi, host.All)
    ones = host.TaylorHoodFunction(st, "ones", lo, hi)
    ones.interpolate(1.0, hi, host.All)
    n_vel = ones.velocity[0].dot(ones.velocity[0], hi, host.All)
    n_p = ones.pressure.dot(ones.pressure, hi, host.All)
    solver = host.TaylorHoodSolver.gmg(st, lo, hi, uzawa_relax=0.4, pre=3, post=3, increment=0, coarse_max_iter=int(__import__("os").environ.get("TH_COARSE_ITER", "2000")), coarse_rel_tol=1e-12)
    flag = host.Inner | host.NeumannBoundary
    f.interpolate(0.0, hi, host.All)
    res = []
    for it in range(3):
        solver.solve(L, u, f, hi)
        u.project_pressure_mean(hi)
        exact.project_pressure_mean(hi)
        r.interpolate(0.0, hi, host.All)
        L.apply(u, r, hi, flag)
        err.assign([1.0, -1.0], [u, exact], hi, host.All)
        res.append(np.sqrt(r.dot(r, hi, host.All) / (3 * n_vel + n_p)))
    e_uvw = sum(np.sqrt(err.velocity[k].dot(err.velocity[k], hi, host.All) / n_vel) for k in range(3))
    e_p = np.sqrt(err.pressure.dot(err.pressure, hi, host.All) / n_p)
    assert e_uvw < 3e-3, (e_uvw, e_p, res)
    assert e_p < 0.8, (e_uvw, e_p, res)
    assert res[-1] < 5.0e-5, (e_uvw, e_p, res)
    for o in (solver, u, f, r, exact, err, tmp, ones, L, st):
        o.close()


def test_taylor_hood_v_cycle_on_the_spherical_shell():
    """config 5 in its literal wording -- the P2-P1 Stokes block operator's V-cycle on a spherical-shell mesh: the set-up of
    apps/stokesSphere/StokesSphere.cpp (plume right-hand side, V(2,2) with increment 2, Uzawa( 0.3 ), pressure-preconditioned MINRES
    on the coarsest level) with the Taylor-Hood operator on the shell of hyteg_amd/meshgen.py (ntan 2, three layers, 120 tetrahedra,
    levels 2-3): the residual of the inner equations falls in every cycle"""
    host, hu, po = _env()
    lo, hi = 2, 3
    st = host.Storage.from_gmsh(hu.MESHES / "spherical_shell_ntan2_3layers.msh")
    L = host.TaylorHoodStokesOperator(st, lo, hi)
    u, f, r = (host.TaylorHoodFunction(st, n_, lo, hi) for n_ in ("u", "f", "r"))
    src, radius = np.array([0.0, 0.0, 1.5]), 0.6

    def plume(k):
        def fn(p):
            d = np.sqrt(((p - src[None, :]) ** 2).sum(axis=1))
            return np.where(d < radius, p[:, k] * (radius - d), 0.0)
        return fn

    f.interpolate(0.0, hi, host.All)
    u.interpolate(0.0, hi, host.All)
    for k in range(3):
        for c, (v, e) in enumerate(_fields(hu, po, st, hi, plume(k))):
            f.velocity[k].upload(hi, v, e, c)
    flag = host.Inner | host.NeumannBoundary

    def residual():
        r.interpolate(0.0, hi, host.All)
        L.apply(u, r, hi, flag)
        r.assign([1.0, -1.0], [f, r], hi, flag)
        return np.sqrt(r.dot(r, hi, flag))

    solver = host.TaylorHoodSolver.gmg(st, lo, hi, uzawa_relax=0.3, pre=2, post=2, increment=2, coarse_max_iter=50, coarse_rel_tol=1e-16)
    res = [residual()]
    assert res[0] > 0.0
    for _ in range(4):
        solver.solve(L, u, f, hi)
        u.project_pressure_mean(hi)
        res.append(residual())
    assert all(b < 0.9 * a for a, b in zip(res, res[1:])), res
    assert res[-1] < 0.2 * res[0], res
    for o in (solver, u, f, r, L, st):
        o.close()
