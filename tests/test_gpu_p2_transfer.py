"""Quadratic (P2) grid transfer on the GPU: the gather kernels of hyteg_amd/csrc/p2_transfer.hip against the
push-formulated oracle (oracle/p2_transfer_oracle.py, pinned by the reference's known answers in
tests/test_oracle_p2_transfer.py), and through the host layer on several macro-cells with the reference's own checks
(tests/hyteg/P2/P2QuadraticProlongation3DTest.cpp:160-330: exact on constants / linears / quadratics, prolongateAndAdd =
prolongate + add; restriction = transpose of prolongation)."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
MESHES = ROOT / "hyteg_amd" / "data" / "meshes"


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi, host
    from oracle import p1_oracle as po
    from oracle import p2_transfer_oracle as pt

    assert torch.cuda.is_available()
    capi.lib()
    host.lib()
    return torch, capi, host, po, pt


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.mark.parametrize("lower", [0, 1, 2, 3, 4])
def test_prolongate_cell_matches_the_oracle(env, lower):
    torch, capi, host, po, pt = env
    rng = np.random.default_rng(lower)
    cv, ce = rng.standard_normal(po.cell_size(lower)), rng.standard_normal(po.edge_array_size(lower))
    nvf, nef = po.cell_size(lower + 1), po.edge_array_size(lower + 1)
    ov, oe = pt.prolongate_cell(cv, ce, lower)
    f0v, f0e = rng.standard_normal(nvf), rng.standard_normal(nef)
    cls_v, cls_e = po.slot_of_points(lower + 1), po.edge_classes(lower + 1)
    cvd, ced = _dev(torch, cv), _dev(torch, ce)  # kept alive: the C-ABI takes raw pointers
    for update in (capi.REPLACE, capi.ADD):
        for mask in (0x7FFF, 1 << 14, 0x03C0 | (1 << 14)):
            fv, fe = _dev(torch, f0v), _dev(torch, f0e)
            capi.p2_prolongate_cell(fv.data_ptr(), fe.data_ptr(), cvd.data_ptr(), ced.data_ptr(), lower, update, mask)
            torch.cuda.synchronize()
            sv, se = ((mask >> cls_v) & 1).astype(bool), ((mask >> cls_e) & 1).astype(bool)
            wv = np.where(sv, ov + (f0v if update == capi.ADD else 0.0), f0v)
            we = np.where(se, oe + (f0e if update == capi.ADD else 0.0), f0e)
            gv, ge = fv.cpu().numpy(), fe.cpu().numpy()
            assert np.array_equal(gv[~sv], f0v[~sv]) and np.array_equal(ge[~se], f0e[~se])  # unselected DoFs untouched
            assert np.abs(gv - wv).max() <= 1e-13 * 4 and np.abs(ge - we).max() <= 1e-13 * 4


@pytest.mark.parametrize("lower", [0, 1, 2, 3, 4])
def test_restrict_cell_matches_the_oracle(env, lower):
    torch, capi, host, po, pt = env
    rng = np.random.default_rng(10 + lower)
    fv, fe = rng.standard_normal(po.cell_size(lower + 1)), rng.standard_normal(po.edge_array_size(lower + 1))
    nnc = np.array([2, 3, 1, 4, 2, 5, 2, 1, 2, 2, 6, 7, 3, 8], dtype=np.float64)
    ov, oe = pt.restrict_cell(fv, fe, lower + 1, nnc)
    nvc, nec = po.cell_size(lower), po.edge_array_size(lower)
    c0v, c0e = rng.standard_normal(nvc), rng.standard_normal(nec)
    cls_v, cls_e = po.slot_of_points(lower), po.edge_classes(lower)
    fvd, fed = _dev(torch, fv), _dev(torch, fe)  # kept alive: the C-ABI takes raw pointers
    for mask in (0x7FFF, 1 << 14, 0x003F | (1 << 14)):
        cv, ce = _dev(torch, c0v), _dev(torch, c0e)
        capi.p2_restrict_cell(cv.data_ptr(), ce.data_ptr(), fvd.data_ptr(), fed.data_ptr(), lower, nnc, mask)
        torch.cuda.synchronize()
        sv, se = ((mask >> cls_v) & 1).astype(bool), ((mask >> cls_e) & 1).astype(bool)
        gv, ge = cv.cpu().numpy(), ce.cpu().numpy()
        assert np.array_equal(gv[~sv], c0v[~sv]) and np.array_equal(ge[~se], c0e[~se])
        scale = max(np.abs(ov).max(), np.abs(oe).max())
        if sv.any():
            assert np.abs(gv[sv] - ov[sv]).max() <= 1e-13 * scale
        if se.any():
            assert np.abs(ge[se] - oe[se]).max() <= 1e-13 * scale


def _fields(po, hu, st, level, fn):
    out = []
    for c in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(c)
        out.append((fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))))
    return out


@pytest.mark.parametrize("mesh", ["tet_1el", "cube_6el", "regular_octahedron_8el"])
def test_host_prolongation_reference_checks(env, mesh):
    """P2QuadraticProlongation3DTest.cpp testGridTransfer3D (:160-255) and testProlongateAndAdd3D (:257-318)"""
    torch, capi, host, po, pt = env
    import hostutil as hu

    lower = 2
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    flag = host.Inner | host.NeumannBoundary
    fns = [lambda p: 42.0 + 0.0 * p[:, 0], lambda p: 42.0 * p[:, 0] + p[:, 1] + 1337.0 * p[:, 2],
           lambda p: 2.0 * p[:, 0] ** 2 + 3.0 * p[:, 0] + 13.0 + 4.0 * p[:, 1] + 5.0 * p[:, 1] ** 2 + p[:, 2] ** 2 + 6.0]
    u, exact, err = (host.P2Function(st, n, lower, lower + 1) for n in ("u", "exact", "err"))
    for fn in fns:
        for c, (fv, fe) in enumerate(_fields(po, hu, st, lower, fn)):
            u.upload(lower, fv, fe, c)
        u.interpolate(0.0, lower + 1, host.All)
        exact.interpolate(0.0, lower + 1, host.All)
        # resultExact.interpolate( uFunction, lowerLevel + 1, Inner | NeumannBoundary )
        tmp = host.P2Function(st, "tmp", lower + 1, lower + 1)
        for c, (fv, fe) in enumerate(_fields(po, hu, st, lower + 1, fn)):
            tmp.upload(lower + 1, fv, fe, c)
        exact.assign([1.0], [tmp], lower + 1, flag)
        tmp.close()
        host.p2_prolongate(u, lower, flag)
        err.assign([1.0, -1.0], [u, exact], lower + 1, flag)
        scale = exact.dot(exact, lower + 1, flag)
        assert err.dot(err, lower + 1, flag) <= 1e-26 * max(scale, 1.0)  # the reference's limit is 1e-15 for values of O(1)
    # prolongateAndAdd = prolongate + add
    t_pro = lambda p: np.sin(p[:, 0]) + np.sinh(p[:, 1]) + p[:, 2] ** 2  # noqa: E731
    t_add = lambda p: np.cos(p[:, 0]) + np.cosh(p[:, 1]) + p[:, 2] ** 3  # noqa: E731
    a, b, add = (host.P2Function(st, n, lower, lower + 1) for n in ("a", "b", "add"))
    for f in (a, b):
        for c, (fv, fe) in enumerate(_fields(po, hu, st, lower, t_pro)):
            f.upload(lower, fv, fe, c)
    for f in (a, b, add):
        for c, (fv, fe) in enumerate(_fields(po, hu, st, lower + 1, t_add)):
            f.upload(lower + 1, fv, fe, c)
    host.p2_prolongate(a, lower, flag)
    host.p2_prolongate(b, lower, flag, add=True)
    a.add([1.0], [add], lower + 1, flag)
    err.assign([1.0, -1.0], [a, b], lower + 1, flag)
    assert err.dot(err, lower + 1, flag) < 1e-26 * max(1.0, a.dot(a, lower + 1, flag))
    for o in (u, exact, err, a, b, add, st):
        o.close()


@pytest.mark.parametrize("mesh", ["cube_6el", "regular_octahedron_8el"])
def test_host_restriction_is_the_transpose_of_the_prolongation(env, mesh):
    """< R f, c > = < f, P c > with every DoF counted once, over several macro-cells: checks the 1 / numNeighborCells scaling of
    the fine DoFs on shared macro-primitives together with the additive exchange"""
    torch, capi, host, po, pt = env
    import hostutil as hu

    lower = 2
    st = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    st.set_boundary_type(host.NeumannBoundary)  # no DoF is excluded
    f, c_ = host.P2Function(st, "f", lower, lower + 1), host.P2Function(st, "c", lower, lower + 1)
    Rf, Pc = host.P2Function(st, "Rf", lower, lower + 1), host.P2Function(st, "Pc", lower, lower + 1)
    ffn = lambda p: np.sin(5.0 * p[:, 0] + 2.0 * p[:, 1]) + p[:, 2] ** 2  # noqa: E731
    cfn = lambda p: np.cos(3.0 * p[:, 1] - p[:, 2]) * (1.0 + p[:, 0])  # noqa: E731
    for cell, (fv, fe) in enumerate(_fields(po, hu, st, lower + 1, ffn)):
        f.upload(lower + 1, fv, fe, cell)
        Rf.upload(lower + 1, fv, fe, cell)
    for cell, (cv, ce) in enumerate(_fields(po, hu, st, lower, cfn)):
        c_.upload(lower, cv, ce, cell)
        Pc.upload(lower, cv, ce, cell)
    host.p2_restrict(Rf, lower + 1, host.All)
    host.p2_prolongate(Pc, lower, host.All)
    lhs = Rf.dot(c_, lower, host.All)
    rhs = f.dot(Pc, lower + 1, host.All)
    assert abs(lhs - rhs) <= 1e-12 * abs(rhs)
    for o in (f, c_, Rf, Pc, st):
        o.close()
