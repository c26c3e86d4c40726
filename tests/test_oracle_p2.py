"""CPU pins of the P2 restatement (oracle/p1_oracle.c, section P2): the element matrix against the reference's generated
form compiled in place (oracle/_ref), index tables against the reference's definitions, and exact known answers of the
P2 Laplace operator (the checks the reference's P2 convergence tests rest on)."""
import ctypes as C

import numpy as np
import pytest

from conftest import OCT_TET, REF_TET, SKEW_TET
from oracle import p1_oracle as po


def test_edge_array_layout_matches_the_reference_formulas():
    """EdgeDoFIndexing.hpp:920-972: blocks X, Y, Z, XY, XZ, YZ of tet(2^L) entries, XYZ of tet(2^L - 1); VertexDoFMemory /
    EdgeDoFMemory sizes"""
    for level in range(0, 6):
        n = 1 << level
        tet = lambda w: w * (w + 1) * (w + 2) // 6  # noqa: E731
        assert po.edge_array_size(level) == 6 * tet(n) + tet(n - 1)
        ec = po.edge_coords(level)
        assert len(ec) == po.edge_array_size(level)
        for k in (0, len(ec) // 3, len(ec) - 1):
            x, y, z, o = (int(v) for v in ec[k])
            assert po.edge_index(level, x, y, z, o) == k


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_inner_edge_dofs_are_the_references_inner_edge_dofs(level):
    """edgedof::macrocell::isInner{X,Y,Z,XY,XZ,YZ,XYZ}EdgeDoF, EdgeDoFIndexing.hpp:987-1020"""
    n = 1 << level
    for (x, y, z, o), cls in zip(po.edge_coords(level), po.edge_classes(level)):
        s = x + y + z
        want = [level > 0 and y > 0 and z > 0 and s < n, level > 0 and x > 0 and z > 0 and s < n, level > 0 and x > 0 and y > 0 and s < n,
                level >= 2 and z > 0 and s < n - 1, level >= 2 and y > 0 and s < n - 1, level >= 2 and x > 0 and s < n - 1,
                level > 0 and s < n - 1][o]
        assert (cls == 14) == bool(want), (x, y, z, o)


def test_every_micro_cell_touches_ten_distinct_valid_dofs_and_every_dof_is_touched():
    level = 3
    n = 1 << level
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    hits_v, hits_e = np.zeros(nv, int), np.zeros(ne, int)
    deficit = [0, 1, 1, 2, 1, 1]
    ncells = 0
    for t in range(6):
        rows = n - deficit[t]
        for z in range(rows):
            for y in range(rows - z):
                for x in range(rows - z - y):
                    idx = po.p2_micro_cell_dofs(level, t, x, y, z)
                    assert len(set(idx[:4])) == 4 and len(set(idx[4:])) == 6
                    assert all(0 <= i < nv for i in idx[:4]) and all(0 <= i < ne for i in idx[4:])
                    hits_v[idx[:4]] += 1
                    hits_e[idx[4:]] += 1
                    ncells += 1
    assert ncells == n ** 3  # numMicroCellsPerMacroCellTotal: a macro-tet splits into 8^L micro-tets
    assert hits_v.min() >= 1 and hits_e.min() >= 1
    assert hits_v[po.slot_of_points(level) == 14].min() == 24 and hits_v[po.slot_of_points(level) == 14].max() == 24


@pytest.mark.parametrize("tet", [REF_TET, OCT_TET, SKEW_TET])
def test_p2_element_matrix_equals_the_references_generated_form(tet):
    ref = po.ref_fenics()
    if ref is None or not hasattr(ref, "ref_p2_tet_diffusion"):
        pytest.skip("oracle/_ref not built (reference not mounted)")
    c = np.ascontiguousarray(tet, dtype=np.float64).reshape(12)
    R = np.empty(100)
    ref.ref_p2_tet_diffusion(R.ctypes.data_as(C.POINTER(C.c_double)), c.ctypes.data_as(C.POINTER(C.c_double)))
    A = po.p2_tet_diffusion(c)
    assert np.abs(A - R.reshape(10, 10)).max() <= 2e-14 * np.abs(R).max()
    assert np.abs(A - A.T).max() <= 1e-15 * np.abs(A).max() and np.abs(A.sum(axis=1)).max() <= 1e-14 * np.abs(A).max()


def _fields(co, level, fn):
    import hostutil as hu

    return fn(hu.cell_points(co, level)), fn(po.edge_midpoints(co, level))


@pytest.mark.parametrize("level", [1, 2, 3])
def test_p2_laplace_known_answers(level):
    """u in P2 is represented exactly: u^T A u = int |grad u|^2; on the reference tetrahedron int 4x^2 = 1/15, int (x^2+y^2) = 1/30"""
    co = np.asarray(REF_TET, dtype=np.float64).reshape(12)
    em = po.p2_cell_element_matrices(co, level)
    for fn, want in ((lambda p: p[:, 0] ** 2, 1.0 / 15.0), (lambda p: p[:, 0] * p[:, 1], 1.0 / 30.0), (lambda p: 1.0 + 0 * p[:, 0], 0.0),
                     (lambda p: 2 * p[:, 0] - p[:, 2], 5.0 / 6.0)):
        uv, ue = _fields(co, level, fn)
        dv, de = po.p2_elementwise_apply_cell(np.zeros_like(uv), np.zeros_like(ue), uv, ue, level, em)
        assert abs(uv @ dv + ue @ de - want) < 1e-13


@pytest.mark.parametrize("tet", [REF_TET, SKEW_TET])
def test_harmonic_quadratics_have_zero_inner_residual_and_masks_are_respected(tet):
    level = 3
    co = np.asarray(tet, dtype=np.float64).reshape(12)
    em = po.p2_cell_element_matrices(co, level)
    uv, ue = _fields(co, level, lambda p: p[:, 0] ** 2 - p[:, 2] ** 2 + p[:, 0] * p[:, 1] - 2 * p[:, 1])
    dv, de = po.p2_elementwise_apply_cell(np.full_like(uv, 7.0), np.full_like(ue, 7.0), uv, ue, level, em, mask=1 << 14)
    iv, ie = po.slot_of_points(level) == 14, po.edge_classes(level) == 14
    scale = max(np.abs(uv).max(), 1.0) * np.abs(em).max()
    assert np.abs(dv[iv]).max() < 1e-13 * scale and np.abs(de[ie]).max() < 1e-13 * scale
    assert np.all(dv[~iv] == 7.0) and np.all(de[~ie] == 7.0)
    # Add mode and alpha
    av, ae = po.p2_elementwise_apply_cell(np.ones_like(uv), np.ones_like(ue), uv, ue, level, em, alpha=-2.0, update=1, mask=0x7FFF)
    fv, fe = po.p2_elementwise_apply_cell(np.zeros_like(uv), np.zeros_like(ue), uv, ue, level, em)
    assert np.allclose(av, 1.0 - 2.0 * fv, rtol=0, atol=1e-12 * scale) and np.allclose(ae, 1.0 - 2.0 * fe, rtol=0, atol=1e-12 * scale)
