"""CPU pins of oracle/p2_sor_oracle.py (the restatement of the reference's P2 macro-cell Gauss-Seidel sweep,
sor_3D_macrocell_P2_update_{vertexdofs,edgedofs_by_type}*): the assembled matrix is the operator of the apply oracle
(symmetric, annihilates quadratics' inner residual exactly as P2ElementwiseOperator does), the DoFs of one edge type do not
couple (what makes the reference's per-type sweeps order-free), and swept to convergence the inner equations hold."""
import numpy as np

from conftest import SKEW_TET
from oracle import p1_oracle as po
from oracle import p2_sor_oracle as ps


def test_matrix_is_the_apply_oracle_and_edge_types_do_not_couple():
    level = 2
    co = np.asarray(SKEW_TET, dtype=np.float64).reshape(12)
    A = ps.assemble_cell_matrix(co, level)
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    assert A.shape == (nv + ne, nv + ne)
    assert abs(A - A.T).max() < 1e-14
    rng = np.random.default_rng(1)
    sv, se = rng.standard_normal(nv), rng.standard_normal(ne)
    em = po.p2_cell_element_matrices(co, level)
    ov, oe = po.p2_elementwise_apply_cell(np.zeros(nv), np.zeros(ne), sv, se, level, em, 1.0, 0, 0x7FFF)
    assert np.abs(A @ np.concatenate([sv, se]) - np.concatenate([ov, oe])).max() < 1e-12
    # no two DoFs of one edge type couple (a micro-cell has one edge per type)
    n = 1 << level
    tet = lambda w: w * (w + 1) * (w + 2) // 6  # noqa: E731
    starts = [k * tet(n) for k in range(6)] + [6 * tet(n), 6 * tet(n) + tet(n - 1)]
    for t in range(7):
        blk = A[nv + starts[t]:nv + starts[t + 1], nv + starts[t]:nv + starts[t + 1]].tocoo()
        assert np.all(blk.row[np.abs(blk.data) > 0] == blk.col[np.abs(blk.data) > 0])


def test_sweeps_converge_to_the_inner_equations():
    level = 2
    co = np.asarray(SKEW_TET, dtype=np.float64).reshape(12)
    A = ps.assemble_cell_matrix(co, level)
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    rng = np.random.default_rng(2)
    uv, ue, bv, be = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
    u0 = np.concatenate([uv, ue])
    inner = np.concatenate([po.slot_of_points(level) == 14, po.edge_classes(level) == 14])
    # relax = 0: nothing changes; forward then backward sweeps converge
    v, e = ps.sor_cell(A, uv, ue, bv, be, level, 0.0)
    assert np.array_equal(np.concatenate([v, e]), u0)
    for k in range(200):
        uv, ue = ps.sor_cell(A, uv, ue, bv, be, level, 1.0, backwards=bool(k & 1))
    u = np.concatenate([uv, ue])
    assert np.array_equal(u[~inner], u0[~inner])  # Dirichlet data untouched
    res = (np.concatenate([bv, be]) - A @ u)[inner]
    assert np.abs(res).max() < 1e-10
