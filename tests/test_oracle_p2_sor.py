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


# ---- the sweep over a whole mesh (oracle/p2_sor_oracle.GlobalSweep) -------------------------------------------------------
def _octahedron():
    import hostutil as hu

    return hu.read_msh(hu.MESHES / "regular_octahedron_8el.msh")


def test_global_sweep_on_one_cell_is_the_cell_sweep():
    """one macro-cell with Dirichlet values: every DoF on the cell boundary is fixed, the global sweep IS the macro-cell sweep"""
    level = 2
    co = np.asarray(SKEW_TET, dtype=np.float64).reshape(4, 3)
    G = ps.GlobalSweep(co, np.array([[0, 1, 2, 3]]), level)
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    assert G.ndof == nv + ne
    rng = np.random.default_rng(5)
    uv, ue, bv, be = rng.standard_normal(nv), rng.standard_normal(ne), rng.standard_normal(nv), rng.standard_normal(ne)
    A = ps.assemble_cell_matrix(co.reshape(12), level)
    for backwards in (False, True):
        for relax in (1.0, 0.7):
            v, e = ps.sor_cell(A, uv, ue, bv, be, level, relax, backwards)
            g = G.sweep(G.to_global([(uv, ue)]), G.to_global([(bv, be)]), relax, backwards)
            gv, ge = G.to_cells(g)[0]
            assert np.abs(gv - v).max() < 1e-13 and np.abs(ge - e).max() < 1e-13  # the two sum a row in different orders


def test_global_sweep_properties_on_eight_cells():
    """the assembled operator is the P2 Laplacian of the mesh (symmetric, zero inner residual on a harmonic quadratic), a
    solution is a fixed point of the sweep in both directions, boundary DoFs stay, alternating sweeps converge"""
    level = 2
    v, c = _octahedron()
    G = ps.GlobalSweep(v, c, level)
    assert abs(G.A - G.A.T).max() < 1e-14
    n = 1 << level
    X = np.zeros((G.ndof, 3))
    for key, i in G.index.items():
        X[i] = sum(np.asarray(v[a]) * w for a, w in key) / (2 * n)
    q = X[:, 0] ** 2 - X[:, 1] ** 2 + 0.3 * X[:, 0] * X[:, 2]
    inner = ~np.array(G.boundary)
    assert np.abs((G.A @ q)[inner]).max() < 1e-13
    b = G.A @ q
    for backwards in (False, True):
        assert np.abs(G.sweep(q, b, 1.0, backwards) - q).max() < 1e-13
    rng = np.random.default_rng(6)
    u0, rhs = rng.standard_normal(G.ndof), rng.standard_normal(G.ndof)
    u = u0.copy()
    r0 = np.abs((rhs - G.A @ u)[inner]).max()
    for k in range(40):
        u = G.sweep(u, rhs, 1.0, backwards=bool(k & 1))
    assert np.array_equal(u[~inner], u0[~inner])
    assert np.abs((rhs - G.A @ u)[inner]).max() < 0.05 * r0
    # every class of macro-primitive has inner DoFs on this mesh (the centre vertex, 6 inner macro-edges, 12 inner macro-faces)
    dims = np.array([G.dim(p) for p in range(G.ndof)])
    assert [int(((dims == d) & inner).sum()) for d in (1, 2, 3, 4)] == [1, 42, 252, 280]


def test_closure_split_and_face_weights_of_the_operator_table():
    """host-side pieces of the C-ABI for the shared-primitive sweeps (no GPU needed): the three parts of the closure split add up
    to the boundary-class rows, and the couplings between the edge DoFs inside a macro-face, read off the table in the FACE's
    frame, are the entries of the oracle's cell matrix between exactly those DoFs -- for every ordering of the face's vertices"""
    import itertools

    from hyteg_amd import capi

    level = 2
    n = 1 << level
    co = np.asarray(SKEW_TET, dtype=np.float64).reshape(12)
    em = po.p2_cell_element_matrices(co, level)
    table = capi.p2_build_operator_table(em)
    out, cv, ce = capi.p2_operator_table_closure_split(table)
    total = np.asarray(out) + np.asarray(cv) + np.asarray(ce)
    nz = np.flatnonzero(total)
    assert nz.size > 0 and np.array_equal(total[nz], np.asarray(table)[nz])
    assert np.count_nonzero(cv) > 0 and np.count_nonzero(ce) > 0 and np.count_nonzero(out) > 0
    A = ps.assemble_cell_matrix(co, level).tocsr()
    nv = po.cell_size(level)
    unit = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    ends = ps._EDGE_ENDS

    def edge_dof(p, q):
        """array index of the edge DoF between the micro-vertices p and q of the cell"""
        for o, (a, b) in enumerate(ends):
            for s, e in ((p, q), (q, p)):
                base = s - np.asarray(a)
                if np.array_equal(base + np.asarray(b), e) and base.min() >= 0:
                    return nv + po.edge_index(level, int(base[0]), int(base[1]), int(base[2]), o)
        raise KeyError((p, q))

    nb = {0: [(1, 0, 0), (2, 0, 0), (1, 0, -1), (2, 1, -1)], 1: [(0, 0, 0), (2, 0, 0), (0, 0, 1), (2, 1, 0)],
          2: [(0, 0, 0), (1, 0, 0), (0, -1, 1), (1, -1, 0)]}
    for face in ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)):
        for lv in itertools.permutations(face):
            w = capi.p2_operator_table_face_edge_weights(table, lv)
            O, a, b = n * unit[lv[0]], unit[lv[1]] - unit[lv[0]], unit[lv[2]] - unit[lv[0]]
            P = lambda i, j: O + i * a + j * b  # noqa: E731

            def dof(t, i, j):
                return edge_dof(P(i, j), P(i + 1, j)) if t == 0 else (edge_dof(P(i + 1, j), P(i, j + 1)) if t == 1 else edge_dof(P(i, j), P(i, j + 1)))

            for t in range(3):
                i, j = 1, 1  # an edge DoF inside the face with all its neighbours, at level 2
                d = dof(t, i, j)
                assert abs(w[t][0] - A[d, d]) < 1e-13
                for k, (t2, di, dj) in enumerate(nb[t]):
                    assert abs(w[t][1 + k] - A[d, dof(t2, i + di, j + dj)]) < 1e-13, (lv, t, k)
