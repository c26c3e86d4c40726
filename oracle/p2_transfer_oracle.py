"""TEST INFRASTRUCTURE (CPU oracle), NOT PRODUCT CODE: only tests/ may import this.

Quadratic (P2) grid transfer on one macro-cell, restated from its definition the way the reference's kernels are organised:
  src/hyteg/gridtransferoperators/P2toP2QuadraticProlongation.cpp:217-424 (prolongateAdditively3D) with
  generatedKernels/prolongate_3D_macrocell_P2_push_from_{vertexdofs,edgedofs}.cpp: a PUSH over the coarse grid -- every
  coarse DoF adds its hierarchical-basis weights (1, 3/8, -1/8, 3/4, 1/2, 1/4) into the fine DoFs of its support, scaled by
  1 / (number of macro-cells sharing the macro-primitive the fine DoF lies on) so that the additive communication that follows
  completes the value;
  P2toP2QuadraticRestriction.cpp:131-286 (restrictAdditively3D) with generatedKernels/restrict_3D_macrocell_P2_update_
  {vertexdofs,edgedofs}.cpp: the transpose.
Here the push runs over the coarse MICRO-CELLS: each one evaluates its quadratic interpolant (ten shape functions of its ten
coarse DoFs) at the fine DoFs it contains; a fine DoF contained in several micro-cells receives the same value from each
(P2 functions are continuous), so the oracle stores the mean of the contributions.  Weights come out of the shape functions
lambda_i (2 lambda_i - 1) and 4 lambda_i lambda_j -- nothing is transcribed from the generated kernels.  The micro-cell ->
DoF index map is the P2 oracle's (oracle/p1_oracle.c ho_p2_micro_cell_dofs, pinned by the elementwise-operator tests).

Pinned (tests/test_oracle_p2_transfer.py) by the reference's own known answers: supports of a coarse vertex / edge DoF
(tests/hyteg/P2/P2QuadraticProlongation3DTest.cpp:84,157), exactness on constants, linears and quadratics (:163-255),
restriction of the constant one (tests/hyteg/P2/P2QuadraticRestriction3DTest.cpp:50-88)."""
from __future__ import annotations

import functools
import itertools

import numpy as np

from . import p1_oracle as po

# micro-vertices of the six micro-cell types relative to the cell index (volumedofspace/CellDoFIndexing.hpp:155-198), in
# the order the P2 oracle uses: WHITE_UP, BLUE_UP, GREEN_UP, WHITE_DOWN, BLUE_DOWN, GREEN_DOWN
MICRO_VERTS = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], [[1, 0, 0], [1, 1, 0], [0, 1, 0], [1, 0, 1]],
                        [[1, 0, 0], [0, 1, 0], [1, 0, 1], [0, 0, 1]], [[1, 1, 0], [1, 1, 1], [0, 1, 1], [1, 0, 1]],
                        [[1, 0, 1], [0, 1, 1], [0, 0, 1], [0, 1, 0]], [[0, 1, 0], [1, 1, 0], [1, 0, 1], [0, 1, 1]]])
# end points of an edge DoF relative to its logical index: X, Y, Z, XY, XZ, YZ, XYZ (edgedofspace/EdgeDoFIndexing.hpp)
EDGE_ENDS = np.array([[[0, 0, 0], [1, 0, 0]], [[0, 0, 0], [0, 1, 0]], [[0, 0, 0], [0, 0, 1]], [[1, 0, 0], [0, 1, 0]],
                      [[1, 0, 0], [0, 0, 1]], [[0, 1, 0], [0, 0, 1]], [[0, 1, 0], [1, 0, 1]]])


def _edge_dof(level, a, b):
    """index in the edge-DoF array of the micro-edge with end points a, b (integer micro-vertex coordinates), or None"""
    a, b = np.asarray(a), np.asarray(b)
    for o in range(7):
        d = EDGE_ENDS[o][1] - EDGE_ENDS[o][0]
        for p, q in ((a, b), (b, a)):
            if np.array_equal(q - p, d):
                base = p - EDGE_ENDS[o][0]
                n = 1 << level
                w = n - 1 if o == 6 else n
                if base.min() < 0 or base.sum() > w - 1:
                    return None
                return po.edge_index(level, int(base[0]), int(base[1]), int(base[2]), o)
    raise AssertionError(f"not a micro-edge: {a} {b}")


def micro_cells(level):
    """(type, x, y, z) of every micro-cell of the macro-cell at `level`"""
    n = 1 << level
    out = []
    for t in range(6):
        for z in range(n):
            for y in range(n - z):
                for x in range(n - z - y):
                    if (MICRO_VERTS[t] + np.array([x, y, z])).sum(axis=1).max() <= n:
                        out.append((t, x, y, z))
    return out


def _fine_points_of_micro_cell():
    """barycentric coordinates (in quarters) of the fine DoFs a coarse micro-cell contains: the 10 fine vertices (multiples
    of 1/2) and the midpoints of the fine micro-edges between them that lie in the closed cell"""
    verts = [np.array(c) for c in itertools.product(range(0, 5, 2), repeat=4) if sum(c) == 4]
    assert len(verts) == 10
    edges = []
    for i, j in itertools.combinations(range(10), 2):
        d = verts[j] - verts[i]
        if sorted(np.abs(d)) in ([0, 0, 2, 2], [2, 2, 2, 2]):  # candidates: half a coarse edge apart, or opposite mid-edge points
            edges.append((verts[i], verts[j]))
    # the four fine vertices at the coarse vertices are joined to the mid-edge points only; the six mid-edge points are joined
    # among themselves when they share a coarse face: 12 + 12 = 24 fine edges on the coarse faces, plus ONE interior edge:
    # of the three pairs of opposite mid-edge points only the one the refinement rule connects (the fine grid has the
    # same six micro-cell types) is a fine micro-edge; it is found by its direction below
    return verts, edges


@functools.lru_cache(maxsize=None)
def prolongation_matrix(level):
    """sparse P (scipy CSR): (fine vertex DoFs, then fine edge DoFs at level + 1) x (coarse vertex DoFs, then coarse edge DoFs
    at `level`) of one macro-cell"""
    import scipy.sparse as sp

    fl = level + 1
    nvc, nec = po.cell_size(level), po.edge_array_size(level)
    nvf, nef = po.cell_size(fl), po.edge_array_size(fl)
    rows, cols, vals = [], [], []
    cnt = np.zeros(nvf + nef)
    verts4, edges4 = _fine_points_of_micro_cell()
    valid_dirs = {tuple(EDGE_ENDS[o][1] - EDGE_ENDS[o][0]) for o in range(7)} | {tuple(EDGE_ENDS[o][0] - EDGE_ENDS[o][1]) for o in range(7)}
    pairs = list(itertools.combinations(range(4), 2))
    for t, x, y, z in micro_cells(level):
        dofs = po.p2_micro_cell_dofs(level, t, x, y, z)  # 4 vertex DoFs, then 6 edge DoFs
        V = (MICRO_VERTS[t] + np.array([x, y, z])) * 2  # the cell's vertices in FINE micro-vertex coordinates
        # column of every local DoF: vertices, then the edge of each vertex pair (identified through its edge index)
        col = list(dofs[:4])
        for (i, j) in pairs:
            e = _edge_dof(level, V[i] // 2, V[j] // 2)
            assert e in dofs[4:]
            col.append(nvc + e)

        def shape(lam4):
            lam = np.asarray(lam4, dtype=np.float64) / 4.0
            return [lam[i] * (2.0 * lam[i] - 1.0) for i in range(4)] + [4.0 * lam[i] * lam[j] for (i, j) in pairs]

        def push(row, lam4):
            for c_, w in zip(col, shape(lam4)):
                if w != 0.0:
                    rows.append(row), cols.append(c_), vals.append(w)
            cnt[row] += 1

        for lam4 in verts4:
            p = (lam4[:, None] * V).sum(axis=0) // 4
            push(po.cell_index(fl, int(p[0]), int(p[1]), int(p[2])), lam4)
        for la, lb in edges4:
            pa, pb = (la[:, None] * V).sum(axis=0) // 4, (lb[:, None] * V).sum(axis=0) // 4
            if tuple(pb - pa) not in valid_dirs:
                continue  # the two opposite mid-edge pairs the refinement does not connect
            push(nvf + _edge_dof(fl, pa, pb), (la + lb) / 2.0)
    assert cnt.min() >= 1, "every fine DoF lies in some coarse micro-cell"
    vals = np.asarray(vals) / cnt[np.asarray(rows)]  # mean over the micro-cells that contain the fine DoF
    return sp.csr_matrix((vals, (rows, cols)), shape=(nvf + nef, nvc + nec))


def prolongate_cell(coarse_v, coarse_e, level):
    """fine vertex / edge arrays at level + 1 of the quadratic interpolant of the coarse function (one macro-cell)"""
    f = prolongation_matrix(level) @ np.concatenate([coarse_v, coarse_e])
    nvf = po.cell_size(level + 1)
    return f[:nvf], f[nvf:]


def restrict_cell(fine_v, fine_e, level, nnc=None):
    """coarse arrays at level - 1 = P^T applied to the fine function; nnc (14 values, slot order { edge0..5, face0..3,
    vertex0..3 }): number of macro-cells sharing each macro-primitive of this cell -- every fine DoF on such a primitive is
    scaled by 1 / nnc, so that summing the results of all cells counts it once (restrictAdditively3D)"""
    cl = level - 1
    f = np.concatenate([fine_v, fine_e]).astype(np.float64)
    if nnc is not None:
        inv = np.concatenate([1.0 / np.asarray(nnc, dtype=np.float64), [1.0]])
        f = f * np.concatenate([inv[po.slot_of_points(level)], inv[po.edge_classes(level)]])
    c = prolongation_matrix(cl).T @ f
    nv = po.cell_size(cl)
    return c[:nv], c[nv:]
