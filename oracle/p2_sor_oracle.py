"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN section 4): CPU restatement of the reference's P2 Gauss-Seidel / SOR
sweep on ONE macro-cell with Dirichlet data on the cell boundary.

Reference: P2ConstantOperator::smooth_sor_macro_cells (src/constant_stencil_operator/P2ConstantOperator.cpp:913-1200) calls
  forward:   sor_3D_macrocell_P2_update_vertexdofs, then sor_3D_macrocell_P2_update_edgedofs_by_type_{X,Y,Z,XY,XZ,YZ,XYZ}
  backwards: the edge types in the reverse order, then sor_3D_macrocell_P2_update_vertexdofs_backwards
(src/constant_stencil_operator/P2generatedKernels/sor_3D_macrocell_P2_update_*.cpp).  Every kernel loops lexicographically
(z, y, x) over the inner DoFs of its kind and sets
    u_i = (1 - relax) u_i + relax / a_ii ( rhs_i - sum_{j != i} a_ij u_j )
with the current values of all other DoFs.  The rows a_i. are the constant stencils the reference assembles from the element
matrices; here they are read off the oracle's literal restatement of P2ElementwiseOperator::gemv (ho_p2_elementwise_apply_cell)
applied to unit vectors -- the same operator, assembled as a sparse matrix (small levels only)."""
import numpy as np
import scipy.sparse as sp

from . import p1_oracle as po


def assemble_cell_matrix(coords12, level):
    """sparse (nv + ne) x (nv + ne) matrix of the P2 Laplace operator on one macro-cell, all DoFs (rows of boundary DoFs included)"""
    nv, ne = po.cell_size(level), po.edge_array_size(level)
    em = po.p2_cell_element_matrices(np.asarray(coords12, dtype=np.float64).reshape(12), level)
    cols = []
    for j in range(nv + ne):
        sv, se = np.zeros(nv), np.zeros(ne)
        (sv if j < nv else se)[j if j < nv else j - nv] = 1.0
        ov, oe = po.p2_elementwise_apply_cell(np.zeros(nv), np.zeros(ne), sv, se, level, em, 1.0, 0, 0x7FFF)
        cols.append(sp.csc_matrix(np.concatenate([ov, oe]).reshape(-1, 1)))
    return sp.hstack(cols).tocsr()


def sor_cell(A, uv, ue, bv, be, level, relax, backwards=False):
    """one sweep over the INNER DoFs of the cell (point class 14), in place on copies; returns (uv, ue)"""
    nv = po.cell_size(level)
    u = np.concatenate([uv, ue]).astype(np.float64)
    b = np.concatenate([bv, be]).astype(np.float64)
    inner_v = np.flatnonzero(po.slot_of_points(level) == 14)  # array order = lexicographic (z, y, x)
    cls_e = po.edge_classes(level)
    n = 1 << level
    # edge array: blocks X, Y, Z, XY, XZ, YZ (width n) and XYZ (width n - 1), each lexicographic
    tet = lambda w: w * (w + 1) * (w + 2) // 6  # noqa: E731
    starts = [k * tet(n) for k in range(6)] + [6 * tet(n), 6 * tet(n) + tet(n - 1)]
    diag = A.diagonal()

    def update(i):
        lo, hi = A.indptr[i], A.indptr[i + 1]
        s = b[i] - (A.data[lo:hi] @ u[A.indices[lo:hi]] - diag[i] * u[i])
        u[i] = (1.0 - relax) * u[i] + relax * s / diag[i]

    def vertices():
        for i in (inner_v[::-1] if backwards else inner_v):
            update(int(i))

    def edges(t):
        idx = np.arange(starts[t], starts[t + 1])
        idx = idx[cls_e[idx] == 14]
        for i in (idx[::-1] if backwards else idx):
            update(nv + int(i))

    if not backwards:
        vertices()
        for t in range(7):
            edges(t)
    else:
        for t in range(6, -1, -1):
            edges(t)
        vertices()
    return u[:nv], u[nv:]
