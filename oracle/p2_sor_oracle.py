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


# ---- the whole mesh ------------------------------------------------------------------------------------------------------
_CELL_FACES = [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)]
# end points of an edge DoF relative to its logical index, by orientation X, Y, Z, XY, XZ, YZ, XYZ
# (src/hyteg/edgedofspace/EdgeDoFIndexing.hpp: the micro-edge an index and an orientation denote)
_EDGE_ENDS = [((0, 0, 0), (1, 0, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 0, 1)), ((1, 0, 0), (0, 1, 0)),
              ((1, 0, 0), (0, 0, 1)), ((0, 1, 0), (0, 0, 1)), ((0, 1, 0), (1, 0, 1))]


class GlobalSweep:
    """P2ConstantOperator::smooth_sor on a mesh of several macro-cells (P2ConstantOperator.cpp:1267-1330), written on the global
    matrix -- it knows nothing about copies of shared DoFs.  Forward: macro-vertices (:157-201), communicate, macro-edges
    (:205-266: P2::macroedge::smoothSOR3D, P2MacroEdge.cpp:617-680 = the vertex DoFs along the edge, then its edge DoFs),
    communicate, macro-faces (:269-880: sor_3D_macroface_P2_update_vertexdofs = rows ascending, x ascending; then
    ..._update_edgedofs = the same loop with X, XY, Y at every index, each in place), communicate, macro-cells (:913-1260:
    vertex DoFs in array order, then the edge DoFs type by type).  Backwards: the classes in reverse, every loop reversed --
    except that a macro-edge still does its vertex DoFs before its edge DoFs (P2MacroEdge.cpp:617-680 passes `backwards`
    to the two iterators only).
    Which value a sweep sees: the current one if the neighbour lies on the primitive being swept or on its boundary
    (lower-dimensional primitives are communicated upwards before every class), else what the primitive's ghost layers
    hold: the values of the synchronisation at the start (forward; nothing is communicated downwards in between), the
    values at the start of the class (backwards, where every class is preceded by a downward communication).
    A macro-primitive is the set of mesh vertices that span it; vertices of an edge / a face are ordered by id, as HyTeG
    orders them.  Small levels only (the matrix is assembled from unit vectors)."""

    def __init__(self, vertices, cells, level):
        self.vertices, self.mesh_cells, self.level = np.asarray(vertices, float), np.asarray(cells, int), level
        n = 1 << level
        ijk = po.cell_coords(level).astype(int)
        ec = po.edge_coords(level).astype(int)
        self.nv, self.ne = len(ijk), len(ec)
        face_count = {}
        for cv in self.mesh_cells:
            for f in _CELL_FACES:
                k = tuple(sorted(int(cv[a]) for a in f))
                face_count[k] = face_count.get(k, 0) + 1
        bfaces = [set(f) for f, c in face_count.items() if c == 1]
        # doubled index coordinates of every DoF of a cell, in array order (vertex DoFs, then edge DoFs)
        p2 = [2 * p for p in ijk]
        for x, y, z, o in ec:
            a, b = _EDGE_ENDS[o]
            p2.append(np.array([2 * x + a[0] + b[0], 2 * y + a[1] + b[1], 2 * z + a[2] + b[2]]))
        self.index, self.support, self.order, self.is_edge, self.boundary, self.gidx = {}, [], [], [], [], []
        for c, cv in enumerate(self.mesh_cells):
            g = np.empty(self.nv + self.ne, dtype=np.int64)
            for k, q in enumerate(p2):
                bary = (2 * n - int(q.sum()), int(q[0]), int(q[1]), int(q[2]))
                key = tuple(sorted((int(cv[a]), bary[a]) for a in range(4) if bary[a] > 0))
                if key not in self.index:
                    self.index[key] = len(self.support)
                    sup, wt, e = tuple(v for v, _ in key), dict(key), k >= self.nv
                    self.is_edge.append(e)
                    self.boundary.append(any(set(sup) <= f for f in bfaces))
                    if len(sup) == 1:
                        order = (0,)
                    elif len(sup) == 2:
                        order = (int(e), wt[sup[1]])
                    elif len(sup) == 3:
                        wb, wc = wt[sup[1]], wt[sup[2]]
                        # face frame: x towards the middle vertex, y towards the last; X: (odd, even), XY: (odd, odd), Y: (even, odd)
                        order = (int(e), wc // 2, wb // 2, (0 if wc % 2 == 0 else (1 if wb % 2 else 2)) if e else 0)
                    else:
                        sup = ("cell", c) + sup
                        order = (0, k) if not e else (1 + int(ec[k - self.nv][3]), k)
                    self.support.append(sup)
                    self.order.append(order)
                g[k] = self.index[key]
            self.gidx.append(g)
        self.ndof = len(self.support)
        rows, cols, vals = [], [], []
        for c, cv in enumerate(self.mesh_cells):
            A = assemble_cell_matrix(self.vertices[cv].reshape(12), level).tocoo()
            g = self.gidx[c]
            rows.append(g[A.row]), cols.append(g[A.col]), vals.append(A.data)
        self.A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.ndof, self.ndof)).tocsr()
        self.A.sum_duplicates()
        self.diag = self.A.diagonal()

    def dim(self, p):
        s = self.support[p]
        return 4 if s[0] == "cell" else len(s)

    def closure_contains(self, p, j):
        sp_, sj = self.support[p], self.support[j]
        if sp_[0] == "cell":
            return sj[0] != "cell" or sj[1] == sp_[1]
        return sj[0] != "cell" and set(sj) <= set(sp_)

    def to_global(self, cell_arrays):
        """cell_arrays[c] = (vertex array, edge array) of cell c -> global vector (copies must agree)"""
        u = np.zeros(self.ndof)
        for g, (v, e) in zip(self.gidx, cell_arrays):
            u[g] = np.concatenate([v, e])
        return u

    def to_cells(self, u):
        return [(u[g[:self.nv]].copy(), u[g[self.nv:]].copy()) for g in self.gidx]

    def sweep(self, u, b, relax=1.0, backwards=False, dirichlet=True):
        u = u.copy()
        snap = u.copy()
        A, diag = self.A, self.diag
        for dm in ([4, 3, 2, 1] if backwards else [1, 2, 3, 4]):
            if backwards:
                snap = u.copy()
            prims = {}
            for p in range(self.ndof):
                if self.dim(p) == dm and not (dirichlet and self.boundary[p]):
                    prims.setdefault(self.support[p], []).append(p)
            for sup, pts in prims.items():
                if dm == 2 and backwards:
                    pts.sort(key=lambda p: (self.order[p][0], -self.order[p][1]))
                else:
                    pts.sort(key=lambda p: self.order[p], reverse=backwards)
                for p in pts:
                    tmp = b[p]
                    for q in range(A.indptr[p], A.indptr[p + 1]):
                        j = A.indices[q]
                        if j != p:
                            tmp -= A.data[q] * (u[j] if self.closure_contains(p, j) else snap[j])
                    u[p] = (1.0 - relax) * u[p] + relax * tmp / diag[p]
        return u
