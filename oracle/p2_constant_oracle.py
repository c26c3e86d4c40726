"""CPU restatement (TEST INFRASTRUCTURE: only tests/ may import this) of the constant-stencil P2 operator on one macro-cell.

Follows, under /root/reference/src/:
  * stencil assembly: hyteg/p2functionspace/P2Elements3D.hpp:185-420 (calculateEdgeToVertexStencilInMacroCell,
    calculateVertexToEdgeStencilInMacroCell, calculateEdgeToEdgeStencilInMacroCell: loops over the micro-cells around a
    micro-vertex / micro-edge that lie inside the macro-cell, one entry of the local P2 matrix per (centre, leaf) pair, keyed by
    the leaf's index offset) and hyteg/p1functionspace/P1Elements.hpp:303-380 for the vertex-to-vertex block;
  * the element lists hyteg/p1functionspace/P1Elements.hpp:93-143 (allCellsAtInnerVertex) and getNeighboringElements :215-301
    (restated as "the elements whose vertices all lie in the macro-cell");
  * edge-DoF index conventions hyteg/edgedofspace/EdgeDoFIndexing.hpp:89-210 (calcEdgeDoFOrientation, calcEdgeDoFIndex,
    calcNeighboringVertexDoFIndices) and the inner-DoF predicates :987-1044;
  * the macro-cell apply loops of the four sub-operators: mixedoperators/VertexDoFToEdgeDoFOperator/VertexDoFToEdgeDoFApply.hpp:347-430,
    mixedoperators/EdgeDoFToVertexDoFOperator/EdgeDoFToVertexDoFApply.hpp:483-521 and their edge-to-edge / vertex-to-vertex
    analogues, composed as P2ConstantOperator::apply, constant_stencil_operator/P2ConstantOperator.cpp:100-112.
The local matrix comes from oracle.p1_oracle.p2_tet_diffusion (pinned to the reference's p2_tet_diffusion.h through oracle/_ref)
with the FEniCS dof map fenics::P2DoFMap (hyteg/fenics/fenics.hpp:120-124).  Pure Python: small levels only."""
from __future__ import annotations

import numpy as np

from . import p1_oracle as po

# stencil directions as index offsets; allCellsAtInnerVertex: the 24 micro-tetrahedra around an inner micro-vertex
_D = {"C": (0, 0, 0), "W": (-1, 0, 0), "E": (1, 0, 0), "N": (0, 1, 0), "S": (0, -1, 0), "NW": (-1, 1, 0), "SE": (1, -1, 0),
      "TC": (0, 0, 1), "TW": (-1, 0, 1), "TS": (0, -1, 1), "TSE": (1, -1, 1), "BC": (0, 0, -1), "BN": (0, 1, -1), "BE": (1, 0, -1),
      "BNW": (-1, 1, -1)}
_ELEMENTS = [("C", "BC", "BE", "BN"), ("C", "S", "SE", "TS"), ("C", "W", "NW", "TW"), ("C", "N", "E", "TC"), ("C", "W", "BC", "S"),
             ("C", "E", "SE", "BE"), ("C", "N", "NW", "BN"), ("C", "TS", "TC", "TW"), ("C", "BC", "BN", "BNW"), ("C", "W", "S", "TS"),
             ("C", "E", "SE", "TSE"), ("C", "NW", "N", "TC"), ("C", "BC", "S", "SE"), ("C", "W", "NW", "BNW"), ("C", "E", "BN", "N"),
             ("C", "TC", "TS", "TSE"), ("C", "W", "BC", "BNW"), ("C", "E", "BE", "BN"), ("C", "TC", "TW", "NW"), ("C", "SE", "TS", "TSE"),
             ("C", "BC", "BE", "SE"), ("C", "BN", "BNW", "NW"), ("C", "E", "TSE", "TC"), ("C", "W", "TS", "TW")]
P2_DOF_MAP = [[0, 9, 8, 7], [9, 1, 6, 5], [8, 6, 2, 4], [7, 5, 4, 3]]
X, Y, Z, XY, XZ, YZ, XYZ = range(7)
ORIENTATIONS = [X, Y, Z, XY, XZ, YZ, XYZ]
NEIGHBOR_VERTICES = {X: ((0, 0, 0), (1, 0, 0)), Y: ((0, 0, 0), (0, 1, 0)), Z: ((0, 0, 0), (0, 0, 1)), XY: ((1, 0, 0), (0, 1, 0)),
                     XZ: ((1, 0, 0), (0, 0, 1)), YZ: ((0, 1, 0), (0, 0, 1)), XYZ: ((0, 1, 0), (1, 0, 1))}


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def edge_orientation(a, b):
    """edgedof::calcEdgeDoFOrientation"""
    d = tuple(abs(v) for v in _sub(b, a))
    return {(1, 0, 0): X, (0, 1, 0): Y, (0, 0, 1): Z, (1, 1, 0): XY, (1, 0, 1): XZ, (0, 1, 1): YZ, (1, 1, 1): XYZ}[d]


def edge_index(a, b):
    """edgedof::calcEdgeDoFIndex: the logical index of the edge DoF between the micro-vertices a and b"""
    o = edge_orientation(a, b)
    if o == X:
        return a if a[0] < b[0] else b
    if o == Y:
        return a if a[1] < b[1] else b
    if o == Z:
        return a if a[2] < b[2] else b
    if o == XY:
        lo = a if a[0] < b[0] else b
        return (lo[0], lo[1] - 1, lo[2])
    if o == XZ:
        lo = a if a[0] < b[0] else b
        return (lo[0], lo[1], lo[2] - 1)
    if o == YZ:
        lo = a if a[1] < b[1] else b
        return (lo[0], lo[1], lo[2] - 1)
    lo = a if a[0] < b[0] else b
    return (lo[0], lo[1] - 1, lo[2])


def _inside(p, N):
    return p[0] >= 0 and p[1] >= 0 and p[2] >= 0 and p[0] + p[1] + p[2] <= N - 1


def neighboring_elements(idx, level):
    """P1Elements3D::getNeighboringElements: the elements at micro-vertex idx that lie inside the macro-cell"""
    N = (1 << level) + 1
    return [e for e in _ELEMENTS if all(_inside(_add(idx, _D[d]), N) for d in e)]


def _element_matrix(cell, level, vertices):
    c = np.array([po.coordinate_from_index(cell, level, *v) for v in vertices])
    return po.p2_tet_diffusion(c)


def _edge_with_orientation(offsets, o):
    """edgeWithOrientationFromElement: the (sorted) local vertex ids spanning the element's edge of orientation o, if any"""
    for v0 in range(4):
        for v1 in range(v0):
            if edge_orientation(offsets[v0], offsets[v1]) == o:
                return (v1, v0)
    return None


def vertex_to_vertex_stencil(idx, cell, level):
    out = {}
    for e in neighboring_elements(idx, level):
        offs = [_D[d] for d in e]
        M = _element_matrix(cell, level, [_add(idx, o) for o in offs])
        for k in range(4):
            out[offs[k]] = out.get(offs[k], 0.0) + M[0][k]
    return out


def edge_to_vertex_stencil(idx, leaf, cell, level):
    out = {}
    for e in neighboring_elements(idx, level):
        offs = [_D[d] for d in e]
        edge = _edge_with_orientation(offs, leaf)
        if edge is None:
            continue
        M = _element_matrix(cell, level, [_add(idx, o) for o in offs])
        key = edge_index(offs[edge[0]], offs[edge[1]])
        out[key] = out.get(key, 0.0) + M[P2_DOF_MAP[0][0]][P2_DOF_MAP[edge[0]][edge[1]]]
    return out


def _elements_at_edge(idx, center, level):
    v0, v1 = NEIGHBOR_VERTICES[center]
    second = _sub(v1, v0)
    base = _add(idx, v0)
    for e in neighboring_elements(base, level):
        offs = [_D[d] for d in e]
        if second in offs:
            yield v0, base, offs


def vertex_to_edge_stencil(idx, center, cell, level):
    out = {}
    for v0, base, offs in _elements_at_edge(idx, center, level):
        M = _element_matrix(cell, level, [_add(base, o) for o in offs])
        ce = _edge_with_orientation(offs, center)
        row = P2_DOF_MAP[ce[0]][ce[1]]
        for k in range(4):
            key = _add(v0, offs[k])
            out[key] = out.get(key, 0.0) + M[row][P2_DOF_MAP[k][k]]
    return out


def edge_to_edge_stencil(idx, center, leaf, cell, level):
    out = {}
    for v0, base, offs in _elements_at_edge(idx, center, level):
        le = _edge_with_orientation(offs, leaf)
        if le is None:
            continue
        M = _element_matrix(cell, level, [_add(base, o) for o in offs])
        ce = _edge_with_orientation(offs, center)
        key = edge_index(_add(v0, offs[le[0]]), _add(v0, offs[le[1]]))
        out[key] = out.get(key, 0.0) + M[P2_DOF_MAP[ce[0]][ce[1]]][P2_DOF_MAP[le[0]][le[1]]]
    return out


def is_inner_edge(level, idx, o):
    """edgedof::macrocell::isInner{X,...,XYZ}EdgeDoF, EdgeDoFIndexing.hpp:987-1020"""
    n, s = 1 << level, sum(idx)
    x, y, z = idx
    if o == X:
        return level > 0 and y > 0 and z > 0 and s < n
    if o == Y:
        return level > 0 and x > 0 and z > 0 and s < n
    if o == Z:
        return level > 0 and x > 0 and y > 0 and s < n
    if o == XY:
        return level >= 2 and z > 0 and s < n - 1
    if o == XZ:
        return level >= 2 and y > 0 and s < n - 1
    if o == YZ:
        return level >= 2 and x > 0 and s < n - 1
    return level > 0 and s < n - 1


def reference_position(level, kind):
    """an inner DoF of the kind (0 vertex, 1..7 edge X..XYZ): the first one in array order"""
    if kind == 0:
        return (1, 1, 1)
    for z in range(3):
        for y in range(3):
            for x in range(3):
                if is_inner_edge(level, (x, y, z), kind - 1):
                    return (x, y, z)
    raise ValueError("no inner edge DoF")


def stencils_at(cell, level, positions=None):
    """the four stencil maps assembled at the given DoF positions (default: inner ones) -- also valid for DoFs on the macro-cell's
    boundary, where the maps hold this cell's share (fewer entries)"""
    pos = {k: reference_position(level, k) for k in range(8)}
    if positions:
        pos.update(positions)
    v2v = vertex_to_vertex_stencil(pos[0], cell, level)
    e2v = {o: edge_to_vertex_stencil(pos[0], o, cell, level) for o in ORIENTATIONS}
    v2e = {o: vertex_to_edge_stencil(pos[o + 1], o, cell, level) for o in ORIENTATIONS}
    e2e = {c: {l: edge_to_edge_stencil(pos[c + 1], c, l, cell, level) for l in ORIENTATIONS} for c in ORIENTATIONS}
    return v2v, e2v, v2e, e2e


def inner_stencils(cell, level):
    assert level >= 2
    return stencils_at(cell, level)


def _index_key(k):
    return (k[2], k[1], k[0])  # indexing::Index ordering: z, then y, then x


def flatten(v2v, e2v, v2e, e2e):
    """the values in the iteration order of the reference's std::map types, as a binding passes them to the C-ABI; the keys
    (destination kind, source kind, dx, dy, dz); the sizes of the four maps"""
    vals, keys = [], []
    for k in sorted(v2v, key=_index_key):
        vals.append(v2v[k]), keys.append((0, 0) + k)
    for o in ORIENTATIONS:
        for k in sorted(e2v[o], key=_index_key):
            vals.append(e2v[o][k]), keys.append((0, o + 1) + k)
    n_v2v, n_e2v = len(v2v), len(vals) - len(v2v)
    for c in ORIENTATIONS:
        for k in sorted(v2e[c], key=_index_key):
            vals.append(v2e[c][k]), keys.append((c + 1, 0) + k)
    n_v2e = len(vals) - n_v2v - n_e2v
    for c in ORIENTATIONS:
        for l in ORIENTATIONS:
            for k in sorted(e2e[c][l], key=_index_key):
                vals.append(e2e[c][l][k]), keys.append((c + 1, l + 1) + k)
    return vals, keys, [n_v2v, n_e2v, n_v2e, len(vals) - n_v2v - n_e2v - n_v2e]


def apply_cell_inner(dst_v, dst_e, src_v, src_e, level, stencils, update=0, parts=("v2v", "e2v", "v2e", "e2e")):
    """P2ConstantOperator::apply on the INNER DoFs of one macro-cell (the macro-cell kernels' iteration spaces): vertex DoFs:
    vertexToVertex (updateType) then edgeToVertex (Add); edge DoFs: edgeToEdge (updateType) then vertexToEdge (Add).  With a
    subset of `parts` the first part present takes updateType (a single sub-operator called on its own)"""
    v2v, e2v, v2e, e2e = stencils
    N = (1 << level) + 1
    n = N - 1
    vi = lambda p: po.cell_index(level, *p)  # noqa: E731
    ei = lambda p, o: po.edge_index(level, p[0], p[1], p[2], o)  # noqa: E731
    for z in range(1, N):
        for y in range(1, N - z):
            for x in range(1, N - z - y):
                if x + y + z >= N - 1:
                    continue
                p, i = (x, y, z), vi((x, y, z))
                first = True
                if "v2v" in parts:
                    acc = sum(w * src_v[vi(_add(p, k))] for k, w in v2v.items())
                    dst_v[i] = acc if update == 0 else dst_v[i] + acc
                    first = False
                if "e2v" in parts:
                    acc = sum(w * src_e[ei(_add(p, k), o)] for o in ORIENTATIONS for k, w in e2v[o].items())
                    dst_v[i] = acc if (update == 0 and first) else dst_v[i] + acc
    for c in ORIENTATIONS:
        W = n - 1 if c == XYZ else n
        for z in range(W):
            for y in range(W - z):
                for x in range(W - z - y):
                    p = (x, y, z)
                    if not is_inner_edge(level, p, c):
                        continue
                    i = ei(p, c)
                    first = True
                    if "e2e" in parts:
                        acc = sum(w * src_e[ei(_add(p, k), l)] for l in ORIENTATIONS for k, w in e2e[c][l].items())
                        dst_e[i] = acc if update == 0 else dst_e[i] + acc
                        first = False
                    if "v2e" in parts:
                        acc = sum(w * src_v[vi(_add(p, k))] for k, w in v2e[c].items())
                        dst_e[i] = acc if (update == 0 and first) else dst_e[i] + acc
    return dst_v, dst_e
