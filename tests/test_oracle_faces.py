"""CPU pins of the HyTeG-layout macro-face functions of the oracle (ghost copies in all 24 orientations and the
face apply), after tests/hyteg/vertexdofspace/VertexDoFMacroCellPackInfoTest.cpp:88-108 (after the sync every cell
entry equals the interpolated analytic function) and tests/hyteg/P1/P1LaplaceOperator3DTest.cpp (A u = 0 for linear u
on macro-face DoFs of multi-cell meshes)."""
import itertools

import numpy as np
import pytest

from conftest import SKEW_TET
from hostutil import MESHES, cell_points
from hyteg_amd import host
from oracle import p1_oracle as po

ORIENTATIONS = list(itertools.permutations(range(4), 3))  # 24 ordered triples of cell-local vertex ids


def _field(p):
    return np.sin(3 * p[..., 0]) + 2.0 * p[..., 1] * p[..., 2] - p[..., 0] * p[..., 1]


def _face_points(cc, v, level, layer=0):
    """physical coordinates of the face-local points (x, y, layer) in face array order"""
    n = 1 << level
    cc = np.asarray(cc, dtype=float)
    v3 = 6 - sum(v)
    w = po.width(level) - layer
    pts = []
    for y in range(w):
        for x in range(w - y):
            pts.append(cc[v[0]] + (cc[v[1]] - cc[v[0]]) * x / n + (cc[v[2]] - cc[v[0]]) * y / n + (cc[v3] - cc[v[0]]) * layer / n)
    return np.array(pts)


@pytest.mark.parametrize("v", ORIENTATIONS)
def test_ghost_copies_in_all_24_orientations(v):
    level = 3
    cc = np.array(SKEW_TET)
    P = cell_points(cc, level)
    analytic = _field(P)
    # face -> cell
    face = np.zeros(po.face_array_size(level, 2))
    nf = po.face_size_w(po.width(level))
    face[:nf] = _field(_face_points(cc, v, level))
    cell = np.full(po.cell_size(level), -77.0)
    po.copy_face_to_cell(cell, face, level, v)
    local_face = {(0, 1, 2): 0, (0, 1, 3): 1, (0, 2, 3): 2, (1, 2, 3): 3}[tuple(sorted(v))]
    on_face = np.array([po.prim_slot(level, *map(int, c)) >= 0 and _on_face(level, c, local_face) for c in po.cell_coords(level)])
    assert np.allclose(cell[on_face], analytic[on_face], atol=1e-13)
    assert np.all(cell[~on_face] == -77.0)
    # cell -> face ghost layer (both ghost slots)
    for nb in (0, 1):
        f2 = np.full(po.face_array_size(level, 2), -5.0)
        po.copy_cell_to_face(f2, analytic, level, v, nb)
        ng = po.face_size_w(po.width(level) - 1)
        ghost = f2[nf + nb * ng: nf + (nb + 1) * ng]
        assert np.allclose(ghost, _field(_face_points(cc, v, level, layer=1)), atol=1e-13)
        assert np.all(f2[:nf] == -5.0) and np.all(f2[nf + (1 - nb) * ng: nf + (2 - nb) * ng] == -5.0)


def _on_face(level, c, f):
    n = 1 << level
    x, y, z = map(int, c)
    return (z == 0, y == 0, x == 0, x + y + z == n)[f]


def _face_setup(st, level, face_verts):
    """for the macro-face with the given global vertex ids: per neighbour cell (ascending id) its local index, the
    vertex map v (cell-local ids of the face's vertices, in the order given) and its face slot"""
    out = []
    for i in range(st.n_local_cells):
        gid, co, nnc = st.local_cell(i)
        # global vertex ids of the cell: recover from coordinates
        ids = [int(np.argmin(np.linalg.norm(ALLV - c, axis=1))) for c in co]
        if all(g in ids for g in face_verts):
            v = [ids.index(g) for g in face_verts]
            out.append((i, co, v, 6 + {(0, 1, 2): 0, (0, 1, 3): 1, (0, 2, 3): 2, (1, 2, 3): 3}[tuple(sorted(v))]))
    return out


ALLV = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 1], [1, 1, 0], [0.5, 0.5, 0], [0.5, 0.5, -1.0]])  # octahedron nodes


@pytest.mark.parametrize("face_verts", [(0, 3, 5), (5, 0, 3), (1, 5, 6), (2, 4, 5)])
@pytest.mark.parametrize("level", [2, 3])
def test_face_apply_equals_the_sum_of_the_cells_shares_and_annihilates_linears(face_verts, level):
    st = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    cells = _face_setup(st, level, face_verts)
    assert len(cells) == 2
    n = 1 << level
    nf, ng = po.face_size_w(po.width(level)), po.face_size_w(po.width(level) - 1)
    for fn in (_field, lambda p: 42 * p[..., 0] + p[..., 1] + 1337 * p[..., 2]):
        face = np.zeros(po.face_array_size(level, 2))
        vmaps, ws, shares = [], [], []
        for k, (i, co, v, slot) in enumerate(cells):
            arr = np.ascontiguousarray(fn(cell_points(co, level)))
            if k == 0:
                face[:nf] = fn(_face_points(co, v, level))
            po.copy_cell_to_face(face, arr, level, v, k)
            slots = po.assemble_cell_slot_stencils(co, level)
            vmaps.append(v)
            ws.append(slots[slot])
            part = np.zeros_like(arr)
            po.apply_cell_boundary(part, arr, level, slots, 1 << slot)
            shares.append((co, v, part))
        out = np.zeros(po.face_array_size(level, 2))
        po.apply_face3d(out, face, level, vmaps, ws)
        # expected: sum of the two cells' shares at the same physical point
        k = 0
        for y in range(n + 1):
            for x in range(n + 1 - y):
                if x >= 1 and y >= 1 and x + y <= n - 1:
                    tot = 0.0
                    for co, v, part in shares:
                        bary = [0, 0, 0, 0]
                        bary[v[0]], bary[v[1]], bary[v[2]] = n - x - y, x, y
                        tot += part[po.cell_index(level, bary[1], bary[2], bary[3])]
                    assert abs(out[k] - tot) < 1e-12 * max(1.0, abs(tot))
                    if fn is not _field:
                        assert abs(out[k]) < 2.8e-13 * 1337  # Laplace annihilates linears on the macro-face DoFs
                else:
                    assert out[k] == 0.0
                k += 1
    st.close()


@pytest.mark.parametrize("ncells", [1, 2])
@pytest.mark.parametrize("backwards", [False, True])
def test_sor_face3d_is_gauss_seidel_for_the_face_apply(ncells, backwards):
    """ho_sor_face3d restates P1Operator::smooth_sor_face3D (P1Operator.hpp:1424-1503).  Pins: (i) one update leaves the
    face equation of the LAST updated DoF satisfied exactly (relax = 1): (apply_face3d(u) - rhs) vanishes there; (ii) swept to
    convergence, apply_face3d(u) = rhs at every inner face DoF while ghost layers and face boundary keep their values;
    (iii) relax = 0 changes nothing."""
    level = 3
    n = 1 << level
    rng = np.random.default_rng(5 + ncells)
    vmaps = [(0, 1, 2), (2, 0, 3)][:ncells]
    tets = [SKEW_TET, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]]
    ws = []
    for k, v in enumerate(vmaps):
        slot = 6 + {(0, 1, 2): 0, (0, 1, 3): 1, (0, 2, 3): 2, (1, 2, 3): 3}[tuple(sorted(v))]
        ws.append(po.assemble_cell_slot_stencils(tets[k], level)[slot])
    size = po.face_array_size(level, 2)
    u0, rhs = rng.random(size), rng.random(size)
    nf = po.face_size_w(po.width(level))
    inner = np.zeros(size, dtype=bool)
    k, order = 0, []
    for y in range(n + 1):
        for x in range(n + 1 - y):
            if x >= 1 and y >= 1 and x + y <= n - 1:
                inner[k] = True
                order.append(k)
            k += 1
    u = u0.copy()
    po.sor_face3d(u, rhs, level, vmaps, ws, 0.0, backwards)
    assert np.array_equal(u, u0)
    po.sor_face3d(u, rhs, level, vmaps, ws, 1.0, backwards)
    assert np.array_equal(u[~inner], u0[~inner])
    res = np.zeros(size)
    po.apply_face3d(res, u, level, vmaps, ws)
    last = order[0] if backwards else order[-1]
    assert abs(res[last] - rhs[last]) < 1e-13 * abs(rhs[last])
    for _ in range(400):
        po.sor_face3d(u, rhs, level, vmaps, ws, 1.0, backwards)
    po.apply_face3d(res, u, level, vmaps, ws)
    assert np.abs(res[inner] - rhs[inner]).max() < 1e-11
    assert np.array_equal(u[~inner], u0[~inner])
