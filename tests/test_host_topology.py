"""CPU-only tests of the C++ host layer's topology and additive-exchange plans (no GPU work is launched)."""
import numpy as np
import pytest

from hostutil import MESHES, cell_points
from hyteg_amd import host
from oracle import p1_oracle as po


@pytest.mark.parametrize("mesh,counts", [("tet_1el", (1, 4, 6, 4)), ("regular_octahedron_8el", (8, 20, 18, 7)),
                                         ("pyramid_2el", (2, 7, 9, 5)), ("pyramid_4el", (4, 12, 13, 6)),
                                         ("cube_6el", (6, 18, 19, 8)), ("cube_24el", (24, 60, 49, 14))])
def test_primitive_counts(mesh, counts):
    """Euler characteristic V - E + F - C = 1 for these ball-like meshes, and the counts of the mesh files"""
    s = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    got = (s.n_cells, s.n_faces, s.n_edges, s.n_vertices)
    assert got == counts
    assert s.n_vertices - s.n_edges + s.n_faces - s.n_cells == 1
    s.close()


def test_single_tet_has_no_shared_points_and_dirichlet_masks():
    s = host.Storage.from_gmsh(MESHES / "tet_1el.msh")
    for level in (2, 4):
        for cls in (0, 1):
            assert s.plan(level, cls)["ngroups"] == 0
    assert s.mask(0, host.Inner) == 1 << 14           # whole shell is Dirichlet boundary
    assert s.mask(0, host.All) == 0x7FFF
    assert s.mask(0, host.DirichletBoundary) == 0x3FFF
    gid, coords, nnc = s.local_cell(0)
    assert np.all(nnc == 1)
    s.close()


@pytest.mark.parametrize("mesh", ["regular_octahedron_8el", "pyramid_tilted_4el", "cube_24el"])
@pytest.mark.parametrize("level", [1, 2, 3])
def test_every_copy_of_a_shared_dof_is_the_same_physical_point(mesh, level):
    """Orientation check (the cell-centric analogue of VertexDoFMacroCellPackInfoTest.cpp:88-108): all entries of a
    group must refer to one micro-vertex, and every shared micro-vertex must appear in exactly one group."""
    s = host.Storage.from_gmsh(MESHES / f"{mesh}.msh")
    pts = [cell_points(s.local_cell(i)[1], level) for i in range(s.n_local_cells)]
    seen = set()
    for cls in (0, 1):
        p = s.plan(level, cls)
        gp, eb, eo = p["group_ptr"], p["entry_buf"], p["entry_off"]
        for g in range(p["ngroups"]):
            ents = [(eb[e], eo[e]) for e in range(gp[g], gp[g + 1])]
            assert len(ents) >= 2
            ref = pts[ents[0][0]][ents[0][1]]
            for c, o in ents[1:]:
                assert np.allclose(pts[c][o], ref, atol=1e-13)
            assert sorted(c for c, _ in ents) == [c for c, _ in ents]  # ascending cell order = fixed summation order
            for c, o in ents:
                assert po.slot_of_points(level)[o] != 14
                assert (c, o) not in seen
                seen.add((c, o))
    # completeness: every boundary entry of every cell whose point also lies in another cell is in some group
    all_pts = {}
    for c, P in enumerate(pts):
        for o in np.nonzero(po.slot_of_points(level) != 14)[0]:
            all_pts.setdefault(tuple(np.round(P[o], 9)), []).append((c, o))
    shared = {e for ents in all_pts.values() if len(ents) > 1 for e in ents}
    assert shared == seen
    s.close()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_send_and_receive_orders_agree_between_ranks(nranks):
    """For every ordered pair (a -> b): the points a packs for b, in order, are the points b expects from a, in order
    (each rank builds its plan independently from the global mesh)."""
    level = 2
    mesh = MESHES / "regular_octahedron_8el.msh"
    sts = [host.Storage.from_gmsh(mesh, r, nranks) for r in range(nranks)]
    assert sum(s.n_local_cells for s in sts) == 8
    for cls in (0, 1):
        plans = [s.plan(level, cls) for s in sts]
        pts = [[cell_points(s.local_cell(i)[1], level) for i in range(s.n_local_cells)] for s in sts]
        for a in range(nranks):
            pa = plans[a]
            off = 0
            for k, b in enumerate(pa["peers"]):
                cnt = int(pa["send_count"][k])
                sent = [pts[a][pa["send_buf"][off + j]][pa["send_off"][off + j]] for j in range(cnt)]
                off += cnt
                pb = plans[b]
                assert a in list(pb["peers"]), "peer relation must be symmetric"
                kb = list(pb["peers"]).index(a)
                assert int(pb["recv_count"][kb]) == cnt
                # entries of b that read from a's segment, ordered by their receive offset
                nloc = sts[b].n_local_cells
                exp = {}
                gp, eb, eo = pb["group_ptr"], pb["entry_buf"], pb["entry_off"]
                for g in range(pb["ngroups"]):
                    loc = [(eb[e], eo[e]) for e in range(gp[g], gp[g + 1]) if eb[e] < nloc]
                    for e in range(gp[g], gp[g + 1]):
                        if eb[e] == nloc + kb:
                            exp[int(eo[e])] = pts[b][loc[0][0]][loc[0][1]]  # the physical point of this group
                assert sorted(exp) == list(range(cnt))
                for j in range(cnt):
                    assert np.allclose(sent[j], exp[j], atol=1e-13)
    for s in sts:
        s.close()


def test_owned_masks_count_every_dof_once():
    level = 2
    s = host.Storage.from_gmsh(MESHES / "regular_octahedron_8el.msh")
    pts = {}
    for i in range(s.n_local_cells):
        P = cell_points(s.local_cell(i)[1], level)
        sel = (s.mask(i, host.All, owned=True) >> po.slot_of_points(level)) & 1
        for o in np.nonzero(sel)[0]:
            key = tuple(np.round(P[o], 9))
            assert key not in pts
            pts[key] = 1
    # number of distinct micro-vertices of the refined octahedron = sum over primitives of their interior points
    n = 1 << level
    expect = 7 + 18 * (n - 1) + 20 * (n - 1) * (n - 2) // 2 + 8 * (n - 1) * (n - 2) * (n - 3) // 6
    assert len(pts) == expect
    s.close()
