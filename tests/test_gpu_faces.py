"""GPU parity of the HyTeG-layout macro-face kernels (ghost copies face<->cell, one-/two-sided face apply) against
the CPU oracle, through the C-ABI."""
import itertools

import numpy as np
import pytest

from conftest import OCT_TET, SKEW_TET

pytestmark = pytest.mark.gpu
ORIENTATIONS = list(itertools.permutations(range(4), 3))


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    return torch, capi, po


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


@pytest.mark.parametrize("level", [1, 2, 4, 6])
def test_ghost_copies_all_orientations(env, level):
    torch, capi, po = env
    rng = np.random.default_rng(level)
    nface = po.face_array_size(level, 2)
    face_h, cell_h = rng.random(nface), rng.random(po.cell_size(level))
    face, cell = _dev(torch, face_h), _dev(torch, cell_h)
    for v in ORIENTATIONS:
        c = _dev(torch, cell_h)
        capi.p1_copy_face_to_cell(c.data_ptr(), face.data_ptr(), level, v)
        torch.cuda.synchronize()
        ref = cell_h.copy()
        po.copy_face_to_cell(ref, face_h, level, v)
        assert np.array_equal(c.cpu().numpy(), ref)
        for nb in (0, 1):
            f = _dev(torch, face_h)
            capi.p1_copy_cell_to_face(f.data_ptr(), cell.data_ptr(), level, v, nb)
            torch.cuda.synchronize()
            ref = face_h.copy()
            po.copy_cell_to_face(ref, cell_h, level, v, nb)
            assert np.array_equal(f.cpu().numpy(), ref)


@pytest.mark.parametrize("level", [2, 3, 5])
@pytest.mark.parametrize("ncells", [1, 2])
def test_apply_face3d(env, level, ncells):
    torch, capi, po = env
    rng = np.random.default_rng(7 * level + ncells)
    tets = [SKEW_TET, OCT_TET]
    nface = po.face_array_size(level, 2)
    src_h, dst0 = rng.random(nface), rng.random(nface)
    src = _dev(torch, src_h)
    for vs in ([(0, 1, 2), (2, 0, 3)], [(3, 1, 0), (1, 2, 3)], [(1, 3, 2), (0, 2, 1)]):
        vmaps = vs[:ncells]
        ws = []
        for k, v in enumerate(vmaps):
            slot = 6 + {(0, 1, 2): 0, (0, 1, 3): 1, (0, 2, 3): 2, (1, 2, 3): 3}[tuple(sorted(v))]
            ws.append(po.assemble_cell_slot_stencils(tets[k], level)[slot])
        for update in (0, 1):
            dst = _dev(torch, dst0)
            capi.p1_apply_face3d(dst.data_ptr(), src.data_ptr(), level, vmaps, ws, update)
            torch.cuda.synchronize()
            ref = dst0.copy()
            po.apply_face3d(ref, src_h, level, vmaps, ws, update)
            got = dst.cpu().numpy()
            assert np.linalg.norm(got - ref) <= 1e-13 * np.linalg.norm(ref)
            # only the inner face DoFs are written
            nf = po.face_size_w(po.width(level))
            assert np.array_equal(got[nf:], dst0[nf:])


@pytest.mark.parametrize("level", [2, 3, 5, 7])
@pytest.mark.parametrize("ncells", [1, 2])
def test_sor_face3d(env, level, ncells):
    """sor_3D_macroface_P1{,_one_sided}{,_backwards} in HyTeG's face layout against the literal restatement of
    P1Operator::smooth_sor_face3D (P1Operator.hpp:1424-1503): the updates in the same order, the sums reassociated."""
    torch, capi, po = env
    rng = np.random.default_rng(11 * level + ncells)
    tets = [SKEW_TET, OCT_TET]
    nface = po.face_array_size(level, 2)
    u0, rhs_h = rng.random(nface), rng.random(nface)
    rhs = _dev(torch, rhs_h)
    work = torch.zeros(capi.p1_sor_face3d_workspace(level) // 8, dtype=torch.float64, device="cuda")
    for vs in ([(0, 1, 2), (2, 0, 3)], [(3, 1, 0), (1, 2, 3)], [(1, 3, 2), (0, 2, 1)]):
        vmaps = vs[:ncells]
        ws = []
        for k, v in enumerate(vmaps):
            slot = 6 + {(0, 1, 2): 0, (0, 1, 3): 1, (0, 2, 3): 2, (1, 2, 3): 3}[tuple(sorted(v))]
            ws.append(po.assemble_cell_slot_stencils(tets[k], level)[slot])
        for relax, backwards in ((1.0, False), (1.0, True), (0.7, False), (1.3, True)):
            u = _dev(torch, u0)
            capi.p1_sor_face3d(u.data_ptr(), rhs.data_ptr(), work.data_ptr(), level, vmaps, ws, relax, backwards)
            torch.cuda.synchronize()
            ref = u0.copy()
            po.sor_face3d(ref, rhs_h, level, vmaps, ws, relax, backwards)
            got = u.cpu().numpy()
            nf = po.face_size_w(po.width(level))
            assert np.array_equal(got[nf:], u0[nf:])  # ghost layers untouched
            assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
