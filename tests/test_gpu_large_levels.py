"""The upper end of the supported level range (HYTEG_HIP_MAX_LEVEL = 11): the kernels switch index arithmetic and code
paths with the size -- 32-bit slice offsets up to level 10, 64-bit above; buffer addressing with a 32-bit byte range up to
level 8..10 arrays (23 MB .. 1.4 GB), the LDS-tiled apply with 64-bit pointers for the 11.5 GB arrays of level 11.
Level 9 is compared with the CPU oracle entry by entry for every kernel of the path; levels 10 and 11 check the apply against
a stencil evaluation at random points done with torch on the device (the arrays never leave the GPU) and against the
reference's own property that the Laplace stencil annihilates linear functions (P1LaplaceOperator3DTest.cpp:109-126)."""
import numpy as np
import pytest

from conftest import OCT_TET, SKEW_TET

pytestmark = pytest.mark.gpu

OFFS = [(0, 0, -1), (1, 0, -1), (-1, 1, -1), (0, 1, -1), (0, -1, 0), (1, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (-1, 1, 0),
        (0, 1, 0), (0, -1, 1), (1, -1, 1), (-1, 0, 1), (0, 0, 1)]  # order of the 15 weights (include/hyteg_hip.h)


@pytest.fixture(scope="module")
def env():
    import torch

    from hyteg_amd import capi
    from oracle import p1_oracle as po

    assert torch.cuda.is_available()
    capi.lib()
    return torch, capi, po


def _rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def test_level9_every_kernel_against_the_oracle(env):
    torch, capi, po = env
    level = 9
    n, nc = po.cell_size(level), po.cell_size(level - 1)
    assert n == 22632705
    rng = np.random.default_rng(9)
    w = po.assemble_cell_stencil(SKEW_TET, level)
    a_h, b_h = rng.random(n), rng.random(n)
    a, b = torch.from_numpy(a_h).cuda(), torch.from_numpy(b_h).cuda()
    m = po.inner_mask(level)
    # apply Replace / Add
    d = torch.from_numpy(b_h).cuda()
    capi.p1_apply_cell(d.data_ptr(), a.data_ptr(), level, w, capi.ADD)
    ref = b_h.copy()
    po.apply_cell(ref, a_h, level, w, capi.ADD)
    torch.cuda.synchronize()
    got = d.cpu().numpy()
    assert np.array_equal(got[~m], b_h[~m])
    assert _rel(got[m], ref[m]) < 1e-13
    # fused Jacobi
    d = torch.zeros(n, dtype=torch.float64, device="cuda")
    capi.p1_jacobi_cell(d.data_ptr(), b.data_ptr(), a.data_ptr(), level, w, 0.6, None)
    ref = np.zeros(n)
    po.jacobi_cell(ref, b_h, a_h, level, w, 0.6)
    torch.cuda.synchronize()
    assert _rel(d.cpu().numpy(), ref) < 1e-13
    # assign, dot
    capi.p1_assign_cell(d.data_ptr(), [2.0, -0.5], [a.data_ptr(), b.data_ptr()], level)
    torch.cuda.synchronize()
    assert _rel(d.cpu().numpy()[m], (2.0 * a_h - 0.5 * b_h)[m]) < 1e-15
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.zeros(capi.dot_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
    capi.p1_dot_cell(a.data_ptr(), b.data_ptr(), level, res.data_ptr(), ws.data_ptr())
    torch.cuda.synchronize()
    assert abs(float(res[0]) - float(np.dot(a_h[m], b_h[m]))) < 1e-11 * float(np.dot(a_h[m], b_h[m]))
    # grid transfer 9 <-> 8
    nnc = [1.0] * 14
    coarse = torch.zeros(nc, dtype=torch.float64, device="cuda")
    capi.p1_restrict_cell(coarse.data_ptr(), a.data_ptr(), level - 1, nnc)
    ref = np.zeros(nc)
    po.restrict_cell(ref, a_h, level - 1, nnc)
    torch.cuda.synchronize()
    assert _rel(coarse.cpu().numpy(), ref) < 1e-13
    c_h = rng.random(nc)
    fine = torch.zeros(n, dtype=torch.float64, device="cuda")
    capi.p1_prolongate_cell(torch.from_numpy(c_h).cuda().data_ptr(), fine.data_ptr(), level - 1, nnc, capi.REPLACE)
    ref = np.zeros(n)
    po.prolongate_prepare(ref, level, capi.REPLACE)
    po.prolongate_cell(c_h, ref, level - 1, nnc)
    torch.cuda.synchronize()
    assert _rel(fine.cpu().numpy(), ref) < 1e-13
    # Gauss-Seidel, exact order (blocked form: 32 x 32 x 32 blocks of 16^3)
    u = torch.from_numpy(a_h).cuda()
    capi.p1_sor_cell(u.data_ptr(), b.data_ptr(), level, w, 1.0, False)
    ref = a_h.copy()
    po.sor_cell(ref, b_h, level, w, 1.0, False)
    torch.cuda.synchronize()
    assert _rel(u.cpu().numpy(), ref) < 1e-12


def _sampled_apply_check(torch, capi, po, level, nsamples=3000):
    """apply on device arrays that never leave the GPU; the stencil is re-evaluated with torch at random interior points"""
    n = capi.cell_size(level)
    N = (1 << level) + 1
    w = po.assemble_cell_stencil(OCT_TET, level)
    g = torch.Generator(device="cuda")
    g.manual_seed(level)
    src = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    dst = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
    capi.p1_apply_cell(dst.data_ptr(), src.data_ptr(), level, w, capi.REPLACE)
    torch.cuda.synchronize()
    rng = np.random.default_rng(level)
    pts = []
    while len(pts) < nsamples:
        x, y, z = (int(v) for v in rng.integers(1, N - 2, 3))
        if x + y + z <= N - 2:
            pts.append((x, y, z))
    # corners of the interior too: first / last inner point of the array, longest row, tip
    pts += [(1, 1, 1), (N - 4, 1, 1), (1, N - 4, 1), (1, 1, N - 4), (N - 5, 2, 1)]
    centre = torch.tensor([capi.cell_index(level, *p) for p in pts], dtype=torch.int64, device="cuda")
    want = torch.zeros(len(pts), dtype=torch.float64, device="cuda")
    for k, (dx, dy, dz) in enumerate(OFFS):
        idx = torch.tensor([capi.cell_index(level, x + dx, y + dy, z + dz) for x, y, z in pts], dtype=torch.int64, device="cuda")
        want += w[k] * src[idx]
    got = dst[centre]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-13 * scale
    # boundary entries are not written: first and last array entries, a face point
    for p in ((0, 0, 0), (N - 1, 0, 0), (0, 0, N - 1), (3, 0, 5), (0, 4, 4)):
        assert float(dst[capi.cell_index(level, *p)]) == 7.0
    # the stencil annihilates linear functions: rows are contiguous in x, so a linear field is built row by row on the device
    del dst
    return n, N, w, src


@pytest.mark.parametrize("level", [10, 11])
def test_apply_at_the_largest_levels(env, level):
    torch, capi, po = env
    free, _ = torch.cuda.mem_get_info()
    n = capi.cell_size(level)
    if free < 3.2 * n * 8:
        pytest.skip(f"needs {3.2 * n * 8 / 2**30:.0f} GiB of device memory")
    n, N, w, src = _sampled_apply_check(torch, capi, po, level)
    # Laplace of a linear function vanishes at every inner point (checked through the largest magnitude): fill src with
    # 3x - 2y + 5z + 1 using the array layout (index of (x,y,z) = slice start + row start + x)
    k = 0
    src.zero_()
    z_starts = [capi.cell_index(level, 0, 0, z) for z in range(N)]
    for z in range(N):
        W = N - z
        # rows of this slice: lengths W, W-1, ..., 1; coordinates from the slice-local offset
        j = torch.arange(W * (W + 1) // 2, dtype=torch.int64, device="cuda")
        # row y of offset j: largest y with y*W - y(y-1)/2 <= j
        y = torch.floor(((2 * W + 1) - torch.sqrt(((2 * W + 1) ** 2 - 8 * j).double())) / 2).long()
        y = torch.where(y * W - y * (y - 1) // 2 > j, y - 1, y)
        y = torch.where((y + 1) * W - (y + 1) * y // 2 <= j, y + 1, y)
        x = j - (y * W - y * (y - 1) // 2)
        src[z_starts[z]:z_starts[z] + j.numel()] = 3.0 * x.double() - 2.0 * y.double() + 5.0 * z + 1.0
    dst = torch.zeros(n, dtype=torch.float64, device="cuda")
    capi.p1_apply_cell(dst.data_ptr(), src.data_ptr(), level, w, capi.REPLACE)
    torch.cuda.synchronize()
    # |w| ~ 2^level-independent O(1) sums of values up to ~1e4: rounding level 1e-16 * 15 * 1e4
    assert float(dst.abs().max()) < 1e-9
    assert float(src[capi.cell_index(level, 2, 3, 4)]) == 3 * 2 - 2 * 3 + 5 * 4 + 1


def _inner_mask_on_device(torch, capi, level):
    """bool array: True at inner points (x, y, z >= 1, x + y + z <= N - 2), built slice by slice from the layout"""
    N = (1 << level) + 1
    n = capi.cell_size(level)
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    for z in range(1, N - 2):
        W = N - z
        j = torch.arange(W * (W + 1) // 2, dtype=torch.int64, device="cuda")
        y = torch.floor(((2 * W + 1) - torch.sqrt(((2 * W + 1) ** 2 - 8 * j).double())) / 2).long()
        y = torch.where(y * W - y * (y - 1) // 2 > j, y - 1, y)
        y = torch.where((y + 1) * W - (y + 1) * y // 2 <= j, y + 1, y)
        x = j - (y * W - y * (y - 1) // 2)
        s0 = capi.cell_index(level, 0, 0, z)
        mask[s0:s0 + j.numel()] = (x >= 1) & (y >= 1) & (x + y + z <= N - 2)
    return mask


def test_vector_kernels_jacobi_and_transfer_at_level_11(env):
    """11.5 GB per array: every index of these kernels has to be 64-bit clean.  Checked on the device against torch."""
    torch, capi, po = env
    level = 11
    n = capi.cell_size(level)
    free, _ = torch.cuda.mem_get_info()
    if free < 6.5 * n * 8:
        pytest.skip(f"needs {6.5 * n * 8 / 2**30:.0f} GiB of device memory")
    N = (1 << level) + 1
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    a = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    b = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    mask = _inner_mask_on_device(torch, capi, level)
    assert int(mask.sum()) == capi.cell_inner_size(level)
    # assign: inner points get 2a - 0.5b, everything else keeps its value
    d = torch.full((n,), -3.0, dtype=torch.float64, device="cuda")
    capi.p1_assign_cell(d.data_ptr(), [2.0, -0.5], [a.data_ptr(), b.data_ptr()], level)
    torch.cuda.synchronize()
    want = torch.where(mask, 2.0 * a - 0.5 * b, torch.full_like(a, -3.0))
    assert bool(torch.equal(d, want))
    del want
    # dot over the inner points
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.zeros(capi.dot_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
    capi.p1_dot_cell(a.data_ptr(), b.data_ptr(), level, res.data_ptr(), ws.data_ptr())
    torch.cuda.synchronize()
    ref = float((a * b * mask).sum())
    assert abs(float(res[0]) - ref) <= 1e-11 * ref
    # fused Jacobi at random points (and untouched boundary)
    w = po.assemble_cell_stencil(OCT_TET, level)
    d.fill_(-3.0)
    capi.p1_jacobi_cell(d.data_ptr(), b.data_ptr(), a.data_ptr(), level, w, 0.6, None)
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    pts = []
    while len(pts) < 2000:
        x, y, z = (int(v) for v in rng.integers(1, N - 2, 3))
        if x + y + z <= N - 2:
            pts.append((x, y, z))
    pts += [(1, 1, 1), (N - 4, 1, 1), (1, N - 4, 1), (1, 1, N - 4)]
    centre = torch.tensor([capi.cell_index(level, *p) for p in pts], dtype=torch.int64, device="cuda")
    au = torch.zeros(len(pts), dtype=torch.float64, device="cuda")
    for k, (dx, dy, dz) in enumerate(OFFS):
        idx = torch.tensor([capi.cell_index(level, x + dx, y + dy, z + dz) for x, y, z in pts], dtype=torch.int64, device="cuda")
        au += w[k] * a[idx]
    want = a[centre] + 0.6 * (1.0 / w[7]) * (b[centre] - au)
    assert float((d[centre] - want).abs().max()) <= 1e-12 * float(want.abs().max())
    assert float(d[capi.cell_index(level, 0, 5, 5)]) == -3.0 and float(d[n - 1]) == -3.0
    del d, mask
    # grid transfer 11 <-> 10 on constants: restriction of 1 with all neighbour counts 1 gives the column sums of the
    # prolongation (8 at inner coarse points: 1 + 14/2), prolongation of 1 gives 1 at every fine point
    nc = capi.cell_size(level - 1)
    a.fill_(1.0)
    coarse = torch.zeros(nc, dtype=torch.float64, device="cuda")
    capi.p1_restrict_cell(coarse.data_ptr(), a.data_ptr(), level - 1, [1.0] * 14)
    torch.cuda.synchronize()
    assert float(coarse[capi.cell_index(level - 1, 3, 4, 5)]) == 8.0
    assert float(coarse[capi.cell_index(level - 1, 1, 1, (1 << (level - 1)) - 3)]) == 8.0  # the last inner point along z
    assert float(coarse.max()) == 8.0
    coarse.fill_(1.0)
    b.fill_(-1.0)
    capi.p1_prolongate_cell(coarse.data_ptr(), b.data_ptr(), level - 1, [1.0] * 14, capi.REPLACE)
    torch.cuda.synchronize()
    assert float(b.min()) == 1.0 and float(b.max()) == 1.0


def test_gauss_seidel_level_10_against_the_oracle(env):
    """180 M points, 64^3 / 6 blocks of 16^3 in 190 block wavefronts: block tables, row-base table and staging at 1.4 GB"""
    torch, capi, po = env
    level = 10
    n = po.cell_size(level)
    rng = np.random.default_rng(10)
    w = po.assemble_cell_stencil(SKEW_TET, level)
    u_h, b_h = rng.random(n), rng.random(n)
    u, b = torch.from_numpy(u_h).cuda(), torch.from_numpy(b_h).cuda()
    capi.p1_sor_cell(u.data_ptr(), b.data_ptr(), level, w, 1.0, False)
    capi.p1_sor_cell(u.data_ptr(), b.data_ptr(), level, w, 1.2, True)
    po.sor_cell(u_h, b_h, level, w, 1.0, False)
    po.sor_cell(u_h, b_h, level, w, 1.2, True)
    torch.cuda.synchronize()
    assert _rel(u.cpu().numpy(), u_h) < 1e-12


def test_gauss_seidel_level_11_leaves_a_linear_function_alone(env):
    """A u = 0 for linear u, so with rhs = 0 every update of a sweep reproduces the value it replaces: a fixed point of the
    smoother whatever the order -- and any wrong index at 11.5 GB shows as a jump of O(1000)"""
    torch, capi, po = env
    level = 11
    n = capi.cell_size(level)
    free, _ = torch.cuda.mem_get_info()
    if free < 3.5 * n * 8:
        pytest.skip(f"needs {3.5 * n * 8 / 2**30:.0f} GiB of device memory")
    N = (1 << level) + 1
    w = po.assemble_cell_stencil(OCT_TET, level)
    u = torch.zeros(n, dtype=torch.float64, device="cuda")
    for z in range(N):
        W = N - z
        j = torch.arange(W * (W + 1) // 2, dtype=torch.int64, device="cuda")
        y = torch.floor(((2 * W + 1) - torch.sqrt(((2 * W + 1) ** 2 - 8 * j).double())) / 2).long()
        y = torch.where(y * W - y * (y - 1) // 2 > j, y - 1, y)
        y = torch.where((y + 1) * W - (y + 1) * y // 2 <= j, y + 1, y)
        x = j - (y * W - y * (y - 1) // 2)
        s0 = capi.cell_index(level, 0, 0, z)
        u[s0:s0 + j.numel()] = 3.0 * x.double() - 2.0 * y.double() + 5.0 * z + 1.0
    u0 = u.clone()
    rhs = torch.zeros(n, dtype=torch.float64, device="cuda")
    capi.p1_sor_cell(u.data_ptr(), rhs.data_ptr(), level, w, 1.0, False)
    capi.p1_sor_cell(u.data_ptr(), rhs.data_ptr(), level, w, 1.0, True)
    torch.cuda.synchronize()
    assert float((u - u0).abs().max()) < 1e-8  # values up to 1e4, 2 x 15-term sums per point


def test_grid_transfer_level_9_to_10_on_constants_and_a_linear(env):
    """fine level 10 (1.44 GB per array) is the largest level of the brick prolongation and of 32-bit buffer offsets: prolongation
    of 1 is 1 everywhere; prolongation of a linear function in the index coordinates reproduces it (P1 interpolation is exact
    for linears: VertexDoFLinearProlongation3DTest.cpp:107-137), checked at every fine point on the device; restriction of 1
    with all neighbour counts 1 gives 8 at inner coarse points."""
    torch, capi, po = env
    lc, lf = 9, 10
    nc, nf = capi.cell_size(lc), capi.cell_size(lf)
    free, _ = torch.cuda.mem_get_info()
    if free < 4.5 * nf * 8:
        pytest.skip(f"needs {4.5 * nf * 8 / 2**30:.0f} GiB of device memory")

    def linear(level):
        # value 3x - 2y + 5z + 1 in units of the level's mesh width: built slice by slice, row by row on the device
        N = (1 << level) + 1
        h = 1.0 / (N - 1)
        parts = []
        for z in range(N):
            W = N - z
            ys = torch.repeat_interleave(torch.arange(W, device="cuda"), torch.arange(W, 0, -1, device="cuda"))
            starts = torch.cumsum(torch.arange(W, 0, -1, device="cuda"), 0) - torch.arange(W, 0, -1, device="cuda")
            xs = torch.arange(ys.numel(), device="cuda") - starts[ys]
            parts.append((3.0 * xs - 2.0 * ys + 5.0 * z).to(torch.float64) * h + 1.0)
        return torch.cat(parts)

    ones = [1.0] * 14
    coarse = torch.ones(nc, dtype=torch.float64, device="cuda")
    fine = torch.full((nf,), -1.0, dtype=torch.float64, device="cuda")
    capi.p1_prolongate_cell(coarse.data_ptr(), fine.data_ptr(), lc, ones, capi.REPLACE)
    torch.cuda.synchronize()
    assert float(fine.min()) == 1.0 and float(fine.max()) == 1.0
    coarse = linear(lc)
    assert coarse.numel() == nc
    capi.p1_prolongate_cell(coarse.data_ptr(), fine.data_ptr(), lc, ones, capi.REPLACE)
    torch.cuda.synchronize()
    want = linear(lf)
    assert float((fine - want).abs().max()) <= 1e-13 * float(want.abs().max())
    del want
    fine.fill_(1.0)
    capi.p1_restrict_cell(coarse.data_ptr(), fine.data_ptr(), lc, ones)
    torch.cuda.synchronize()
    assert float(coarse[capi.cell_index(lc, 3, 4, 5)]) == 8.0 and float(coarse.max()) == 8.0
