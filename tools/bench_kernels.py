#!/usr/bin/env python3
"""Per-kernel timings of the whole P1 hot path on one GPU (level 8 by default) with algorithmic GB/s, plus V-cycle
timings through the host layer.  Not the headline bench (that is bench.py); this is the evidence table for
DESIGN.md section 3.  Usage: python tools/bench_kernels.py [--level 8] [--reps 200]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from hyteg_amd import capi, host  # noqa: E402


def cpu_rows(L, w, rows):
    """CPU baseline beside the GPU rows: the oracle's restatement of the reference's generated kernels, built with the flags of a
    HyTeG Release build (oracle/Makefile: -O3 -march=native), one thread as one MPI rank runs them; protocol of
    apps/benchmarks/KernelBench/3DKernelBench.cpp:59-82 (double the sweeps until a block lasts 0.5 s), bounded to ~2 s per row."""
    from oracle import p1_oracle as po

    n, nc, inner = po.cell_size(L), po.cell_size(L - 1), po.cell_inner_size(L)
    rng = np.random.default_rng(1)
    a, b, c, co = rng.random(n), rng.random(n), rng.random(n), rng.random(nc)
    ones = np.ones(14)
    fast = po.lib(fast=True)

    def protocol(fn, budget=2.0):
        fn()
        sweeps, total = 1, 0.0
        while True:
            t0 = time.perf_counter()
            for _ in range(sweeps):
                fn()
            dt = time.perf_counter() - t0
            total += dt
            if dt > 0.5 or total > budget:
                return dt / sweeps
            sweeps *= 2

    table = [
        ("apply Replace", lambda: po.apply_cell(b, a, L, w, fast=True), 16 * inner, inner),
        ("SOR forward sweep", lambda: po.sor_cell(b, a, L, w, 1.0, False, fast=True), 24 * inner, inner),
        ("Jacobi composition (apply + 3 vector passes)", lambda: po.jacobi_cell(b, c, a, L, w, 0.66, None, fast=True), 104 * inner, inner),
        ("restrict (fine L -> coarse L-1)", lambda: fast.ho_restrict_cell(po._p(co), po._p(a), L - 1, po._p(ones)), 8 * (n + nc), nc),
        ("prolongate (coarse L-1 -> fine L, incl. zeroing)",
         lambda: (fast.ho_prolongate_prepare(po._p(b), L, 0), fast.ho_prolongate_cell(po._p(co), po._p(b), L - 1, po._p(ones))), 8 * (n + nc), n),
    ]
    for name, fn, nbytes, updates in table:
        s_ = protocol(fn)
        us = s_ * 1e6
        rows.append(dict(kernel="CPU 1 core: " + name, us=us, GBps=nbytes / us * 1e-3, GDoFps=updates / us * 1e-3, algorithmic_bytes=nbytes))
        print(f"{'CPU 1 core: ' + name:58s} {us:12.1f} us  {nbytes / us * 1e-3:8.2f} GB/s  {updates / us * 1e-3:8.3f} GDoF/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=8)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default=None, help="time only the kernel rows whose name contains this text (skips the cycles)")
    ap.add_argument("--cpu", action="store_true",
                    help="also time the CPU restatement of the reference kernels (oracle/, gcc -O3 -march=native, 1 thread) for apply, "
                         "SOR, the Jacobi composition, restriction and prolongation (BASELINE.md section 2) -- a reported baseline")
    args = ap.parse_args()
    L, reps = args.level, args.reps
    n, inner = capi.cell_size(L), capi.cell_inner_size(L)
    nc = capi.cell_size(L - 1)
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream
    # every array class (sources, destinations, third operands) alone exceeds the 256 MiB Infinity Cache 2.2 times: a ring that only
    # exceeds it in total is READ from that cache when the destination is written with nontemporal stores (round 3)
    nbuf = max(2, -(-int(2.2 * 256 * 2**20) // (n * 8)))
    A = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    B = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    Cc = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    Co = [torch.rand(nc, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh")
    st.set_stream(sh)
    op = host.P1ConstantOperator(st, 2, L)
    w = list(op.stencils(0, L)[0])
    ones = [1.0] * 14
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = torch.zeros(capi.dot_workspace_bytes() // 8, dtype=torch.float64, device="cuda")
    rows = []

    def timeit(name, fn, bytes_per_call, updates, r=reps):
        if args.only and args.only not in name:
            return
        for k in range(5):
            fn(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(r):
            fn(k)
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / r
        rows.append(dict(kernel=name, us=us, GBps=bytes_per_call / us * 1e-3, GDoFps=updates / us * 1e-3,
                         algorithmic_bytes=bytes_per_call))
        print(f"{name:44s} {us:10.2f} us  {bytes_per_call / us * 1e-3:8.1f} GB/s  {updates / us * 1e-3:8.1f} GDoF/s", flush=True)

    if args.cpu:
        cpu_rows(L, w, rows)
    p = lambda t, k: t[k % nbuf].data_ptr()  # noqa: E731
    timeit("apply Replace", lambda k: capi.p1_apply_cell(p(B, k), p(A, k), L, w, 0, sh), 16 * inner, inner)
    timeit("apply Add", lambda k: capi.p1_apply_cell(p(B, k), p(A, k), L, w, 1, sh), 24 * inner, inner)
    timeit("Jacobi fused (scalar inverse diagonal)", lambda k: capi.p1_jacobi_cell(p(B, k), p(Cc, k), p(A, k), L, w, 0.66, None, sh),
           24 * inner, inner)
    timeit("Jacobi fused (inverse-diagonal function)",
           lambda k: capi.p1_jacobi_cell(p(B, k), p(Cc, k), p(A, k), L, w, 0.66, p(Cc, k + 1), sh), 32 * inner, inner)
    timeit("assign 1 source", lambda k: capi.p1_assign_cell(p(B, k), [2.0], [p(A, k)], L, sh), 16 * inner, inner)
    timeit("assign 2 sources", lambda k: capi.p1_assign_cell(p(B, k), [2.0, -1.0], [p(A, k), p(Cc, k)], L, sh), 24 * inner, inner)
    timeit("add 1 source", lambda k: capi.p1_add_cell(p(B, k), [2.0], [p(A, k)], L, sh), 24 * inner, inner)
    timeit("multElementwise 2 sources", lambda k: capi.p1_mult_cell(p(B, k), [p(A, k), p(Cc, k)], L, sh), 24 * inner, inner)
    timeit("dot", lambda k: capi.p1_dot_cell(p(A, k), p(B, k), L, res.data_ptr(), ws.data_ptr(), sh), 16 * inner, inner)
    timeit("restrict (fine L -> coarse L-1)", lambda k: capi.p1_restrict_cell(p(Co, k), p(A, k), L - 1, ones, sh), 8 * (n + nc), nc)
    timeit("prolongate Replace (coarse L-1 -> fine L)", lambda k: capi.p1_prolongate_cell(p(Co, k), p(B, k), L - 1, ones, 0, sh),
           8 * (n + nc), n)
    # float32 instantiations: algorithmic bytes halve
    Af = [t.to(torch.float32) for t in A] + [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(nbuf)]  # 2 nbuf: half the bytes
    Bf = [t.to(torch.float32) for t in B] + [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    Cf = [t.to(torch.float32) for t in Cc] + [torch.rand(n, dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    pf = lambda t, k: t[k % len(t)].data_ptr()  # noqa: E731
    timeit("apply Replace, float32", lambda k: capi.p1_apply_cell_f32(pf(Bf, k), pf(Af, k), L, w, 0, sh), 8 * inner, inner)
    timeit("Jacobi fused, float32 (scalar inverse diagonal)",
           lambda k: capi.p1_jacobi_cell_f32(pf(Bf, k), pf(Cf, k), pf(Af, k), L, w, 0.66, None, sh), 12 * inner, inner)
    for name, alg in (("dataflow, one launch", capi.SOR_DATAFLOW), ("16^3 blocks, one launch per block wavefront", capi.SOR_BLOCKS)):
        capi.set_sor_algorithm(alg)
        timeit(f"SOR forward sweep ({name})", lambda k: capi.p1_sor_cell(p(B, k), p(A, k), L, w, 1.0, False, sh), 24 * inner, inner,
               r=max(3, reps // 40))
        timeit(f"SOR backward sweep ({name})", lambda k: capi.p1_sor_cell(p(B, k), p(A, k), L, w, 1.0, True, sh), 24 * inner, inner,
               r=max(3, reps // 40))
    capi.set_sor_algorithm(capi.SOR_AUTO)
    # the sweeps of a smoothing phase as one pipeline of block wavefronts (bytes / DoFs of ALL the sweeps of a call)
    for ns in (2, 3, 4):
        timeit(f"SOR {ns} forward sweeps, pipelined (per call)", lambda k: capi.p1_sor_cell_sweeps(p(B, k), p(A, k), L, w, 1.0, ns, False, sh),
               ns * 24 * inner, ns * inner, r=max(3, reps // 40))
    # Gauss-Seidel on the cell's macro-vertices/-edges/-faces (what a multi-cell sweep adds per cell); octahedron weights
    sys.path.insert(0, str(ROOT / "tests"))
    import hostutil as hu  # noqa: E402

    mv, mc = hu.read_msh(ROOT / "hyteg_amd/data/meshes/regular_octahedron_8el.msh")
    T = hu.sor_tables(mv, mc, L)[0]
    shell_pts = 4 * ((1 << L) + 1) * ((1 << L) + 2) // 2
    timeit("SOR shell forward (4 faces, 6 edges, 4 vertices)",
           lambda k: capi.p1_sor_shell_cell(p(B, k), p(A, k), p(Cc, k), L, T["edge_verts"], T["edge_w"], T["face_verts"], T["face_w"],
                                            T["vertex_w"], 1.0, 0x3FFF, False, sh), 40 * shell_pts, shell_pts, r=max(3, reps // 10))
    timeit("SOR shell backward", lambda k: capi.p1_sor_shell_cell(p(B, k), p(A, k), p(Cc, k), L, T["edge_verts"], T["edge_w"], T["face_verts"],
                                                                  T["face_w"], T["vertex_w"], 1.0, 0x3FFF, True, sh),
           40 * shell_pts, shell_pts, r=max(3, reps // 10))
    # P2 elementwise Laplace apply on one macro-cell, level 7 as in BASELINE config 4
    L2 = min(L, 7)
    nv2, ne2 = capi.cell_size(L2), capi.p2_edge_array_size(L2)
    p2op = host.P2ElementwiseLaplaceOperator(st, L2, L2)  # element matrices (kernel input) from the host layer's P2LaplaceForm
    em = torch.from_numpy(capi.p2_build_operator_table(p2op.element_matrices(L2))).to("cuda")
    nb2 = max(2, -(-int(2.2 * 256 * 2**20) // ((nv2 + ne2) * 8)))  # the sources alone 2.2 x the Infinity Cache
    SV = [torch.rand(nv2, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    SE = [torch.rand(ne2, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    DV = [torch.zeros(nv2, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    DE = [torch.zeros(ne2, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    timeit(f"P2 elementwise Laplace apply, level {L2} ({nv2 + ne2} DoFs)",
           lambda k: capi.p2_elementwise_apply_cell(DV[k % nb2].data_ptr(), DE[k % nb2].data_ptr(), SV[k % nb2].data_ptr(),
                                                    SE[k % nb2].data_ptr(), L2, em.data_ptr(), 1.0, 0, 0x7FFF, sh),
           16 * (nv2 + ne2), nv2 + ne2, r=max(3, reps // 10))
    # P2 quadratic grid transfer between level L2 - 1 and L2
    nvc, nec = capi.cell_size(L2 - 1), capi.p2_edge_array_size(L2 - 1)
    CV = [torch.rand(nvc, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    CE = [torch.rand(nec, dtype=torch.float64, device="cuda") for _ in range(nb2)]
    timeit(f"P2 prolongate level {L2 - 1} -> {L2}",
           lambda k: capi.p2_prolongate_cell(DV[k % nb2].data_ptr(), DE[k % nb2].data_ptr(), CV[k % nb2].data_ptr(), CE[k % nb2].data_ptr(),
                                             L2 - 1, 0, 0x7FFF, sh), 8 * (nv2 + ne2 + nvc + nec), nv2 + ne2, r=max(3, reps // 10))
    timeit(f"P2 restrict level {L2} -> {L2 - 1}",
           lambda k: capi.p2_restrict_cell(CV[k % nb2].data_ptr(), CE[k % nb2].data_ptr(), SV[k % nb2].data_ptr(), SE[k % nb2].data_ptr(),
                                           L2 - 1, ones, 0x7FFF, sh), 8 * (nv2 + ne2 + nvc + nec), nvc + nec, r=max(3, reps // 10))
    if args.only:
        print(json.dumps({"level": L, "kernels": rows}))
        return
    # BASELINE config 4 shape on one GPU: P2 Laplace, level 7, the 6 macro-cells one GPU holds (cube_6el = the unit cube)
    if L >= 7:
        s6 = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/cube_6el.msh")
        s6.set_stream(sh)
        A6 = host.P2ElementwiseLaplaceOperator(s6, 7, 7)
        u6, r6 = host.P2Function(s6, "u", 7, 7), host.P2Function(s6, "r", 7, 7)
        u6.interpolate(1.0, 7)
        for _ in range(3):
            A6.apply(u6, r6, 7, host.Inner)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n6 = 10
        for _ in range(n6):
            A6.apply(u6, r6, 7, host.Inner)
        torch.cuda.synchronize()
        us6 = (time.perf_counter() - t0) / n6 * 1e6
        dofs6 = 6 * (nv2 + ne2)
        rows.append(dict(kernel="P2 elementwise apply, cube_6el, level 7, host layer (6 cells + exchange)", us=us6, GDoFps=dofs6 / us6 * 1e-3))
        print(f"P2 elementwise apply, cube_6el (6 macro-cells), level 7, host layer: {us6:9.1f} us  {dofs6 / us6 * 1e-3:7.1f} GDoF/s "
              f"({dofs6} DoFs incl. copies of shared ones)", flush=True)
        for o in (u6, r6, A6, s6):
            o.close()
    # V-cycles through the host layer
    for mesh, lo, hi, smoother, name in (("tet_1el", 2, L, host.JACOBI, "Jacobi(2/3)"), ("tet_1el", 2, L, host.JACOBI_FP32, "Jacobi fp32"),
                                          ("tet_1el", 2, min(L, 7), host.GAUSS_SEIDEL, "GS"),
                                          ("regular_octahedron_8el", 2, min(L, 6), host.JACOBI, "Jacobi(2/3)"),
                                          ("regular_octahedron_8el", 0, min(L, 6), host.GAUSS_SEIDEL, "GS")):
        s2 = host.Storage.from_gmsh(ROOT / f"hyteg_amd/data/meshes/{mesh}.msh")
        s2.set_stream(sh)
        A2 = host.P1ConstantOperator(s2, lo, hi)
        A2.compute_inverse_diagonal()
        x, b, one = (host.P1Function(s2, nm, lo, hi) for nm in ("x", "b", "one"))
        one.interpolate(1.0, hi, host.All)
        dofs = one.dot(one, hi, host.Inner)
        rng = np.random.default_rng(0)
        for c in range(s2.n_local_cells):
            x.upload_cell(c, hi, rng.random(capi.cell_size(hi)))
        x.sync_shared(hi, host.All)
        x.interpolate(0.0, hi, host.DirichletBoundary)
        gmg = host.Solver.gmg(s2, lo, hi, smoother=smoother, relax=2.0 / 3.0, pre=3, post=3, cg_max_iter=50, cg_tol=1e-10)
        for _ in range(3):
            gmg.solve(A2, x, b, hi)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ncyc = 5
        for _ in range(ncyc):
            gmg.solve(A2, x, b, hi)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / ncyc
        rows.append(dict(kernel=f"V(3,3) {name} {mesh} L{lo}-{hi}", ms=ms, inner_dofs=dofs))
        print(f"V(3,3) {name:12s} {mesh:24s} levels {lo}-{hi}: {ms:9.2f} ms/cycle, {dofs:.0f} inner DoFs "
              f"({dofs / ms * 1e-6:.2f} GDoF/s per cycle)", flush=True)
    # P1-P1 Stokes: the reference's P1P1Stokes3DUzawaConvergenceTest configuration (cube_24el, levels 2-5, V(3,3) increment 2, Uzawa)
    s3 = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/cube_24el.msh")
    s3.set_stream(sh)
    Ls = host.P1P1StokesOperator(s3, 2, 5)
    us_, fs_ = host.P1StokesFunction(s3, "u", 2, 5), host.P1StokesFunction(s3, "f", 2, 5)
    for fn_ in (us_, fs_):
        for comp in fn_.components:
            comp.interpolate(0.0, 5, host.All)
    us_.u.interpolate(1.0, 5, host.DirichletBoundary)
    uz = host.StokesSolver.uzawa(s3, 2, 5, 0.3, velocity_iterations=2, velocity_smoother=host.GAUSS_SEIDEL)
    sg = host.StokesSolver.gmg(s3, uz, 2, 5, pre=3, post=3, increment=2)
    sg.solve(Ls, us_, fs_, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sg.solve(Ls, us_, fs_, 5)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    rows.append(dict(kernel="Stokes V(3,3)+2 Uzawa/GS cube_24el L2-5", ms=ms))
    print(f"P1-P1 Stokes V(3,3) increment 2, Uzawa(0.3) over GS, cube_24el levels 2-5: {ms:9.2f} ms/cycle", flush=True)
    t0 = time.perf_counter()
    for _ in range(20):
        Ls.apply(us_, fs_, 5, host.Inner | host.NeumannBoundary)
    torch.cuda.synchronize()
    us1 = (time.perf_counter() - t0) * 1e6 / 20
    rows.append(dict(kernel="P1P1StokesOperator::apply cube_24el L5", us=us1))
    print(f"P1P1StokesOperator::apply, cube_24el level 5 (24 cells, 10 scalar applies): {us1:9.1f} us", flush=True)
    # P2-P1 Taylor-Hood (config 5's literal wording): the reference's P2P1Stokes3DUzawaConvergenceTest configuration (cube_24el, levels 2-3,
    # V(3,3), Uzawa(0.4) over Gauss-Seidel, MINRES on the coarsest level) and one level more; operator apply; one P2 Gauss-Seidel sweep
    for lo_t, hi_t in ((2, 3), (2, 4)):
        Lt = host.TaylorHoodStokesOperator(s3, lo_t, hi_t)
        ut, ft = host.TaylorHoodFunction(s3, "u", lo_t, hi_t), host.TaylorHoodFunction(s3, "f", lo_t, hi_t)
        ut.interpolate(0.0, hi_t, host.All)
        ft.interpolate(1.0, hi_t, host.Inner)
        th = host.TaylorHoodSolver.gmg(s3, lo_t, hi_t, uzawa_relax=0.4, pre=3, post=3, increment=0, coarse_max_iter=60, coarse_rel_tol=1e-16)
        th.solve(Lt, ut, ft, hi_t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            th.solve(Lt, ut, ft, hi_t)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        rows.append(dict(kernel=f"Taylor-Hood V(3,3) Uzawa/GS cube_24el L{lo_t}-{hi_t}, MINRES(60) coarse", ms=ms))
        print(f"P2-P1 Taylor-Hood V(3,3), Uzawa(0.4) over GS, cube_24el levels {lo_t}-{hi_t}, MINRES(60) on the coarsest level: {ms:9.2f} ms/cycle", flush=True)
        t0 = time.perf_counter()
        for _ in range(10):
            Lt.apply(ut, ft, hi_t, host.Inner | host.NeumannBoundary)
        torch.cuda.synchronize()
        us1 = (time.perf_counter() - t0) * 1e6 / 10
        rows.append(dict(kernel=f"P2P1TaylorHoodStokesOperator::apply cube_24el L{hi_t}", us=us1))
        print(f"P2P1TaylorHoodStokesOperator::apply, cube_24el level {hi_t} (24 cells: 3 P2 Laplace + 3 divT + 3 div): {us1:9.1f} us", flush=True)
        for o in (th, ut, ft, Lt):
            o.close()
    A2g = host.P2ConstantLaplaceOperator(s3, 4, 4)
    A2g.compute_inverse_diagonal()
    xg, bg = host.P2Function(s3, "x", 4, 4), host.P2Function(s3, "b", 4, 4)
    xg.interpolate(0.0, 4, host.All)
    bg.interpolate(1.0, 4, host.Inner)
    A2g.smooth_sor(xg, bg, 1.0, 4, host.Inner)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        A2g.smooth_sor(xg, bg, 1.0, 4, host.Inner)
    torch.cuda.synchronize()
    usg = (time.perf_counter() - t0) * 1e6 / 5
    rows.append(dict(kernel="P2 Gauss-Seidel sweep (reference order on shared primitives) cube_24el L4", us=usg))
    print(f"P2 Gauss-Seidel sweep in the reference's order, cube_24el level 4 (24 cells): {usg:9.1f} us", flush=True)
    print(json.dumps({"level": L, "device": capi.device_name(), "rows": rows}))


if __name__ == "__main__":
    main()
