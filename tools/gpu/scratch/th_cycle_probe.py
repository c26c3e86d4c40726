"""one Taylor-Hood V(3,3) cycle on cube_24el, levels 2-3 (the reference's P2P1Stokes3DUzawaConvergenceTest configuration): wall time;
under rocprofv3 --kernel-trace --stats the kernel counts and the GPU's busy time"""
import sys, pathlib, time
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import host
host.lib()
st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/cube_24el.msh")
st.set_stream(torch.cuda.current_stream().cuda_stream)
L = host.TaylorHoodStokesOperator(st, 2, 3)
u, f = host.TaylorHoodFunction(st, "u", 2, 3), host.TaylorHoodFunction(st, "f", 2, 3)
u.interpolate(0.0, 3, host.All); f.interpolate(1.0, 3, host.Inner)
th = host.TaylorHoodSolver.gmg(st, 2, 3, uzawa_relax=0.4, pre=3, post=3, increment=0, coarse_max_iter=60, coarse_rel_tol=1e-16)
th.solve(L, u, f, 3); torch.cuda.synchronize()
t0 = time.perf_counter()
th.solve(L, u, f, 3); torch.cuda.synchronize()
print(f"Taylor-Hood V(3,3) cycle, cube_24el levels 2-3: {(time.perf_counter() - t0) * 1e3:.1f} ms wall", flush=True)
