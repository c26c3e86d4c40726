"""one P1-P1 Stokes V(3,3)+2 cycle on cube_24el, levels 2-5 (the reference's P1P1Stokes3DUzawaConvergenceTest configuration); under
rocprofv3 --kernel-trace --stats: where its time goes"""
import sys, pathlib, time
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import host
host.lib()
st = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/cube_24el.msh")
st.set_stream(torch.cuda.current_stream().cuda_stream)
L = host.P1P1StokesOperator(st, 2, 5)
u, f = host.P1StokesFunction(st, "u", 2, 5), host.P1StokesFunction(st, "f", 2, 5)
for k in range(4):
    u.components[k].interpolate(0.0, 5, host.All); f.components[k].interpolate(1.0 if k < 3 else 0.0, 5, host.Inner)
uz = host.StokesSolver.uzawa(st, 2, 5, 0.3, velocity_iterations=2, velocity_smoother=host.GAUSS_SEIDEL)
gmg = host.StokesSolver.gmg(st, uz, 2, 5, pre=3, post=3, increment=2, project_mean_after_restriction=True)
gmg.solve(L, u, f, 5); torch.cuda.synchronize()
t0 = time.perf_counter()
gmg.solve(L, u, f, 5); torch.cuda.synchronize()
print(f"P1-P1 Stokes V(3,3)+2 cycle, cube_24el levels 2-5: {(time.perf_counter() - t0) * 1e3:.1f} ms wall", flush=True)
