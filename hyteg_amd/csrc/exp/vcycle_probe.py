import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[3]))
import numpy as np, torch
from hyteg_amd import capi, host
mesh = sys.argv[1] if len(sys.argv) > 1 else "regular_octahedron_8el"
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 2
smoother = host.GAUSS_SEIDEL if (len(sys.argv) > 4 and sys.argv[4] == "gs") else host.JACOBI
ROOT = pathlib.Path(__file__).resolve().parents[3]
st = host.Storage.from_gmsh(ROOT / f"hyteg_amd/data/meshes/{mesh}.msh")
st.set_stream(torch.cuda.current_stream().cuda_stream)
A = host.P1ConstantOperator(st, lo, hi); A.compute_inverse_diagonal()
x, b = host.P1Function(st, "x", lo, hi), host.P1Function(st, "b", lo, hi)
rng = np.random.default_rng(0)
for c in range(st.n_local_cells): x.upload_cell(c, hi, rng.random(capi.cell_size(hi)))
x.sync_shared(hi, host.All); x.interpolate(0.0, hi, host.DirichletBoundary)
gmg = host.Solver.gmg(st, lo, hi, smoother=smoother, relax=2/3, pre=3, post=3, cg_max_iter=50, cg_tol=1e-10)
gmg.solve(A, x, b, hi); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): gmg.solve(A, x, b, hi)
torch.cuda.synchronize()
print(f"{mesh} L{lo}-{hi}: {(time.perf_counter()-t0)/3*1e3:.2f} ms per V(3,3) cycle")
