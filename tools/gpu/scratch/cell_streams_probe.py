"""P2 apply over the 6 / 24 macro-cells one GPU holds (cube_6el, cube_24el: BASELINE config 4's per-GPU share), level 7, through the host
layer: the cell launches on one stream or alternating between two (HYTEG_AMD_CELL_STREAMS=2)"""
import sys, pathlib, time, os
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import host
host.lib()
for mesh in ("cube_6el", "cube_24el"):
    st = host.Storage.from_gmsh(ROOT / f"hyteg_amd/data/meshes/{mesh}.msh")
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    A = host.P2ElementwiseLaplaceOperator(st, 7, 7)
    u, r = host.P2Function(st, "u", 7, 7), host.P2Function(st, "r", 7, 7)
    u.interpolate(1.0, 7)
    for _ in range(3):
        A.apply(u, r, 7, host.Inner)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            A.apply(u, r, 7, host.Inner)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e6)
    print(f"HYTEG_AMD_CELL_STREAMS={os.environ.get('HYTEG_AMD_CELL_STREAMS', '1')} {mesh}: {best:8.1f} us per apply ({st.n_local_cells} cells, {best / st.n_local_cells:6.1f} us per cell)", flush=True)
    for o in (u, r, A, st):
        o.close()
