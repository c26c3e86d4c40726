"""P2 apply, one macro-cell, levels 4-8: the kernels of round 2 (rows + thread-per-DoF boundary) against the row kernel with every point class"""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from hyteg_amd import capi
if os.environ.get("HYTEG_PROBE_LIB"):  # a library built with other compile-time parameters (e.g. -DHYTEG_P2_CLASS_ROWS_WAVES=1)
    from pathlib import Path
    capi._LIB_PATH = Path(os.environ["HYTEG_PROBE_LIB"])
from oracle import p1_oracle as po

REF = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=np.float64)
for level in [int(x) for x in os.environ.get("HYTEG_PROBE_LEVELS", "4,5,6,7,8").split(",")]:
    nv, ne = capi.cell_size(level), capi.p2_edge_array_size(level)
    em = po.p2_cell_element_matrices(REF, min(level, 6))
    dem = torch.from_numpy(capi.p2_build_operator_table(em)).cuda()
    nb = max(3, int(2.2 * 2**28 / ((nv + ne) * 8)) + 1)
    S = [(torch.rand(nv, dtype=torch.float64, device="cuda"), torch.rand(ne, dtype=torch.float64, device="cuda")) for _ in range(nb)]
    D = [(torch.zeros(nv, dtype=torch.float64, device="cuda"), torch.zeros(ne, dtype=torch.float64, device="cuda")) for _ in range(nb)]
    st = torch.cuda.current_stream().cuda_stream
    def run(mask, n=200):
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(n):
                sv, se = S[k % nb]; dv, de = D[k % nb]
                capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), sv.data_ptr(), se.data_ptr(), level, dem.data_ptr(), 1.0, 0, mask)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        return best
    for first, name in ((99, "rows (round 2)"), (3, "class rows")):
        capi.p2_set_class_rows_min_level(first)
        run(0x7FFF, 20)
        print(f"level {level} {name:16s} all {run(0x7FFF):8.2f} us   inner only {run(0x4000):8.2f} us   ({nb} buffer pairs)", flush=True)
    del S, D
