"""P2 apply, level 7: what the boundary DoFs cost on the critical path of the fused launch -- mask ALL vs INNER only vs SHELL only;
and the two grid transfers for reference."""
import sys, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi
from oracle import p1_oracle as po  # element matrices of a tetrahedron (input data only)

L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
capi.lib(); capi.prepare_level(L)
nv, ne = capi.cell_size(L), capi.p2_edge_array_size(L)
co = np.array([0, 0, 0, 1, 0, 0, 0.2, 1, 0, 0.1, 0.3, 1.0])
em = po.p2_cell_element_matrices(co, L)
table = torch.tensor(capi.p2_build_operator_table(em), device="cuda")
st = torch.cuda.current_stream().cuda_stream
nb = 12
sv = [torch.rand(nv, dtype=torch.float64, device="cuda") for _ in range(nb)]
se = [torch.rand(ne, dtype=torch.float64, device="cuda") for _ in range(nb)]
dv = [torch.zeros(nv, dtype=torch.float64, device="cuda") for _ in range(nb)]
de = [torch.zeros(ne, dtype=torch.float64, device="cuda") for _ in range(nb)]
E0, E1 = capi.event_create_timing(), capi.event_create_timing()

def run(name, mask, kinds=0xFF, K=200):
    def go(n):
        for k in range(n):
            j = k % nb
            capi.p2_elementwise_apply_cell(dv[j].data_ptr(), de[j].data_ptr(), sv[j].data_ptr(), se[j].data_ptr(), L, table.data_ptr(), 1.0, 0, mask, st, kinds)
    go(2 * nb); torch.cuda.synchronize()
    best = 1e9
    for r in range(5):
        capi.event_record(E0, st); go(K); capi.event_record(E1, st)
        best = min(best, capi.event_elapsed_ms(E0, E1) * 1e3 / K)
    print(f"level {L}  {name:44s} {best:8.2f} us", flush=True)

run("all DoFs (fused launch)", 0x7FFF)
run("inner DoFs only (rows)", 1 << 14)
run("boundary DoFs only (thread per DoF)", 0x3FFF)
run("faces only", 0xF << 6)
run("edges + vertices only", 0x3F | (0xF << 10))
run("all DoFs, vertex kind only", 0x7FFF, 1)
run("all DoFs, edge kinds only", 0x7FFF, 0xFE)
