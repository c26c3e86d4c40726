"""p1_sor_cell at the levels whose cell array fits into LDS (one workgroup, all hyperplanes), single cell and a batch of 8"""
import sys, pathlib
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi
capi.lib()
st = torch.cuda.current_stream().cuda_stream
w = [0.1 * (k + 1) for k in range(15)]; w[7] = -3.0
E0, E1 = capi.event_create_timing(), capi.event_create_timing()
for L in (2, 3, 4, 5, 6):
    capi.prepare_level(L)
    n = capi.cell_size(L)
    u = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(8)]
    b = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(8)]
    def one(K):
        for k in range(K):
            capi.p1_sor_cell(u[k % 8].data_ptr(), b[k % 8].data_ptr(), L, w, 1.0, False, st)
    one(8); torch.cuda.synchronize()
    best = 1e9
    for r in range(5):
        capi.event_record(E0, st); one(100); capi.event_record(E1, st)
        best = min(best, capi.event_elapsed_ms(E0, E1) * 1e3 / 100)
    print(f"level {L}: sor_cell {best:8.2f} us", flush=True)
