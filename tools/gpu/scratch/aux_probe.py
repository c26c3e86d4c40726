"""apply Replace level 8, shape 4x8x2, in the HBM regime (source arrays 2.2 x the Infinity Cache): one process per cache policy"""
import sys, pathlib, os
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi
L = 8
capi.lib(); capi.prepare_level(L)
n = capi.cell_size(L)
nb = 26
st = torch.cuda.current_stream().cuda_stream
w = [0.1 * (k + 1) for k in range(15)]; w[7] = -3.0
A = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nb)]
B = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nb)]
E0, E1 = capi.event_create_timing(), capi.event_create_timing()
capi.set_apply_shape(4, 8, 2)
def go(K):
    for k in range(K):
        capi.p1_apply_cell(B[k % nb].data_ptr(), A[k % nb].data_ptr(), L, w, 0, st)
go(2 * nb); torch.cuda.synchronize()
best = []
for r in range(7):
    capi.event_record(E0, st); go(500); capi.event_record(E1, st)
    best.append(capi.event_elapsed_ms(E0, E1) * 1e3 / 500)
best.sort()
print(f"aux {os.environ.get('HYTEG_HIP_APPLY_AUX', 'default (2,0)'):14s} min {best[0]:7.3f} median {best[3]:7.3f} us", flush=True)
