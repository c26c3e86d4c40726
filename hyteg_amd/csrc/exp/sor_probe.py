import sys, pathlib; sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[3]))
import torch, numpy as np
from hyteg_amd import capi
L=int(sys.argv[1]) if len(sys.argv)>1 else 8
n=capi.cell_size(L)
u=torch.rand(n,dtype=torch.float64,device='cuda'); rhs=torch.rand(n,dtype=torch.float64,device='cuda')
w=[-1.0]*15; w[7]=20.0
for _ in range(3):
    capi.p1_sor_cell(u.data_ptr(), rhs.data_ptr(), L, w, 1.0, False, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
