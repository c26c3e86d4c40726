"""driver for the PMC passes of the P2 apply (tools/gpu/r03_p2_pmc.sh): calibration copies of the level-7 edge-DoF array (known bytes),
then the class-rows kernel and the kernels of round 2 on rotating buffers (sources 2.2 x the Infinity Cache)"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from hyteg_amd import capi
from oracle import p1_oracle as po

level = 7
REF = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=np.float64)
nv, ne = capi.cell_size(level), capi.p2_edge_array_size(level)
dem = torch.from_numpy(capi.p2_build_operator_table(po.p2_cell_element_matrices(REF, 6))).cuda()
nb = int(2.2 * 2**28 / ((nv + ne) * 8)) + 1
S = [(torch.rand(nv, dtype=torch.float64, device="cuda"), torch.rand(ne, dtype=torch.float64, device="cuda")) for _ in range(nb)]
D = [(torch.zeros(nv, dtype=torch.float64, device="cuda"), torch.zeros(ne, dtype=torch.float64, device="cuda")) for _ in range(nb)]
st = torch.cuda.current_stream().cuda_stream
for k in range(40):
    capi.calib_copy(D[k % nb][1].data_ptr(), S[k % nb][1].data_ptr(), ne, True, st)
for first in (3, 99):
    capi.p2_set_class_rows_min_level(first)
    for k in range(60):
        sv, se = S[k % nb]; dv, de = D[k % nb]
        capi.p2_elementwise_apply_cell(dv.data_ptr(), de.data_ptr(), sv.data_ptr(), se.data_ptr(), level, dem.data_ptr(), 1.0, 0, 0x7FFF)
# the P2 grid transfer between levels 6 and 7 on the same buffers (coarse side: own rotating arrays)
nvc, nec = capi.cell_size(level - 1), capi.p2_edge_array_size(level - 1)
CV = [torch.rand(nvc, dtype=torch.float64, device="cuda") for _ in range(nb)]
CE = [torch.rand(nec, dtype=torch.float64, device="cuda") for _ in range(nb)]
ONES = [1.0] * 14
for k in range(40):
    capi.p2_prolongate_cell(D[k % nb][0].data_ptr(), D[k % nb][1].data_ptr(), CV[k % nb].data_ptr(), CE[k % nb].data_ptr(), level - 1, 0, 0x7FFF, st)
for k in range(40):
    capi.p2_restrict_cell(CV[k % nb].data_ptr(), CE[k % nb].data_ptr(), S[k % nb][0].data_ptr(), S[k % nb][1].data_ptr(), level - 1, ONES, 0x7FFF, st)
torch.cuda.synchronize()
print("edge entries", ne, "vertex entries", nv, "pairs", nb)
