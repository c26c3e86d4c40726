set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r02t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_faces.py tests/test_gpu_sor_shell.py -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hyteg_amd import capi
from oracle import p1_oracle as po
sys.path.insert(0, 'tests')
from conftest import SKEW_TET, OCT_TET
for level in (6, 7, 8):
    vmaps = [(0, 1, 2), (2, 0, 3)]
    ws = [po.assemble_cell_slot_stencils(t, level)[s] for t, s in ((SKEW_TET, 6), (OCT_TET, 8))]
    n = po.face_array_size(level, 2)
    u = torch.rand(n, dtype=torch.float64, device='cuda'); r = torch.rand(n, dtype=torch.float64, device='cuda')
    work = torch.zeros(capi.p1_sor_face3d_workspace(level) // 8, dtype=torch.float64, device='cuda')
    for bw in (False, True):
        for _ in range(3):
            capi.p1_sor_face3d(u.data_ptr(), r.data_ptr(), work.data_ptr(), level, vmaps, ws, 1.0, bw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            capi.p1_sor_face3d(u.data_ptr(), r.data_ptr(), work.data_ptr(), level, vmaps, ws, 1.0, bw)
        e1.record(); torch.cuda.synchronize()
        print(f"sor_face3d level {level} {'backward' if bw else 'forward '} two-sided: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us per sweep", flush=True)
PY
