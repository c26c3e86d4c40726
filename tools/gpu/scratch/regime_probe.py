"""apply Replace at level 8 (and 9) in both regimes for whatever kernel the environment selects; also two independent applies
issued alternately on two streams (what a cell loop over several macro-cells could do)"""
import sys, pathlib, os
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi
capi.lib()
w = [0.1 * (k + 1) for k in range(15)]; w[7] = -3.0
E0, E1 = capi.event_create_timing(), capi.event_create_timing()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("HYTEG_HIP_")) or "default"
for L, rings in ((8, (6, 26)), (9, (4,))):
    capi.prepare_level(L)
    n = capi.cell_size(L)
    for nb in rings:
        A = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nb)]
        B = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(nb)]
        st = torch.cuda.current_stream().cuda_stream
        K = 400 if L == 8 else 60
        def go(K, streams=(st,)):
            for k in range(K):
                capi.p1_apply_cell(B[k % nb].data_ptr(), A[k % nb].data_ptr(), L, w, 0, streams[k % len(streams)])
        go(2 * nb); torch.cuda.synchronize()
        t = []
        for r in range(5):
            capi.event_record(E0, st); go(K); capi.event_record(E1, st)
            t.append(capi.event_elapsed_ms(E0, E1) * 1e3 / K)
        line = f"[{tag}] level {L} ring {nb:2d} pairs: {sorted(t)[2]:8.3f} us per apply"
        if L == 8 and nb == 26:
            s2 = torch.cuda.Stream()
            t2 = []
            for r in range(5):
                torch.cuda.synchronize()
                ta = torch.cuda.Event(enable_timing=True); tb = torch.cuda.Event(enable_timing=True)
                ta.record()
                s2.wait_stream(torch.cuda.current_stream())
                go(K, (st, s2.cuda_stream))
                torch.cuda.current_stream().wait_stream(s2)
                tb.record(); torch.cuda.synchronize()
                t2.append(ta.elapsed_time(tb) * 1e3 / K)
            line += f"   alternating on two streams: {sorted(t2)[2]:8.3f} us per apply"
        print(line, flush=True)
        del A, B
        torch.cuda.empty_cache()
