"""Where do the ~14 us go that a K = 20 region of bench.py loses against a K = 2000 one?  Variants of the region, 25 each:
event time / K, host time of the launch loop, with untimed applies in front of the start event, via a HIP graph,
with the direct C-ABI call instead of the host layer."""
import sys, time, pathlib
import numpy as np, torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi, host

level, K, R = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 25
capi.lib(); host.lib(); capi.prepare_level(level)
n = capi.cell_size(level)
storage = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh", 0, 1)
stream = torch.cuda.current_stream(); storage.set_stream(stream.cuda_stream)
laplace = host.P1ConstantOperator(storage, level, level)
nbuf = 9
rng = np.random.default_rng(1)
srcs = [host.P1Function(storage, f"s{k}", level, level) for k in range(nbuf)]
dsts = [host.P1Function(storage, f"d{k}", level, level) for k in range(nbuf)]
for f in srcs:
    f.upload_cell(0, level, rng.random(n))
cyc = laplace.prepared_cycle(srcs, dsts, level, host.Inner, host.Replace)
inner, _ = laplace.stencils(0, level)
ptr = [(dsts[k].cell_pointer(0, level), srcs[k].cell_pointer(0, level)) for k in range(nbuf)]
w = [float(x) for x in inner]

def direct(first, count):
    for k in range(first, first + count):
        d, s = ptr[k % nbuf]
        capi.p1_apply_cell(d, s, level, w, 0, stream.cuda_stream)

def copy(first, count):
    for k in range(first, first + count):
        d, s = ptr[k % nbuf]
        capi.calib_copy(d, s, n, True, stream.cuda_stream)

def region(fn, pre=0, first=0):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pre:
        fn(first - pre, pre)
    t0 = time.perf_counter()
    e0.record(stream)
    fn(first, K)
    t1 = time.perf_counter()
    e1.record(stream)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return e0.elapsed_time(e1) * 1e3 / K, (t1 - t0) * 1e6, (t2 - t0) * 1e6 / K

def run(name, fn, **kw):
    v = [region(fn, **kw) for _ in range(R)]
    ev = sorted(x[0] for x in v); ho = sorted(x[1] for x in v); wl = sorted(x[2] for x in v)
    print(f"{name:58s} event/K med {ev[R//2]:7.3f} min {ev[0]:7.3f} | host loop med {ho[R//2]:7.1f} us | wall/K med {wl[R//2]:7.3f}", flush=True)

import os
E0, E1 = capi.event_create_timing(), capi.event_create_timing()
os.environ["HYTEG_HIP_TIMING_EVENT_FENCE"] = "1"
F0, F1 = capi.event_create_timing(), capi.event_create_timing()
del os.environ["HYTEG_HIP_TIMING_EVENT_FENCE"]
ring = capi.calib_copy_ring(ptr, n, True, stream.cuda_stream)

def region_c(fn, E0=E0, E1=E1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(0, K, E0, E1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return capi.event_elapsed_ms(E0, E1) * 1e3 / K, (t1 - t0) * 1e6, (t2 - t0) * 1e6 / K

def run_c(name, fn, **kw):
    v = [region_c(fn, **kw) for _ in range(R)]
    ev = sorted(x[0] for x in v); ho = sorted(x[1] for x in v); wl = sorted(x[2] for x in v)
    print(f"{name:58s} event/K med {ev[R//2]:7.3f} min {ev[0]:7.3f} | host loop med {ho[R//2]:7.1f} us | wall/K med {wl[R//2]:7.3f}", flush=True)

cyc(0, 2 * nbuf)
torch.cuda.synchronize()
for rep in range(2):
    run(f"host-layer cycle, K={K}", cyc)
    run(f"host-layer cycle, K={K}, first=2 (pairs not the last two)", cyc, first=2)
    run(f"host-layer cycle, K={K}, 3 untimed applies in front", cyc, pre=3, first=3)
    run(f"host-layer cycle, K={K}, 8 untimed applies in front", cyc, pre=8, first=8)
    run_c(f"host-layer cycle, K={K}, events recorded by the C loop", cyc)
    run_c(f"copy ring, K={K}, events recorded by the C loop", ring)
    run_c(f"host-layer cycle, K={K}, C loop, events WITH system fence", cyc, E0=F0, E1=F1)
    run_c(f"copy ring, K={K}, C loop, events WITH system fence", ring, E0=F0, E1=F1)
    run(f"direct C-ABI loop (python), K={K}", direct)
    run(f"copy floor, K={K}", copy)
    run(f"copy floor, K={K}, 3 untimed copies in front", copy, pre=3, first=3)

# graph of K applies
st = torch.cuda.Stream()
storage.set_stream(st.cuda_stream)
with torch.cuda.stream(st):
    cyc(0, K); torch.cuda.synchronize()
    import ctypes as C
    L = capi.lib()
    assert L.hyteg_hip_graph_begin_capture(C.c_void_p(st.cuda_stream)) == 0
    cyc(0, K)
    g = C.c_void_p()
    assert L.hyteg_hip_graph_end_capture(C.c_void_p(st.cuda_stream), C.byref(g)) == 0
    def graph(first, count):
        assert L.hyteg_hip_graph_launch(g, C.c_void_p(st.cuda_stream)) == 0
    def region_g():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st); graph(0, K); t1 = time.perf_counter(); e1.record(st)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return e0.elapsed_time(e1) * 1e3 / K, (t1 - t0) * 1e6, (t2 - t0) * 1e6 / K
    for rep in range(2):
        v = [region_g() for _ in range(R)]
        ev = sorted(x[0] for x in v); ho = sorted(x[1] for x in v); wl = sorted(x[2] for x in v)
        print(f"{'HIP graph of K applies (own stream)':58s} event/K med {ev[R//2]:7.3f} min {ev[0]:7.3f} | host loop med {ho[R//2]:7.1f} us | wall/K med {wl[R//2]:7.3f}", flush=True)
    storage.set_stream(st.cuda_stream)
    def cyc_s(first, count):
        cyc(first, count)
    def region_s():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st); cyc(0, K); t1 = time.perf_counter(); e1.record(st)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return e0.elapsed_time(e1) * 1e3 / K, (t1 - t0) * 1e6, (t2 - t0) * 1e6 / K
    v = [region_s() for _ in range(R)]
    ev = sorted(x[0] for x in v); ho = sorted(x[1] for x in v); wl = sorted(x[2] for x in v)
    print(f"{'host-layer cycle on a non-default stream':58s} event/K med {ev[R//2]:7.3f} min {ev[0]:7.3f} | host loop med {ho[R//2]:7.1f} us | wall/K med {wl[R//2]:7.3f}", flush=True)
