import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[3]

def worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HYTEG_HIP_P2P_TIMEOUT_MS="300")
    import faulthandler
    faulthandler.dump_traceback_later(50, exit=True)
    import torch, torch.distributed as dist
    from hyteg_amd import host
    from hyteg_amd.distributed import DistributedContext
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    level = 3
    st = host.Storage.from_gmsh(ROOT / "hyteg_amd" / "data" / "meshes" / "pyramid_2el.msh", rank, world)
    st.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx = DistributedContext(st, [level], torch.device("cuda", 0), transport="p2p")
    A = host.P1ConstantOperator(st, level, level)
    u, r = host.P1Function(st, "u", level, level), host.P1Function(st, "r", level, level)
    u.interpolate(1.0, level, host.All)
    A.apply(u, r, level, host.Inner); r.dot(r, level, host.Inner)
    dist.barrier()
    if rank == 1:
        time.sleep(2.0)
    t0 = time.perf_counter()
    A.apply(u, r, level, host.Inner)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    msg = f"rank {rank}: apply host {t1 - t0:.3f} s, device drained after {t2 - t0:.3f} s, transport {st.transport}, plans {st.plan(level, 0)['peers']}"
    try:
        st.check_transport()
        msg += " | check_transport: no error"
    except host.HytegHostError as e:
        msg += " | check_transport: " + str(e)[:120]
    q.put(msg)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs: p.start()
    for _ in range(2):
        try: print(q.get(timeout=90), flush=True)
        except Exception as e: print("no result", repr(e), flush=True)
    for p in procs:
        p.join(timeout=10)
        if p.is_alive(): p.kill()
