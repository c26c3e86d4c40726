"""Developer probe (round 2): cost of ONE exchange through the peer-to-peer C-ABI (hyteg_hip_p2p_pack + hyteg_hip_p2p_wait,
comm_p2p.hip) next to the plain pack kernel (hyteg_hip_gather_entries) it replaces.  One GPU, one process: the three
"peers" are this rank's own arena (loop-back, no link latency) -- the figure to hold against rccl_native_probe.py's
27 / 46 us for the ncclSend / ncclRecv group, which comes ON TOP of the plain pack kernel.  Run: python p2p_probe.py"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hyteg_amd import capi  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L = capi.lib()
seg, npeers = 32385, 3  # three level-8 macro-faces: what a rank of the 8-cell mesh sends per apply
n = seg * npeers
cur = torch.cuda.current_stream().cuda_stream


class Peer(C.Structure):
    _fields_ = [("slot0", C.c_void_p), ("slot1", C.c_void_p), ("flag", C.c_void_p), ("start", C.c_int), ("count", C.c_int)]


for kind in ("uncached", "finegrained", "default"):
    import os
    os.environ["HYTEG_HIP_P2P_ARENA"] = kind
    base, handle, k = C.c_void_p(), C.create_string_buffer(64), C.c_int()
    arena_bytes = 2 * n * 8 + 4096
    capi.check(L.hyteg_hip_p2p_arena_create(arena_bytes, C.byref(base), handle, C.byref(k)), "arena_create")
    flags_off = 2 * n * 8
    peers = (Peer * npeers)()
    for p in range(npeers):
        peers[p] = Peer(base.value + p * seg * 8, base.value + (n + p * seg) * 8, base.value + flags_off + 64 * p, p * seg, seg)
    d_peers = torch.frombuffer(bytearray(bytes(peers)), dtype=torch.uint8).to(dev)
    src = torch.rand(n, dtype=torch.float64, device=dev)
    bases = torch.tensor([src.data_ptr()], dtype=torch.int64, device=dev)
    ebuf = torch.zeros(n, dtype=torch.int32, device=dev)
    eoff = torch.arange(n, dtype=torch.int32, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    plain = torch.zeros(n, dtype=torch.float64, device=dev)
    seq = [0]

    def gather():
        capi.check(L.hyteg_hip_gather_entries(plain.data_ptr(), bases.data_ptr(), ebuf.data_ptr(), eoff.data_ptr(), n, cur), "gather")

    def p2p():
        seq[0] += 1
        capi.check(L.hyteg_hip_p2p_pack(d_peers.data_ptr(), npeers, bases.data_ptr(), ebuf.data_ptr(), eoff.data_ptr(), n, seq[0],
                                        counter.data_ptr(), cur), "p2p_pack")
        capi.check(L.hyteg_hip_p2p_wait(base.value + flags_off, npeers, 8, seq[0], status.data_ptr(), 2000, cur), "p2p_wait")

    reps = 300
    for name, fn in ((f"plain pack kernel (gather {n} doubles)", gather), (f"p2p pack + wait, {kind} arena (loop-back)", p2p)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        host_us = (time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        total_us = (time.perf_counter() - t0) / reps * 1e6
        print(f"{name:56s} host {host_us:7.2f} us per exchange, until drained {total_us:7.2f} us")
    # the values arrived in the slot of the last sequence parity, no wait timed out
    got = np.empty(n)
    capi.check(L.hyteg_hip_download(got.ctypes.data_as(C.c_void_p), C.c_void_p(base.value + (seq[0] & 1) * n * 8), n * 8, cur), "download")
    torch.cuda.synchronize()
    assert np.array_equal(got, src.cpu().numpy()) and int(status.item()) == 0
    capi.check(L.hyteg_hip_p2p_arena_destroy(base), "arena_destroy")
