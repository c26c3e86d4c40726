"""Canary of the peer-to-peer exchange (hyteg_amd/csrc/comm_p2p.hip): one short-lived process per rank that maps the other
ranks' arenas through HIP IPC, stores into them with the real pack kernel, waits with the real wait kernel and checks what
arrived -- BEFORE the application touches its GPU.

Why a process of its own: if the GPUs of a node cannot reach each other's memory the way the transport assumes, the failure
mode may be a GPU memory fault, which ends the faulting process.  bench.py --gpus N starts this canary first (one child per
rank, rendezvous through small files in the temporary directory, no torch.distributed) and only tries the peer-to-peer
transport if every rank's canary exits with 0; otherwise the run stays on RCCL send/recv.

    python -m hyteg_amd.p2p_canary RANK WORLD DEVICE TAG        exit code 0: every value of every peer arrived intact
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import tempfile
import time
from pathlib import Path

SEG = 4099      # doubles per (source rank, slot); odd on purpose
ROUNDS = 6      # both slot parities, several times
TIMEOUT_S = 45.0


class _Peer(C.Structure):
    _fields_ = [("slot0", C.c_void_p), ("slot1", C.c_void_p), ("flag", C.c_void_p), ("start", C.c_int), ("count", C.c_int)]


def _rendezvous_dir(tag: str) -> Path:
    d = Path(tempfile.gettempdir()) / f"hyteg_p2p_canary_{tag}"
    d.mkdir(parents=True, exist_ok=True)
    return d


def _wait_for(path: Path, deadline: float) -> bytes:
    while time.monotonic() < deadline:
        if path.exists():
            b = path.read_bytes()
            if len(b) == 64:
                return b
        time.sleep(0.02)
    raise TimeoutError(f"no handle from {path.name}")


def payload(rank: int, rnd: int):
    import numpy as np

    return (np.arange(SEG, dtype=np.float64) * 1e-3 + 1000.0 * rank + rnd).astype(np.float64)


def run(rank: int, world: int, device: int, tag: str) -> None:
    import numpy as np

    from hyteg_amd import capi

    L = capi.lib()
    ck = capi.check
    ck(L.hyteg_hip_set_device(device), "set_device")
    deadline = time.monotonic() + TIMEOUT_S
    # arena layout, the same on every rank: for source rank r slot s at (s * world + r) * SEG doubles, flag words behind
    flags_off = 2 * world * SEG * 8
    arena_bytes = flags_off + 64 * world
    base, handle = C.c_void_p(), C.create_string_buffer(64)
    ck(L.hyteg_hip_p2p_arena_create(arena_bytes, C.byref(base), handle, None), "arena_create")
    d = _rendezvous_dir(tag)
    tmp = d / f"handle_{rank}.tmp"
    tmp.write_bytes(handle.raw)
    tmp.rename(d / f"handle_{rank}.bin")
    peers = [r for r in range(world) if r != rank]
    mapped = {}
    for r in peers:
        m = C.c_void_p()
        ck(L.hyteg_hip_p2p_arena_open(_wait_for(d / f"handle_{r}.bin", deadline), C.byref(m)), f"arena_open({r})")
        mapped[r] = m.value
    # every rank must have opened every arena before anybody may destroy one: second barrier at the end

    def dev(nbytes):
        p = C.c_void_p()
        ck(L.hyteg_hip_malloc(C.byref(p), nbytes), "malloc")
        return p

    def up(p, arr):
        a = np.ascontiguousarray(arr)
        ck(L.hyteg_hip_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes, None), "upload")
        ck(L.hyteg_hip_stream_synchronize(None), "sync")

    n = SEG * len(peers)
    src = dev(SEG * 8)
    bases = dev(8)
    up(bases, np.array([src.value], dtype=np.uint64))
    ebuf, eoff = dev(n * 4), dev(n * 4)
    up(ebuf, np.zeros(n, dtype=np.int32))
    up(eoff, np.tile(np.arange(SEG, dtype=np.int32), len(peers)))
    desc = (_Peer * len(peers))()
    for k, r in enumerate(peers):
        desc[k] = _Peer(mapped[r] + rank * SEG * 8, mapped[r] + (world + rank) * SEG * 8, mapped[r] + flags_off + 64 * rank, k * SEG, SEG)
    d_desc = dev(C.sizeof(desc))
    ck(L.hyteg_hip_upload(d_desc, C.c_void_p(C.addressof(desc)), C.sizeof(desc), None), "upload")
    counter, status = dev(4), dev(4)
    up(counter, np.zeros(1, dtype=np.uint32))
    up(status, np.zeros(1, dtype=np.uint32))
    # flag words of the peers inside MY arena, contiguous in the order of `peers` would need a gather: wait peer by peer
    got = np.empty(SEG)
    st = np.zeros(1, dtype=np.uint32)
    for rnd in range(1, ROUNDS + 1):
        up(src, payload(rank, rnd))
        ck(L.hyteg_hip_p2p_pack(d_desc, len(peers), bases, ebuf, eoff, n, rnd, counter, None), "p2p_pack")
        for r in peers:
            ck(L.hyteg_hip_p2p_wait(C.c_void_p(base.value + flags_off + 64 * r), 1, 8, rnd, status, 15000, None), "p2p_wait")
        ck(L.hyteg_hip_download(st.ctypes.data_as(C.c_void_p), status, 4, None), "download")
        ck(L.hyteg_hip_stream_synchronize(None), "sync")
        if st[0]:
            raise RuntimeError(f"round {rnd}: a wait timed out")
        for r in peers:
            off = ((rnd & 1) * world + r) * SEG * 8
            ck(L.hyteg_hip_download(got.ctypes.data_as(C.c_void_p), C.c_void_p(base.value + off), SEG * 8, None), "download")
            ck(L.hyteg_hip_stream_synchronize(None), "sync")
            if not np.array_equal(got, payload(r, rnd)):
                bad = int(np.flatnonzero(got != payload(r, rnd))[0])
                raise RuntimeError(f"round {rnd}: values of rank {r} differ from entry {bad} on ({got[bad]!r} != {payload(r, rnd)[bad]!r})")
    # nobody unmaps or frees before everybody is done
    (d / f"done_{rank}").write_bytes(b"ok")
    while time.monotonic() < deadline and not all((d / f"done_{r}").exists() for r in range(world)):
        time.sleep(0.02)
    for r in peers:
        L.hyteg_hip_p2p_arena_close(C.c_void_p(mapped[r]))
    L.hyteg_hip_p2p_arena_destroy(base)


def launch(rank: int, world: int, device: int, tag: str):
    """start the canary of this rank as a child process (call it before this process initialises its GPU)"""
    import subprocess

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PYTHONPATH=str(root) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.Popen([sys.executable, "-m", "hyteg_amd.p2p_canary", str(rank), str(world), str(device), tag], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=str(root))


def finish(proc, timeout: float = TIMEOUT_S + 30.0):
    """-> (ok, text): exit code 0 within the time limit"""
    try:
        out, _ = proc.communicate(timeout=timeout)
        return proc.returncode == 0, out.decode(errors="replace")[-400:]
    except Exception as e:  # noqa: BLE001
        proc.kill()
        proc.communicate()
        return False, f"canary did not finish: {e!r}"


if __name__ == "__main__":
    try:
        run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    except BaseException as e:  # noqa: BLE001
        print(f"p2p canary rank {sys.argv[1]}: {e!r}", flush=True)
        sys.exit(1)
    print(f"p2p canary rank {sys.argv[1]}: ok", flush=True)
