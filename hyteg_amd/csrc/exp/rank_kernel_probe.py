"""Developer probe (round 2): what a rank with ONE level-8 macro-cell does per apply before/with the peer-to-peer exchange,
timed in one process with loop-back peers (the own arena; no link latency, no second process):
  (a) four launches: boundary shares, p2p pack, interior, wait + reduce
  (b) two launches: hyteg_hip_p1_apply_cell_rank (interior bricks + shares + pack in one grid), wait + reduce
The reduce here is a stand-in with the right size and shape (groups of two copies: own value + received value).
Needs hyteg_hip_p1_apply_cell_rank in the library (see the head of exp/p1_apply_rank.hip: it is not built by default).
HYTEG_HIP_RANK_DBG: 1 pack does not wait, 2 plain share stores, 4 no share work, 8 no pack work, 16 no bricks (interior launched here).
Run: python rank_kernel_probe.py [nfaces]"""
import ctypes as C
import os
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
ck = capi.check
level = 8
N = 2**level + 1
ncell = capi.cell_size(level)
nfaces = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cur = torch.cuda.current_stream().cuda_stream
capi.prepare_level(level)


class Peer(C.Structure):
    _fields_ = [("slot0", C.c_void_p), ("slot1", C.c_void_p), ("flag", C.c_void_p), ("start", C.c_int), ("count", C.c_int)]


# shared points: face z = 0 (slot 6), y = 0 (slot 7), x = 0 (slot 8): their array offsets
def face_offsets(f):
    offs = []
    for a in range(N):
        for b in range(N - a):
            x, y, z = {0: (b, a, 0), 1: (b, 0, a), 2: (0, b, a)}[f]
            if f == 0 and (y == 0 or x == 0 or x + y == N - 1):
                continue
            if f == 1 and (z == 0 or x == 0 or x + z == N - 1):
                continue
            if f == 2 and (z == 0 or y == 0 or y + z == N - 1):
                continue
            offs.append(capi.cell_index(level, x, y, z))
    return offs


segs = [np.array(face_offsets(f), dtype=np.int32) for f in range(nfaces)]
n = int(sum(len(s) for s in segs))
mask_shell = sum(1 << (6 + f) for f in range(nfaces))
mask = mask_shell | (1 << 14)
base, handle = C.c_void_p(), C.create_string_buffer(64)
flags_off = 2 * n * 8
ck(L.hyteg_hip_p2p_arena_create(flags_off + 4096, C.byref(base), handle, None), "arena")
peers = (Peer * nfaces)()
start = 0
for p in range(nfaces):
    peers[p] = Peer(base.value + start * 8, base.value + (n + start) * 8, base.value + flags_off + 64 * p, start, len(segs[p]))
    start += len(segs[p])
d_peers = torch.frombuffer(bytearray(bytes(peers)), dtype=torch.uint8).to(dev)
nbuf = 9
srcs = [torch.rand(ncell, dtype=torch.float64, device=dev) for _ in range(nbuf)]
dsts = [torch.zeros(ncell, dtype=torch.float64, device=dev) for _ in range(nbuf)]
eoff = torch.from_numpy(np.concatenate(segs)).to(dev)
ebuf = torch.zeros(n, dtype=torch.int32, device=dev)
counters = torch.zeros(2, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
w = (C.c_double * 15)(*np.random.default_rng(1).random(15))
ws = (C.c_double * 210)(*np.random.default_rng(2).random(210))
# reduce stand-in: group g = { own copy (buffer 0 = dst, offset eoff[g]), received copy (buffer 1 = slot, offset g) }
gptr = torch.arange(0, 2 * n + 1, 2, dtype=torch.int32, device=dev)
rbuf = torch.tensor([0, 1] * n, dtype=torch.int32, device=dev)
roff = torch.stack([eoff, torch.arange(n, dtype=torch.int32, device=dev)], dim=1).reshape(-1).contiguous()
tables = {}
seq = [0]


def bases_for(k, parity):
    key = (k, parity)
    if key not in tables:
        tables[key] = torch.tensor([dsts[k].data_ptr(), base.value + parity * n * 8], dtype=torch.int64, device=dev)
    return tables[key]


def reduce(k):
    b = bases_for(k, seq[0] & 1)
    ck(L.hyteg_hip_reduce_shared_after_p2p(b.data_ptr(), gptr.data_ptr(), rbuf.data_ptr(), roff.data_ptr(), n, 1, 1, base.value + flags_off,
                                           nfaces, 8, seq[0], status.data_ptr(), 2000, cur), "reduce")


def four(k):
    seq[0] += 1
    b = bases_for(k, 0)
    ck(L.hyteg_hip_p1_apply_cell_boundary(dsts[k].data_ptr(), srcs[k].data_ptr(), level, ws, mask_shell, 0, cur), "boundary")
    ck(L.hyteg_hip_p2p_pack(d_peers.data_ptr(), nfaces, b.data_ptr(), ebuf.data_ptr(), eoff.data_ptr(), n, seq[0], counters.data_ptr(), cur), "pack")
    ck(L.hyteg_hip_p1_apply_cell(dsts[k].data_ptr(), srcs[k].data_ptr(), level, w, 0, cur), "apply")
    reduce(k)


def two(k):
    seq[0] += 1
    b = bases_for(k, 0)
    ck(L.hyteg_hip_p1_apply_cell_rank(dsts[k].data_ptr(), srcs[k].data_ptr(), level, w, ws, mask, 0, d_peers.data_ptr(), nfaces, b.data_ptr(),
                                      ebuf.data_ptr(), eoff.data_ptr(), n, seq[0], counters.data_ptr(), status.data_ptr(), 2000, cur), "rank")
    if int(os.environ.get("HYTEG_HIP_RANK_DBG", "0")) & 16:
        interior(k)
    reduce(k)


# CSR over the shell enumeration of the boundary kernel: q = face * tri(N) + row_start(N, j) + i -> entries of the send enumeration
T = N * (N + 1) // 2
by_off = {}
for kk, off in enumerate(np.concatenate(segs)):
    by_off.setdefault(int(off), []).append(kk)
first_h = np.zeros(4 * T + 1, dtype=np.int32)
list_h = []
for f in range(4):
    for j in range(N):
        for i in range(N - j):
            q = f * T + (j * N - j * (j - 1) // 2) + i
            x, y, z = {0: (i, j, 0), 1: (i, 0, j), 2: (0, i, j), 3: (i, j, N - 1 - i - j)}[f]
            lowest = 0 if z == 0 else (1 if y == 0 else (2 if x == 0 else 3))
            if lowest == f:
                list_h += by_off.get(capi.cell_index(level, x, y, z), [])
            first_h[q + 1] = len(list_h)
assert len(list_h) == n
first_d = torch.from_numpy(first_h).to(dev)
list_d = torch.tensor(list_h + [0], dtype=torch.int32, device=dev)


def three(k):
    seq[0] += 1
    ck(L.hyteg_hip_p1_apply_cell_boundary_p2p(dsts[k].data_ptr(), srcs[k].data_ptr(), level, ws, mask_shell, 0, first_d.data_ptr(), list_d.data_ptr(),
                                              d_peers.data_ptr(), nfaces, seq[0], counters.data_ptr(), cur), "boundary + send")
    ck(L.hyteg_hip_p1_apply_cell(dsts[k].data_ptr(), srcs[k].data_ptr(), level, w, 0, cur), "apply")
    reduce(k)


def interior(k):
    ck(L.hyteg_hip_p1_apply_cell(dsts[k].data_ptr(), srcs[k].data_ptr(), level, w, 0, cur), "apply")


have_rank = hasattr(L, "hyteg_hip_p1_apply_cell_rank")
# same results
four(0)
torch.cuda.synchronize()
ref = dsts[0].clone()
dsts[0].zero_()
if have_rank:
    two(0)
torch.cuda.synchronize()
if have_rank and not os.environ.get("HYTEG_HIP_RANK_DBG"):
    assert torch.equal(ref, dsts[0]) and int(status.item()) == 0, "rank kernel differs from the four launches"
reps = 400
print(f"level {level}, {nfaces} shared face(s), {n} values per exchange")
dsts[0].zero_()
three(0)
torch.cuda.synchronize()
assert torch.equal(ref, dsts[0]) and int(status.item()) == 0, "shares + send differs from the four launches"
for name, fn in (("interior kernel alone", interior), ("four launches (shares, pack, interior, wait+reduce)", four),
                 ("three launches (shares + send, interior, wait+reduce)", three), ("four launches once more", four),
                 ("three launches once more", three),
                 ("two launches (rank kernel, wait+reduce)", two), ("four launches again", four), ("two launches again", two)):
    if fn is two and not have_rank:
        continue
    for k in range(2 * nbuf):
        fn(k % nbuf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        fn(k % nbuf)
    host_us = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print(f"{name:56s} host {host_us:7.2f} us per apply, until drained {(time.perf_counter() - t0) / reps * 1e6:7.2f} us")
assert int(status.item()) == 0
