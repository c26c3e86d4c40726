"""Developer probe (round 2): host and device cost of ONE exchange as the C++ RcclTransport issues it (record 'packed',
comm stream waits, grouped ncclSend/ncclRecv, record 'arrived', compute stream waits), next to the same payload through
torch.distributed's all_to_all_single (the round-1 hook transport).  One GPU holds one RCCL rank, so the peer is the rank
itself (loop-back inside the group): call overhead without any link latency.  Run: python rccl_native_probe.py"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from hyteg_amd import capi  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("librccl:", capi.comm_available())
comm = capi.comm_create(1, 0, capi.comm_unique_id())
n = 3 * 32385  # three level-8 macro-faces: what a rank of the 8-cell mesh sends per apply
send = torch.rand(n, dtype=torch.float64, device=dev)
recv = torch.zeros(n, dtype=torch.float64, device=dev)
side, cur = torch.cuda.Stream(), torch.cuda.current_stream()
packed, arrived = capi.event_create(), capi.event_create()
reps = 300


def native():
    capi.event_record(packed, cur.cuda_stream)
    capi.stream_wait_event(side.cuda_stream, packed)
    capi.comm_exchange(comm, [0, 0, 0], send.data_ptr(), [32385] * 3, recv.data_ptr(), [32385] * 3, side.cuda_stream)
    capi.event_record(arrived, side.cuda_stream)
    capi.stream_wait_event(cur.cuda_stream, arrived)


def hooks():
    w = dist.all_to_all_single(recv, send, [n], [n], async_op=True)
    w.wait()


for name, fn in (("native: events + grouped ncclSend/ncclRecv (3 peers)", native), ("torch all_to_all_single(async) + wait", hooks)):
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
assert torch.equal(recv, send)
v = torch.tensor([1.0], dtype=torch.float64, device=dev)
for _ in range(20):
    capi.comm_allreduce_sum(comm, v.data_ptr(), 1, cur.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    capi.comm_allreduce_sum(comm, v.data_ptr(), 1, cur.cuda_stream)
host_us = (time.perf_counter() - t0) / reps * 1e6
torch.cuda.synchronize()
print(f"{'native ncclAllReduce of one double':56s} host {host_us:7.2f} us per call, until drained {(time.perf_counter() - t0) / reps * 1e6:7.2f} us")
capi.comm_destroy(comm)
dist.destroy_process_group()
