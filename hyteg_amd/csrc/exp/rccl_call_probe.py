#!/usr/bin/env python3
"""Host and device cost of the collectives the exchange hooks issue, on the real RCCL backend with ONE rank (all a
one-GPU box can hold): python rccl_call_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
send = torch.zeros(33153 * 3, dtype=torch.float64, device=dev)
recv = torch.zeros_like(send)
n = 33153
for name, fn in (
    ("all_to_all_single async + wait", lambda: dist.all_to_all_single(recv[:n], send[:n], [n], [n], async_op=True).wait()),
    ("all_reduce f64[1]", lambda: dist.all_reduce(send[:1])),
):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps * 1e6
    print(f"{name:34s}: host {t_host:7.1f} us per call, {t_all:7.1f} us per call until the device has drained", flush=True)
dist.destroy_process_group()
