"""torch.distributed plumbing for storages partitioned over several ranks (one process per GPU).

The C++ host layer (hyteg_amd/host/hyteg_host.hpp) packs the partial values of shared macro-face/edge/vertex DoFs
into one send buffer per (level, boundary class), calls the `exchange` hook, and then reduces local and received
values in a fixed order.  This module owns the buffers (torch tensors, so that the RCCL / gloo backends can move
them) and implements the two hooks:

  exchange(level, cls):   dist.all_to_all_single over the registered send/recv tensors, split per peer rank
                          (neighbour exchange: ranks that share no primitive exchange 0 elements);
  allreduce_sum(v, n):    dist.all_reduce(SUM) of n doubles  (walberla::mpi::allReduceInplace in
                          src/hyteg/p1functionspace/VertexDoFFunction.cpp:1717).

Backend "nccl" is RCCL over xGMI on ROCm; backend "gloo" with CPU tensors is used by the CPU tests, which emulate the
pack / reduce kernels with numpy on the exported plan (tests/test_distributed_gloo.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import host


class DistributedContext:
    def __init__(self, storage: host.Storage, levels, device: torch.device | str):
        self.storage = storage
        self.device = torch.device(device)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.buffers = {}
        self._views = {}
        self.plans = {}
        for level in levels:
            for cls in (0, 1):
                p = storage.plan(level, cls)
                self.plans[(level, cls)] = p
                # the exchange is a collective: either every rank takes part in it or none does
                anybody = torch.tensor([1 if len(p["peers"]) else 0], dtype=torch.int32,
                                       device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(anybody, op=dist.ReduceOp.MAX)
                if int(anybody.item()) == 0:
                    continue
                send = torch.zeros(max(1, p["total_send"]), dtype=torch.float64, device=self.device)
                recv = torch.zeros(max(1, p["total_recv"]), dtype=torch.float64, device=self.device)
                in_splits = [0] * self.world
                out_splits = [0] * self.world
                for k, peer in enumerate(p["peers"]):
                    in_splits[int(peer)] = int(p["send_count"][k])
                    out_splits[int(peer)] = int(p["recv_count"][k])
                # views and split lists are built once: the hooks run once per operator application
                self.buffers[(level, cls)] = (send, recv, in_splits, out_splits)
                self._views[(level, cls)] = (send[:sum(in_splits)], recv[:sum(out_splits)])
                if self.device.type == "cuda" and len(p["peers"]):
                    storage.register_comm_buffers(level, cls, send.data_ptr(), recv.data_ptr())
        # gloo cannot move device tensors in all_to_all: stage through pinned host memory (test / fallback transport;
        # the production transport is RCCL, which works on the device buffers directly)
        self._stage = dist.get_backend() == "gloo" and self.device.type == "cuda"
        self._scalar = torch.zeros(8, dtype=torch.float64, device="cpu" if self._stage else self.device)
        self._pending = {}
        storage.set_hooks(self.exchange_begin, self.exchange_end, self.allreduce_sum)

    def send_tensor(self, level, cls):
        return self.buffers[(level, cls)][0]

    def recv_tensor(self, level, cls):
        return self.buffers[(level, cls)][1]

    # ---- hooks ----
    def exchange_begin(self, level: int, cls: int) -> None:
        """start the neighbour all-to-all; the pack kernel ran on torch's current stream (= the storage's stream) and
        the collective orders itself after it.  Returns immediately: kernels launched next overlap the transfer."""
        if (level, cls) not in self.buffers:
            return
        send, recv, in_splits, out_splits = self.buffers[(level, cls)]
        n_in, n_out = sum(in_splits), sum(out_splits)
        if self._stage:
            host_send = send[:n_in].cpu()  # synchronises with the pack kernel on the current stream
            host_recv = torch.empty(n_out, dtype=torch.float64)
            work = dist.all_to_all_single(host_recv, host_send, out_splits, in_splits, async_op=True)
            self._pending[(level, cls)] = (work, host_recv, recv, n_out)
        else:
            send_v, recv_v = self._views[(level, cls)]
            self._pending[(level, cls)] = dist.all_to_all_single(recv_v, send_v, out_splits, in_splits, async_op=True)

    def exchange_end(self, level: int, cls: int) -> None:
        work = self._pending.pop((level, cls), None)
        if work is None:
            return
        if self._stage:
            w, host_recv, recv, n_out = work
            w.wait()
            recv[:n_out].copy_(host_recv)
        else:
            work.wait()  # makes the current stream wait for the collective (no host sync on the nccl backend)

    def exchange(self, level: int, cls: int) -> None:
        self.exchange_begin(level, cls)
        self.exchange_end(level, cls)

    def allreduce_sum(self, values, n: int) -> None:
        arr = np.ctypeslib.as_array(values, shape=(n,))
        t = self._scalar[:n] if n <= self._scalar.numel() else torch.zeros(n, dtype=torch.float64, device=self._scalar.device)
        t.copy_(torch.from_numpy(arr.copy()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        arr[:] = t.cpu().numpy()
