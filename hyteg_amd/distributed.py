"""torch.distributed plumbing for storages partitioned over several ranks (one process per GPU).

Two transports sit behind the shared-point exchange of the C++ host layer (hyteg_amd/host/comm.hpp):

  "rccl"   the production path.  The host layer issues ncclSend / ncclRecv groups and ncclAllReduce itself (C-ABI
           hyteg_hip_comm_*, RCCL over xGMI) on its own communication stream, ordered against the compute stream with
           events -- nothing of an exchange passes through Python.  torch.distributed is only used here to hand the
           128-byte unique id of rank 0 to the other ranks (what MPI_Bcast would do inside HyTeG).
  "hooks"  the test path.  The host layer packs the partial values of shared macro-face/edge/vertex DoFs into one send
           buffer per (level, plan), calls the `exchange` hook, and then reduces local and received values in a fixed
           order.  This module owns the buffers (torch tensors) and implements the hooks with
           dist.all_to_all_single (neighbour exchange: ranks that share no primitive exchange 0 elements) and
           dist.all_reduce (walberla::mpi::allReduceInplace in src/hyteg/p1functionspace/VertexDoFFunction.cpp:1717).
           Backend "gloo" moves them through host memory: this is how the CPU tests and the two-ranks-on-one-GPU tests
           run the multi-rank logic without one GPU per rank.

  "p2p"    on top of either of the two (enable_p2p, or transport="p2p"): the pack kernel stores straight into receive slots
           inside the neighbour GPUs' IPC-mapped arenas and a one-wave kernel waits for their sequence numbers
           (hyteg_amd/csrc/comm_p2p.hip) -- no library call per exchange.  torch.distributed carries the set-up only: the
           64-byte arena handles and, per plan, where every peer expects this rank's values.  The all-reduce of dot
           products and plans of levels that were not connected stay with the transport underneath.

A plan is identified by (level, key) with key = cls + 2 * dof_kind (boundary class 0 / 1; vertex DoFs / edge DoFs of P2
functions)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import capi, host


class DistributedContext:
    def __init__(self, storage: host.Storage, levels, device: torch.device | str, transport: str = "auto", dof_kinds=(0,),
                 stream: int | None = None):
        self.storage = storage
        self.device = torch.device(device)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        want_p2p = transport == "p2p"
        if want_p2p:
            transport = "auto"
        auto = transport == "auto"
        self.transport_note = ""
        self._levels, self._dof_kinds = tuple(levels), tuple(dof_kinds)
        if auto:
            transport = "hooks"
            if dist.get_backend() == "nccl" and self.device.type == "cuda":
                # every rank must be able to load librccl BEFORE the collective communicator set-up is entered
                if self._all_ranks(capi.comm_available()):
                    transport = "rccl"
                else:
                    self.transport_note = "librccl could not be loaded on every rank: exchange through torch.distributed hooks"
        self.transport = transport
        self.buffers = {}
        self._views = {}
        self.plans = {}
        self._pending = {}
        if transport == "rccl":
            # the unique id of rank 0 reaches the other ranks through torch.distributed's store; the communicator is
            # created (collectively) on this rank's current device
            ids = [capi.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            err = None
            try:
                storage.use_rccl(ids[0])
            except Exception as e:  # noqa: BLE001 -- reported below, on every rank
                err = e
            if self._all_ranks(err is None):
                if want_p2p:
                    self.enable_p2p()
                return
            if not auto:
                raise RuntimeError(f"RCCL transport could not be set up on every rank (this rank: {err!r})")
            # "auto": all ranks switch together (set_hooks below replaces the transport of the ranks that did succeed)
            self.transport = transport = "hooks"
            self.transport_note = f"RCCL communicator set-up failed on some rank (this rank: {err!r}): exchange through torch.distributed hooks"
        if transport != "hooks":
            raise ValueError(f"unknown transport {transport!r}")
        # the hooks run on the stream the host layer launches its pack / reduce kernels on
        self._stream = None
        if self.device.type == "cuda":
            raw = storage.stream if stream is None else stream
            self._stream = torch.cuda.ExternalStream(raw, device=self.device) if raw else torch.cuda.default_stream(self.device)
        for level in levels:
            for dof_kind in dof_kinds:
                for cls in (0, 1):
                    key = cls + 2 * dof_kind
                    p = storage.plan(level, key)
                    self.plans[(level, key)] = p
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
                    self.buffers[(level, key)] = (send, recv, in_splits, out_splits)
                    self._views[(level, key)] = (send[:sum(in_splits)], recv[:sum(out_splits)])
                    if self.device.type == "cuda" and len(p["peers"]):
                        storage.register_comm_buffers(level, key, send.data_ptr(), recv.data_ptr())
        # gloo cannot move device tensors in all_to_all: stage through host memory
        self._stage = dist.get_backend() == "gloo" and self.device.type == "cuda"
        self._scalar = torch.zeros(8, dtype=torch.float64, device="cpu" if self._stage else self.device)
        storage.set_hooks(self.exchange_begin, self.exchange_end, self.allreduce_sum)
        if want_p2p:
            self.enable_p2p()

    # ---- peer-to-peer transport on top of the one set up above ----
    def enable_p2p(self, levels=None, dof_kinds=None) -> bool:
        """Collective.  Connects the plans of `levels` x `dof_kinds` (default: those given to the constructor) peer to
        peer.  Returns False -- on every rank, with the storage back on the transport underneath and the reason in
        transport_note -- if any step fails on any rank (no IPC between the ranks' devices, arena allocation, ...)."""
        if self.device.type != "cuda":
            raise RuntimeError("the peer-to-peer transport needs device memory")
        levels = self._levels if levels is None else tuple(levels)
        dof_kinds = self._dof_kinds if dof_kinds is None else tuple(dof_kinds)
        st = self.storage
        keys = [(lv, cls + 2 * dk) for lv in levels for dk in dof_kinds for cls in (0, 1)]
        plans = {k: st.plan(*k) for k in keys}
        align = 256
        need = 4096 + sum(2 * -(-max(8, 8 * int(p["total_recv"])) // align) * align + -(-max(1, 64 * len(p["peers"])) // align) * align
                          for p in plans.values())

        def step(fn):
            err, out = None, None
            try:
                out = fn()
            except Exception as e:  # noqa: BLE001 -- agreed on below, on every rank
                err = e
            return out, err

        (res, err) = step(lambda: st.use_p2p(need))
        if self._all_ranks(err is None):
            handle, kind = res
            handles = [None] * self.world
            dist.all_gather_object(handles, handle)
            _, err = step(lambda: st.p2p_open(handles))
        if self._all_ranks(err is None):
            def lay():
                mine = {}
                for k, p in plans.items():
                    o = st.p2p_layout(k[0], k[1], len(p["peers"]))
                    mine[k] = {int(peer): [int(x) for x in o[j]] for j, peer in enumerate(p["peers"])}
                return mine
            mine, err = step(lay)
        if self._all_ranks(err is None):
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine)

            def connect():
                for k, p in plans.items():
                    st.p2p_connect(k[0], k[1], [everyone[int(peer)][k][self.rank] for peer in p["peers"]])
            _, err = step(connect)
        if self._all_ranks(err is None):
            self.inner_transport = self.transport
            self.transport = "p2p"
            self.p2p_arena = {"bytes": need, "kind": ("uncached", "fine-grained", "default")[kind]}
            return True
        st.drop_p2p()
        self.transport_note = f"peer-to-peer transport not available on every rank (this rank: {err!r}): staying on {self.transport}"
        return False

    def disable_p2p(self, note: str = "") -> None:
        """back to the transport underneath (collective by convention: every rank must take the same decision)"""
        if self.transport == "p2p":
            self.storage.drop_p2p()
            self.transport = self.inner_transport
            if note:
                self.transport_note = note

    def _all_ranks(self, ok: bool) -> bool:
        """logical AND over the ranks of a per-rank success flag"""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def send_tensor(self, level, key):
        return self.buffers[(level, key)][0]

    def recv_tensor(self, level, key):
        return self.buffers[(level, key)][1]

    def _on_stream(self):
        return torch.cuda.stream(self._stream) if self._stream is not None else _NullContext()

    # ---- hooks (an exception raised here fails the host-layer operation that called the hook) ----
    def exchange_begin(self, level: int, key: int) -> None:
        """start the neighbour all-to-all behind the pack kernel (same stream).  Returns immediately: kernels launched
        next overlap the transfer."""
        if (level, key) not in self.buffers:
            raise KeyError(f"exchange of level {level}, plan {key}: no buffers were set up for it "
                           f"(levels / dof_kinds given to DistributedContext)")
        send, recv, in_splits, out_splits = self.buffers[(level, key)]
        n_in, n_out = sum(in_splits), sum(out_splits)
        with self._on_stream():
            if self._stage:
                host_send = send[:n_in].cpu()  # synchronises with the pack kernel on the stream
                host_recv = torch.empty(n_out, dtype=torch.float64)
                work = dist.all_to_all_single(host_recv, host_send, out_splits, in_splits, async_op=True)
                self._pending[(level, key)] = (work, host_recv, recv, n_out)
            else:
                send_v, recv_v = self._views[(level, key)]
                self._pending[(level, key)] = dist.all_to_all_single(recv_v, send_v, out_splits, in_splits, async_op=True)

    def exchange_end(self, level: int, key: int) -> None:
        work = self._pending.pop((level, key))
        with self._on_stream():
            if self._stage:
                w, host_recv, recv, n_out = work
                w.wait()
                recv[:n_out].copy_(host_recv)
            else:
                work.wait()  # makes the stream wait for the collective (no host sync on the nccl backend)

    def exchange(self, level: int, key: int) -> None:
        self.exchange_begin(level, key)
        self.exchange_end(level, key)

    def allreduce_sum(self, values, n: int) -> None:
        arr = np.ctypeslib.as_array(values, shape=(n,))
        t = self._scalar[:n] if n <= self._scalar.numel() else torch.zeros(n, dtype=torch.float64, device=self._scalar.device)
        t.copy_(torch.from_numpy(arr.copy()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        arr[:] = t.cpu().numpy()


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
