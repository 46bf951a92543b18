"""Communicator bootstrap for the row-block shard.

One process per GPU.  ``torch.distributed`` is only the control plane here: it ships the
128-byte RCCL unique id (NCCL backend) or carries the host collectives themselves (gloo
backend — used by the tests that run several ranks on ONE GPU, where RCCL refuses
duplicate devices).  The data path is libgpx.so calling RCCL on its own HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi

OP_SUM, OP_MIN = 0, 1


def init_rccl(lib, handle, rank, world, group=None):
    """Rank 0 draws the unique id, every rank joins: gpx_comm_unique_id / gpx_comm_init."""
    import torch.distributed as dist
    _abi.prefer_torch_rccl()            # same RCCL build torch itself runs on: one RCCL per process
    uid = C.create_string_buffer(128)
    if rank == 0:
        rc = lib.gpx_comm_unique_id(uid)
        if rc != 0:
            raise _abi.GpxError(rc, lib.gpx_last_error(None).decode())
    if world > 1:
        box = [uid.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = C.create_string_buffer(box[0], 128)
    rc = lib.gpx_comm_init(handle, uid)
    if rc != 0:
        raise _abi.GpxError(rc, lib.gpx_last_error(handle).decode())


class HostCollectives:
    """gpx_host_comm implemented with torch.distributed on CPU tensors (gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.last_error = None
        # keep the CFUNCTYPE objects alive as long as the handle may call them
        self._fns = (_abi.BCAST_FN(self._bcast), _abi.ALLGATHER_FN(self._allgather),
                     _abi.REDUCE_FN(self._reduce), _abi.ALLREDUCE_FN(self._allreduce))
        self.vtable = _abi.GpxHostComm(None, *self._fns)

    def _view(self, addr, nbytes, dtype=np.uint8):
        buf = (C.c_uint8 * nbytes).from_address(addr)
        return self.torch.from_numpy(np.frombuffer(buf, dtype=dtype))

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.last_error = e
            return 1

    def _bcast(self, ctx, buf, nbytes, root):
        return self._guard(lambda: self.dist.broadcast(self._view(buf, nbytes), src=root, group=self.group))

    def _allgather(self, ctx, send, recv, nbytes):
        def run():
            out = self._view(recv, nbytes * self.world)
            self.dist.all_gather(list(out.chunk(self.world)), self._view(send, nbytes), group=self.group)
        return self._guard(run)

    def _op(self, op):
        return self.dist.ReduceOp.MIN if op == OP_MIN else self.dist.ReduceOp.SUM

    def _reduce(self, ctx, send, recv, count, root, op):
        def run():
            t = self._view(send, count * 8, np.float64).clone()
            self.dist.reduce(t, dst=root, op=self._op(op), group=self.group)
            if self.rank == root:
                self._view(recv, count * 8, np.float64).copy_(t)
        return self._guard(run)

    def _allreduce(self, ctx, buf, count, op):
        return self._guard(lambda: self.dist.all_reduce(self._view(buf, count * 8, np.float64),
                                                        op=self._op(op), group=self.group))

    def attach(self, lib, handle):
        rc = lib.gpx_comm_init_host(handle, C.byref(self.vtable))
        if rc != 0:
            raise _abi.GpxError(rc, lib.gpx_last_error(handle).decode())


# ---- bookkeeping of the dealing (mirror of struct Deal in csrc/gpx_internal.h) ------------------------------------
# snake (round 4, the library's default): rounds of 2 P row blocks dealt 0, 1, ..., P-1, P-1, ..., 1, 0 — a row block's
# share of every trailing update grows with its index, so the cyclic dealing (block g on rank g mod P) loads the last rank
# 8 % (P = 8, nb = 512, N = 65536) above the mean in every panel; the snake's pairs sum to the same index on every rank.
def owner(g, P, snake=True):
    if not snake:
        return g % P
    pos = g % (2 * P)
    return pos if pos < P else 2 * P - 1 - pos


def local_index(g, P, snake=True):
    """index of block g among its owner's blocks"""
    if not snake:
        return g // P
    return 2 * (g // (2 * P)) + (1 if g % (2 * P) >= P else 0)


def blocks_owned(rank, nblk, P, snake=True):
    """global block indices stored by `rank`, in local order"""
    return [g for g in range(nblk) if owner(g, P, snake) == rank]


def lb0(p, rank, P, snake=True):
    """number of blocks owned by `rank` with global index <= p"""
    if not snake:
        return (p - rank) // P + 1 if p >= rank else 0
    n = p + 1
    c, rem = divmod(n, 2 * P)
    return 2 * c + (1 if rem > rank else 0) + (1 if rem > 2 * P - 1 - rank else 0)
