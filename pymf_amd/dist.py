"""One-process-per-GPU world description and the rendezvous plumbing.

The data path never touches this module's transport: ranks exchange ONE
128-byte RCCL unique id at start-up (and bench barriers / a max-reduce of
timings); everything per iteration is an ncclAllReduce inside libpymf_hip.
The transport is torch.distributed with the gloo backend (CPU) -- plumbing
only, as the launch contract (`python -m torch.distributed.run ...`) provides
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT for it.
"""
import os

import numpy as np


class World(object):
    """rank/size + the row partition of the reference's m x n `data`."""

    def __init__(self, rank=0, size=1, local_rank=0, nccl_id=None):
        self.rank, self.size, self.local_rank, self.nccl_id = rank, size, local_rank, nccl_id

    def row_range(self, m_global):
        """Contiguous row block of this rank: rows are independent units given H."""
        return shard_rows(m_global, self.rank, self.size)


def shard_rows(m_global, rank, size):
    base, rem = divmod(int(m_global), int(size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_WORLD = World()


def world():
    return _WORLD


def _pg():
    import torch.distributed as dist
    return dist


def init_from_env(make_nccl_id=None):
    """Read the torchrun env, bring up gloo, broadcast rank 0's RCCL unique id."""
    global _WORLD
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if size == 1:
        _WORLD = World(0, 1, local_rank, None)
        return _WORLD
    import torch
    dist = _pg()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=size)
    if make_nccl_id is None:
        from . import _lib
        make_nccl_id = _lib.nccl_unique_id
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(make_nccl_id()), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=0)
    _WORLD = World(rank, size, local_rank, bytes(buf.numpy().tobytes()))
    return _WORLD


def barrier():
    if _WORLD.size > 1:
        _pg().barrier()


def allreduce_max(x):
    if _WORLD.size == 1:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64)
    _pg().all_reduce(t, op=_pg().ReduceOp.MAX)
    return float(t.item())


def allreduce_sum_array(a):
    """CPU all-reduce (gloo) of a float array: used by tests of the sharded formulation."""
    if _WORLD.size == 1:
        return np.asarray(a)
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a).copy())
    _pg().all_reduce(t)
    return t.numpy()


def shutdown():
    global _WORLD
    if _WORLD.size > 1 and _pg().is_initialized():
        _pg().destroy_process_group()
    _WORLD = World()
