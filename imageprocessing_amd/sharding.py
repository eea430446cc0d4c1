"""Per-image sharding of a batch over the GPUs of one node.

Images are independent units of work, so a batch is partitioned by image index
over the ranks (one process per GPU); every rank runs the whole front-end on its
slice and the results are gathered on the host.  There is no collective on the
data path and nothing is reduced across devices (Hough accumulators are per
image).  The reference processes uploads one file at a time in a Python loop
(receiveFile.py:22); this is the batched replacement of that loop.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``: sizes differ by at most one, earlier ranks
    take the larger blocks, empty blocks when world_size > n_items."""
    if world_size <= 0 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_round_robin(n_items: int, rank: int, world_size: int) -> np.ndarray:
    """Image indices rank, rank + world, rank + 2*world, ..."""
    if world_size <= 0 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard arguments")
    return np.arange(rank, n_items, world_size, dtype=np.int64)


def gather_to_rank0(local: np.ndarray, n_items: int, rank: int, world_size: int, group=None):
    """Host gather of per-image results produced with ``shard_range``.

    Each rank passes the array for its own block (first axis = images of the
    block); rank 0 gets the concatenation in image order, other ranks ``None``.
    Uses point-to-point gathers of host tensors (gloo or any backend that
    supports CPU tensors); the device data path never sees a collective.
    """
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return local
    t = torch.from_numpy(np.array(local, order="C", copy=True) if any(st < 0 for st in np.asarray(local).strides)
                         else np.ascontiguousarray(local))
    shapes = [(shard_range(n_items, r, world_size)[1] - shard_range(n_items, r, world_size)[0],) + tuple(local.shape[1:])
              for r in range(world_size)]
    if rank == 0:
        outs = [torch.empty(s, dtype=t.dtype) for s in shapes]
        outs[0] = t
        reqs = [dist.irecv(outs[r], src=r, group=group) for r in range(1, world_size) if shapes[r][0] > 0]
        for q in reqs:
            q.wait()
        return np.concatenate([o.numpy() for o in outs], axis=0)
    if shapes[rank][0] > 0:
        dist.send(t, dst=0, group=group)
    return None


def plane_checksums(planes: np.ndarray) -> np.ndarray:
    """Order-sensitive 64-bit checksum of each image of uint8[N, H, W] (host side)."""
    n = planes.shape[0]
    if n == 0:  # a rank whose block is empty (more ranks than images)
        return np.zeros(0, np.uint64)
    flat = planes.reshape(n, -1).astype(np.uint64)
    idx = (np.arange(flat.shape[1], dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) & np.uint64(0xFFFFFFFF)
    return (flat * idx[None, :]).sum(axis=1, dtype=np.uint64)
