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


def plan_shards(n_items: int, n_devices: int, max_batch: int = 0) -> List[Tuple[int, int, int]]:
    """Work list of the single-process dispatcher: (device slot, lo, hi) triples that cover
    [0, n_items) once, in image order.  Every device slot gets the contiguous block of
    ``shard_range``; blocks longer than ``max_batch`` (0 = no limit) are cut into device batches of
    at most that many images, so that a device's planes and workspace stay within its memory."""
    if n_devices <= 0 or n_items < 0 or max_batch < 0:
        raise ValueError("bad shard arguments")
    work = []
    for d in range(n_devices):
        lo, hi = shard_range(n_items, d, n_devices)
        step = max_batch if max_batch else max(hi - lo, 1)
        for a in range(lo, hi, step):
            work.append((d, a, min(a + step, hi)))
    return work


def front_end_sharded(images, devices: Sequence[int] = None, max_batch: int = 256, **kw):
    """The fused front-end of a batch uint8[N,H,W,3] (host array) over several GPUs from ONE process:
    the batched replacement of the per-file loop of receiveFile.py:22 when one process owns the node.

    Images are split by index (``plan_shards``); every device slot has a host thread, a stream, ONE
    planned ``FrontEnd`` per block length (the full blocks share one, a shorter last block gets its own)
    and one pair of page-locked staging buffers, all reused over the slot's blocks; uploads and downloads
    are asynchronous copies on the slot's stream with a single synchronisation per block.  The per-image
    results come back as host arrays concatenated in image order (host gather, no collective).
    ``devices`` defaults to all visible GPUs; a device may be listed more than once (several streams on
    it).  Keyword arguments are those of ``front_end_batch``; returns a ``FrontEndResult`` of numpy arrays
    (``accum`` as a host array too)."""
    import threading

    import torch

    from . import pipeline
    from ._device import DevImage

    imgs = np.asarray(images)
    if imgs.ndim != 4 or imgs.shape[3] != 3 or imgs.dtype != np.uint8:
        raise ValueError("front_end_sharded: uint8[N,H,W,3] expected")
    if devices is None:
        devices = list(range(torch.cuda.device_count()))
    if not devices:
        raise RuntimeError("front_end_sharded: no GPU visible (there is no CPU path)")
    n, h, w = int(imgs.shape[0]), int(imgs.shape[1]), int(imgs.shape[2])
    work = plan_shards(n, len(devices), max_batch)
    kw = dict(kw)
    hough_threshold = kw.pop("hough_threshold", None)
    hough = kw.get("hough", True)
    edges = kw.get("edges", True) or hough
    planes = [np.empty((n, h, w), np.uint8) if (k < 4 or edges) else None for k in range(5)]
    accum = [None] * n
    lines = [None] * n
    errors: List[BaseException] = []

    def run_slot(slot: int):
        try:
            blocks = [(lo, hi) for d, lo, hi in work if d == slot and hi > lo]
            if not blocks or h * w == 0:
                return
            bmax = max(hi - lo for lo, hi in blocks)
            dev = torch.device("cuda", int(devices[slot]))
            with torch.cuda.device(dev):
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    plans = {}  # block length -> FrontEnd (planes, accumulators and workspace of its own)
                    pin_in = torch.empty((bmax, h, w * 3), dtype=torch.uint8, pin_memory=True)
                    pin_out = torch.empty((5, bmax, h, w), dtype=torch.uint8, pin_memory=True)
                    dev_in = DevImage.empty(bmax, h, w, 3)
                    pin_acc = None
                    for lo, hi in blocks:
                        nb = hi - lo
                        fe = plans.get(nb)
                        if fe is None:
                            fe = plans[nb] = pipeline.FrontEnd(nb, h, w, **kw)
                            if fe.hough and pin_acc is None:
                                pin_acc = torch.empty((bmax, fe.na + 2, fe.nr + 2), dtype=torch.int32, pin_memory=True)
                        pin_in.numpy()[:nb] = imgs[lo:hi].reshape(nb, h, w * 3)
                        dev_in.t[:nb, :, : w * 3].copy_(pin_in[:nb], non_blocking=True)
                        fe.run(DevImage(dev_in.t[:nb], nb, h, w, 3))
                        for k in range(5):
                            if planes[k] is not None:
                                pin_out[k, :nb].copy_(fe.plane(k).valid(), non_blocking=True)
                        if fe.hough:
                            pin_acc[:nb].copy_(fe.accum[:nb], non_blocking=True)
                        st.synchronize()  # the one wait of the block: staging is free again after it
                        for k in range(5):
                            if planes[k] is not None:
                                planes[k][lo:hi] = pin_out[k, :nb].numpy()
                        if fe.hough:
                            a = pin_acc[:nb].numpy().copy()
                            for i in range(lo, hi):
                                accum[i] = a[i - lo]
                            if hough_threshold is not None:
                                lines[lo:hi] = fe.lines(hough_threshold)
        except BaseException as ex:  # re-raised by the caller's thread
            errors.append(ex)

    threads = [threading.Thread(target=run_slot, args=(s,)) for s in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    acc = np.stack(accum) if (hough and n) else None
    return pipeline.FrontEndResult(planes[0], planes[1], planes[2], planes[3], planes[4], accum=acc,
                                   lines=lines if hough_threshold is not None and hough else None)
