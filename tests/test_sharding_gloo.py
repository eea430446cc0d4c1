"""Per-image sharding across ranks: pure index logic plus a 2-process gloo run on the CPU
(the data path has no collective; the gather is a host-side point-to-point gather)."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from imageprocessing_amd.sharding import (gather_to_rank0, plan_shards, plane_checksums, shard_range,  # noqa: E402
                                          shard_round_robin)


@pytest.mark.parametrize("n,world", [(0, 1), (1, 1), (1, 8), (7, 2), (8, 8), (4096, 8), (1024, 3), (5, 4)])
def test_shard_range_partitions_exactly(n, world):
    blocks = [shard_range(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (a, b), (c, d) in zip(blocks, blocks[1:]):
        assert b == c and a <= b
    sizes = [b - a for a, b in blocks]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


def test_shard_round_robin_covers_everything():
    n, world = 21, 4
    idx = np.concatenate([shard_round_robin(n, r, world) for r in range(world)])
    assert sorted(idx.tolist()) == list(range(n))
    assert shard_round_robin(3, 3, 4).tolist() == []
    with pytest.raises(ValueError):
        shard_range(5, 2, 2)


@pytest.mark.parametrize("n,devs,mb", [(0, 1, 0), (1, 8, 0), (4096, 8, 256), (4096, 8, 0), (1000, 3, 128), (5, 4, 1), (7, 2, 3)])
def test_plan_shards_covers_every_image_once_in_order(n, devs, mb):
    # index logic of the single-process multi-GPU dispatcher (front_end_sharded)
    work = plan_shards(n, devs, mb)
    covered = [i for _, lo, hi in work for i in range(lo, hi)]
    assert covered == list(range(n))
    for d, lo, hi in work:
        a, b = shard_range(n, d, devs)
        assert a <= lo < hi <= b and (mb == 0 or hi - lo <= mb)
    if n >= devs:
        assert {d for d, _, _ in work} == set(range(devs))
    with pytest.raises(ValueError):
        plan_shards(4, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import frontref as fr
        from imageprocessing_amd.synth import synth_diagram

        lo, hi = shard_range(n_items, rank, world)
        # each rank runs the (CPU oracle stand-in for the) front-end on ITS images only
        edges = np.stack([fr.canny(fr.bgr2gray(synth_diagram(100 + i, 48, 64)), 50, 150) for i in range(lo, hi)]) \
            if hi > lo else np.zeros((0, 48, 64), np.uint8)
        sums = plane_checksums(edges).astype(np.int64)
        all_edges = gather_to_rank0(edges, n_items, rank, world)
        all_sums = gather_to_rank0(sums, n_items, rank, world)
        # the only collective: a count of processed images (control plane, not data path)
        total = torch.tensor([hi - lo], dtype=torch.int64)
        dist.all_reduce(total)
        assert int(total) == n_items
        if rank == 0:
            q.put((all_edges, all_sums))
            time.sleep(0.5)  # let the peer drain its side of the last exchange before teardown
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 2, 1])
def test_two_rank_gloo_gather_matches_single_process(n_items):
    from oracle import frontref as fr
    from imageprocessing_amd.synth import synth_diagram

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    all_edges, all_sums = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.stack([fr.canny(fr.bgr2gray(synth_diagram(100 + i, 48, 64)), 50, 150) for i in range(n_items)])
    assert np.array_equal(all_edges, ref)
    assert np.array_equal(all_sums, plane_checksums(ref).astype(np.int64))
