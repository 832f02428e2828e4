"""Multi-GPU layer on CPU: world_size-2 gloo processes exercise unit sharding and the spectrogram-slab all-gather
(ss_amd.dist) exactly as bench.py / a DD-PPO learner drive them (RCCL replaces gloo on the MI355X node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ss_amd.dist import SlabExchange, owner_of, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 16, 128, 129, 515):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
            for u in range(n):
                lo, hi = ranges[owner_of(u, n, world)]
                assert lo <= u < hi


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = SlabExchange((n_local, 65, 26, 2), device="cpu")
        ok = True
        for k in range(steps):
            slab = ex.next_local()
            slab.fill_(float(100 * k + rank))                        # stands in for the renderer's output
            slab[:, 0, 0, 0] = torch.arange(n_local, dtype=torch.float32) + rank * n_local
            full = ex.gather()
            ex.wait()
            assert tuple(full.shape) == (world * n_local, 65, 26, 2)
            for r in range(world):
                blk = full[r * n_local:(r + 1) * n_local]
                ok &= bool((blk[:, 1, 1, 1] == 100 * k + r).all())
                ok &= bool((blk[:, 0, 0, 0] == torch.arange(n_local) + r * n_local).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_slab_all_gather_two_ranks_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 4, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_single_process_exchange_is_identity():
    ex = SlabExchange((3, 65, 26, 2), device="cpu")
    s = ex.next_local()
    s.copy_(torch.randn(3, 65, 26, 2))
    assert torch.equal(ex.gather(), s)
