"""Multi-GPU layer on CPU: world_size-2 gloo processes exercise unit sharding and the spectrogram-slab all-gather
(ss_amd.dist) exactly as bench.py / a DD-PPO learner drive them (RCCL replaces gloo on the MI355X node)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ss_amd.dist import ChunkedSlabExchange, SlabExchange, owner_of, shard_range


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


def _bench_flow_worker(rank, world, port, units, gather_every, steps, q):
    """The exchange flow of bench.py (ChunkedSlabExchange): per-step rows, a gather per full chunk, a partial last chunk."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def gathered(full, n_steps):
            seen.append((full.clone(), n_steps))
        cx = ChunkedSlabExchange(units, (5, 3, 2), gather_every, device="cpu", gathered=gathered)
        for k in range(steps):
            rows = cx.step_rows()
            assert tuple(rows.shape) == (units, 5, 3, 2)
            rows.fill_(float(1000 * rank + k))                       # stands in for the kernels' output of step k
            cx.step_done()
        cx.flush()
        ok = cx.gathers == -(-steps // gather_every) == len(seen)
        k0 = 0
        for full, n_steps in seen:
            ok &= tuple(full.shape) == (world * gather_every * units, 5, 3, 2)
            for r in range(world):
                blk = full[r * gather_every * units:(r + 1) * gather_every * units]
                for i in range(n_steps):                             # rows of step k0 + i of rank r
                    ok &= bool((blk[i * units:(i + 1) * units] == 1000 * r + k0 + i).all())
            k0 += n_steps
        ok &= k0 == steps
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather_every,steps", [(1, 5), (3, 7), (8, 8)])
def test_bench_exchange_flow_two_ranks_gloo(gather_every, steps):
    """the multi-rank flow of bench.py (per-step gather, chunked gather, partial last chunk) under
    world_size 2 on gloo, not only in a gpurun script."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_flow_worker, args=(r, 2, port, 4, gather_every, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _peer_worker(rank, world, port, q):
    """Two processes sharing ONE GPU (the test box has one): peer memory through HIP IPC, gloo as the control plane."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ss_amd.dist import ChunkedSlabExchange, PeerCopyExchange
        seen = []
        cx = ChunkedSlabExchange(4, (65, 26, 2), 3, device="cuda:0", exchange_cls=PeerCopyExchange,
                                 gathered=lambda full, n: seen.append((full, n)))
        ok = True
        for k in range(7):
            rows = cx.step_rows()
            rows.fill_(float(1000 * rank + k))
            cx.step_done()
            if seen:
                full, n_steps = seen.pop()
                cx.exchange.wait()
                torch.cuda.synchronize()
                for r in range(world):
                    blk = full[r * 12:(r + 1) * 12]
                    for i in range(n_steps):
                        ok &= bool((blk[i * 4:(i + 1) * 4] == 1000 * r + (k - n_steps + 1) + i).all())
        cx.flush()
        torch.cuda.synchronize()
        full, n_steps = seen.pop()
        for r in range(world):
            ok &= bool((full[r * 12:r * 12 + 4] == 1000 * r + 6).all()) and n_steps == 1
        q.put((rank, bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _release_worker(rank, world, port, q, learners):
    """No test-side barrier: rank 1's consumer is SLOW (a long busy kernel in front of every check on its compute stream),
    rank 0 races ahead; the checks are device-side accumulations, nothing syncs the host until the end.  Without the
    consumer-release protocol rank 0's copies of round k + 2 land in rank 1's buffer before rank 1 has checked round k."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ss_amd.dist import PeerCopyExchange
        n = 8
        ex = PeerCopyExchange((n, 65, 26, 2), device="cuda:0", learners=learners)
        bad = torch.zeros((), dtype=torch.int32, device="cuda:0")
        rounds = 24
        for k in range(rounds):
            loc = ex.next_local()
            loc.fill_(float(100 * k + rank))
            full = ex.gather()
            ex.wait()
            if full is not None:
                if rank == 1:
                    torch.cuda._sleep(20_000_000)                                 # ~10 ms: the slow consumer
                for r in range(world):
                    bad += (full[r * n:(r + 1) * n] != 100 * k + r).any().to(torch.int32)
            assert (full is None) == (learners is not None and rank not in learners)
        torch.cuda.synchronize()
        q.put((rank, int(bad.item())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("learners", [None, [1]])
def test_peer_copy_consumer_release_without_a_test_side_barrier(learners):
    """the all-gather the reference never had must at least be safe to consume.  all-gather (every rank a
    destination) and gather-to-learner (rank 1, the slow one, is the only destination)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_release_worker, args=(r, 2, port, q, learners)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, 0), (1, 0)]


@pytest.mark.gpu
def test_peer_copy_exchange_two_processes_one_gpu():
    """the peer-write exchange behind the SlabExchange interface (device-to-device copies into the
    peers' buffers through HIP IPC instead of RCCL kernels), exercised with two processes on the one GPU of the box."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_bench_gpus_flag_launches_that_many_ranks():
    """`bench.py --gpus N` parsed the flag and ran one rank.  Without a torch.distributed environment it now
    re-launches itself as N ranks (python -m torch.distributed.run --nproc-per-node N, the shape of the reference's
    ss_baselines/av_nav/single_node.sh:8-11); --dry-run keeps kernels and GPUs out of it (gloo, CPU tensors): the rank
    flow, the chunked slab exchange and the one JSON line are the real ones."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "11",
                          "--gather-every", "4", "--envs", "6"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["gather_ok"] is True and d["gathers"] == 3
    # under a launcher whose world size contradicts --gpus the script refuses instead of mislabelling the line
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


# ---- RCCL on a real multi-GPU box ---------------------------------------------------------------
# One process per GPU, backend "nccl" (= RCCL on ROCm), world size 2: the arrangement of the reference's DD-PPO launcher
# (ss_baselines/av_nav/ddppo/ddppo_trainer.py:140-142, ss_baselines/av_nav/single_node.sh:8-11).  The gpurun boxes have ONE
# GPU, so these skip there; the first multi-GPU box that runs `pytest -m gpu` exercises RCCL without a code change.
def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = f"cuda:{rank}"
    try:
        from ss_amd.dist import ChunkedSlabExchange, PeerCopyExchange, SlabExchange
        ok = True
        n_local = 16
        ex = SlabExchange((n_local, 65, 26, 2), device=dev)                      # all_gather_into_tensor over xGMI
        for k in range(6):
            slab = ex.next_local()
            slab.fill_(float(100 * k + rank))
            full = ex.gather()
            ex.wait()
            torch.cuda.synchronize()
            for r in range(world):
                ok &= bool((full[r * n_local:(r + 1) * n_local] == 100 * k + r).all())
        for cls in (SlabExchange, PeerCopyExchange):                             # the bench's chunked schedule on both transports
            seen = []
            cx = ChunkedSlabExchange(4, (65, 26, 2), 3, device=dev, exchange_cls=cls,
                                     gathered=lambda full, n: seen.append((full, n)))
            for k in range(7):
                cx.step_rows().fill_(float(1000 * rank + k))
                cx.step_done()
                if seen:
                    full, n_steps = seen.pop()
                    cx.exchange.wait()
                    torch.cuda.synchronize()
                    for r in range(world):
                        blk = full[r * 12:(r + 1) * 12]
                        for i in range(n_steps):
                            ok &= bool((blk[i * 4:(i + 1) * 4] == 1000 * r + (k - n_steps + 1) + i).all())
            cx.flush()
            torch.cuda.synchronize()
            full, n_steps = seen.pop()
            for r in range(world):
                ok &= bool((full[r * 12:r * 12 + 4] == 1000 * r + 6).all()) and n_steps == 1
            dist.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _need_two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs on one node (RCCL over xGMI); this box has %d" %
                    (torch.cuda.device_count() if torch.cuda.is_available() else 0))


@pytest.mark.gpu
def test_rccl_slab_and_peer_copy_exchange_two_gpus():
    _need_two_gpus()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["allgather", "none"])
def test_bench_two_gpus_over_rccl(exchange):
    """`bench.py --gpus 2` as the driver would run it on a multi-GPU node: two ranks, RCCL, one JSON line whose value is the
    whole-job aggregate and which still carries the CPU baseline (rank 0)."""
    _need_two_gpus()
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--exchange", exchange], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["roofline"]["frac"] > 0
