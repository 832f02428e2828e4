"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The audio path shards perfectly over (env, rotation) units — each unit reads its own RIR, sources are a small
replicated bank — so ranks own contiguous blocks of units and render them with no data-path collective, exactly like
the reference's DD-PPO where each rank owns its envs (ss_baselines/av_nav/ddppo/ddppo_trainer.py:140-142).
The one exchange step (new; BASELINE.json north_star) is the all-gather of the per-rank spectrogram slab
[N/G, 65, T4, 2] into the learner-side [N, 65, T4, 2] tensor.  The slab is small (1.7 MB for 128 envs @16 kHz), so the
gather is issued on a side stream and double-buffered: step k's gather overlaps step k+1's convolution kernel.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(unit: int, n_units: int, world: int) -> int:
    base, rem = divmod(n_units, world)
    edge = rem * (base + 1)
    return unit // (base + 1) if unit < edge else rem + (unit - edge) // max(base, 1)


class SlabExchange:
    """All-gather of equally-shaped per-rank slabs, double-buffered on a side stream (GPU) or inline (CPU/gloo)."""

    def __init__(self, slab_shape, dtype=torch.float32, device="cuda", group: Optional[dist.ProcessGroup] = None,
                 depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.slab_shape = tuple(slab_shape)
        full = (self.world * self.slab_shape[0],) + self.slab_shape[1:]
        self.depth = depth
        self.local = [torch.empty(self.slab_shape, dtype=dtype, device=self.device) for _ in range(depth)]
        self.full = [torch.empty(full, dtype=dtype, device=self.device) for _ in range(depth)]
        self._k = 0
        self._cuda = self.device.type == "cuda"
        if self._cuda:
            self.stream = torch.cuda.Stream(device=self.device)
            self.ready = [torch.cuda.Event() for _ in range(depth)]     # gather k finished

    def _producers(self, streams):
        return list(streams) if streams else [torch.cuda.current_stream(self.device)]

    def next_local(self, streams=None) -> torch.Tensor:
        """Buffer the renderer should write this step's slab into (waits for the gather that last used it).
        `streams`: the compute streams that will write into it (default: the current stream)."""
        i = self._k % self.depth
        if self._cuda and self._k >= self.depth:
            for st in self._producers(streams):
                st.wait_event(self.ready[i])
        return self.local[i]

    def gather(self, streams=None) -> torch.Tensor:
        """Issue the all-gather of the slab handed out by the last next_local(); returns the [world*n, ...] tensor
        (valid on the side stream after self.ready[i]; call wait() before consuming on the compute stream).
        `streams`: every compute stream that wrote part of the slab (default: the current stream)."""
        i = self._k % self.depth
        self._k += 1
        if self.world == 1:
            self.full[i] = self.local[i]
            return self.full[i]
        if self._cuda:
            for st in self._producers(streams):             # the slab is complete when all of its writers are
                ev = torch.cuda.Event()
                ev.record(st)
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                dist.all_gather_into_tensor(self.full[i], self.local[i], group=self.group)
                self.ready[i].record(self.stream)
        else:
            try:
                dist.all_gather_into_tensor(self.full[i], self.local[i], group=self.group)
            except (RuntimeError, NotImplementedError):
                parts = list(self.full[i].chunk(self.world, dim=0))
                dist.all_gather(parts, self.local[i], group=self.group)
        return self.full[i]

    def wait(self, streams=None) -> None:
        """Make the compute stream(s) wait for the most recent gather."""
        if self._cuda and self.world > 1 and self._k > 0:
            for st in self._producers(streams):
                st.wait_event(self.ready[(self._k - 1) % self.depth])


class ChunkedSlabExchange:
    """The exchange schedule of bench.py / a rollout learner: every rank renders `units` rows per step; the rows of
    `gather_every` consecutive steps form one slab that is all-gathered at once (a learner consumes rollouts, not single
    steps; fewer, larger collectives suit the point-to-point xGMI fabric).  gather_every = 1 is the per-step gather.

        buf = cx.step_rows(streams)        # [units, ...] view the renderer writes this step's rows into
        ... launch the kernels ...
        cx.step_done(streams)              # issues the all-gather when the chunk is full
        cx.flush(streams)                  # end of the run: gather a partial last chunk, wait for everything

    `gathered` is called with (full slab [world * gather_every * units, ...], number of valid steps in it) on the host
    right after a gather was ISSUED (consume it on a stream after `exchange.wait()`)."""

    def __init__(self, units: int, row_shape, gather_every: int, dtype=torch.float32, device="cuda",
                 group: Optional[dist.ProcessGroup] = None, exchange_cls=None, gathered=None, **exchange_kwargs):
        self.units, self.G = int(units), max(1, int(gather_every))
        cls = exchange_cls or SlabExchange
        self.exchange = cls((self.G * self.units,) + tuple(row_shape), dtype=dtype, device=device, group=group,
                            **exchange_kwargs)
        self._buf = None
        self._n = 0
        self.gathered = gathered
        self.gathers = 0

    def step_rows(self, streams=None) -> torch.Tensor:
        if self._n == 0:
            self._buf = self.exchange.next_local(streams)
        i = self._n
        return self._buf[i * self.units:(i + 1) * self.units]

    def will_gather(self) -> bool:
        """True when the next step_done() completes the chunk, i.e. issues the all-gather (callers with work on streams
        the exchange does not know - ss_ctx overlap lanes - join them first)."""
        return self._n + 1 == self.G

    def step_done(self, streams=None) -> None:
        self._n += 1
        if self._n == self.G:
            self._gather(streams)

    def _gather(self, streams):
        full = self.exchange.gather(streams)
        self.gathers += 1
        if self.gathered is not None:
            self.gathered(full, self._n)
        self._n = 0

    def flush(self, streams=None) -> None:
        if self._n:
            self._gather(streams)
        self.exchange.wait(streams)


class PeerCopyExchange(SlabExchange):
    """Same interface as ``SlabExchange``, different transport: every rank WRITES its slab straight into the other
    ranks' gathered buffers with device-to-device copies over xGMI (peer memory mapped through HIP IPC), and only the
    synchronisation is a collective.

    Why: RCCL's all-gather runs as kernels that need CUs with ~20 KB of LDS and ~270 registers per thread; they can
    never share a CU with a ``k_conv`` workgroup (148 KB of LDS, all of the register file), so while a gather is in
    flight a 256-row launch finds fewer than 256 free CUs.  Peer copies are executed by the
    SDMA engines / blit path: no CUs, no LDS.  The slabs are small (1.7 MB per rank and step at 128 envs), every GPU
    has a direct xGMI link to every other one, so each rank issues one independent copy per destination, one per link.

    ``learners`` (default: every rank = all-gather) restricts the DESTINATIONS: with ``learners=[0]`` the slabs are
    gathered to rank 0 only (the shape of ``dist.gather``): 1/world of the fabric traffic, and only the learner ingests
    (world-1) slabs per gather.  Non-learner ranks get ``None`` back from ``gather()``.

    Consumer-release protocol (peers used to write into ``full[i]`` with nothing telling them the local
    consumer had finished with it).  Buffer i is rewritten by gather k + depth.  Before its copies, that gather
        1. waits for the LOCAL release of buffer i - an event recorded by ``release()``, or implicitly by the
           ``next_local()`` call of this round on the compute streams (contract: the tensor returned by gather k may be
           read by work enqueued on the compute streams BEFORE next_local() of round k + depth is called), then
        2. runs a 4-byte all-reduce on the side stream (the RELEASE barrier): once it completes on a rank, every rank has
           passed step 1, i.e. nobody is still reading its buffer i - only then are the peer copies issued, followed by
        3. the COMPLETION barrier (second 4-byte all-reduce): every rank's copies into every buffer i have landed.
    With the ``gloo`` control plane of the single-GPU test both barriers are stream-sync + host barrier.

    Set-up (once): each rank exports the IPC handle of its ``full`` buffers (``torch`` shares CUDA/HIP storages with
    ``_share_cuda_``; the dmabuf IPC mode needs ``HSA_ENABLE_IPC_MODE_LEGACY=0``, already exported on these boxes) through
    ``all_gather_object`` on the process group, and opens its peers'."""

    def __init__(self, slab_shape, dtype=torch.float32, device="cuda", group: Optional[dist.ProcessGroup] = None,
                 depth: int = 2, learners=None):
        super().__init__(slab_shape, dtype=dtype, device=device, group=group, depth=depth)
        assert self._cuda, "peer copies need device memory"
        self.learners = sorted(set(range(self.world) if learners is None else learners))
        assert self.learners and all(0 <= r < self.world for r in self.learners)
        self.is_learner = self.rank in self.learners
        self._peers = [[None] * self.world for _ in range(depth)]     # [buffer][rank] -> that rank's `full` tensor
        if self.world > 1:
            from torch.multiprocessing.reductions import rebuild_cuda_tensor, reduce_tensor
            for i in range(depth):
                fn, args = reduce_tensor(self.full[i])                # IPC handle + metadata (picklable)
                assert fn is rebuild_cuda_tensor
                handles = [None] * self.world
                dist.all_gather_object(handles, args, group=self.group)
                for r in self.learners:
                    if r == self.rank:
                        self._peers[i][r] = self.full[i]
                    else:
                        # storage_device in the handle is the EXPORTING rank's device index: every process of the node sees
                        # all GPUs under the same numbering (one process per GPU, no *_VISIBLE_DEVICES masking)
                        self._peers[i][r] = rebuild_cuda_tensor(*handles[r])   # maps the peer's allocation into this process
        else:
            for i in range(depth):
                self._peers[i][0] = self.full[i]
        self._flag = torch.zeros((1,), dtype=torch.float32, device=self.device)
        self._device_barrier = self.world > 1 and dist.get_backend(self.group) == "nccl"
        self._released = [None] * depth                               # event: the local consumer is done with full[i]

    def _barrier(self) -> None:
        """on the side stream (current): every rank has reached this point of ITS side stream"""
        if self.world <= 1:
            return
        if self._device_barrier:
            dist.all_reduce(self._flag, group=self.group)
        else:
            self.stream.synchronize()
            dist.barrier(group=self.group)

    def release(self, streams=None) -> None:
        """The local consumer has enqueued its last read of the most recently gathered buffer on `streams` (default: the
        current stream): peers may overwrite it once every rank has said so (see the class docstring)."""
        if self._k == 0:
            return
        i = (self._k - 1) % self.depth
        evs = []
        for st in self._producers(streams):
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        self._released[i] = evs

    def next_local(self, streams=None) -> torch.Tensor:
        i = self._k % self.depth
        if self._k >= self.depth and self._released[i] is None:       # implicit release of full[i] (see the contract)
            evs = []
            for st in self._producers(streams):
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            self._released[i] = evs
        return super().next_local(streams)

    def gather(self, streams=None):
        i = self._k % self.depth
        first_use = self._k < self.depth
        self._k += 1
        n = self.slab_shape[0]
        for st in self._producers(streams):
            ev = torch.cuda.Event()
            ev.record(st)
            self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            if not first_use:
                for ev in (self._released[i] or []):                  # 1. the local consumer is done with full[i]
                    self.stream.wait_event(ev)
                self._released[i] = None
                self._barrier()                                       # 2. ... and so is everybody else
            for r in self.learners:                                   # one copy per destination link (and the local one)
                self._peers[i][r][self.rank * n:(self.rank + 1) * n].copy_(self.local[i], non_blocking=True)
            self._barrier()                                           # 3. every rank's copies have landed
            self.ready[i].record(self.stream)
        return self.full[i] if self.is_learner else None
