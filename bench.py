#!/usr/bin/env python
"""bench.py — audio env-steps/s of the SoundSpaces audio-observation hot path on MI355X.

A *step* = one pass of the hot path over one batch of synthetic input: for every env of the batch, RIR (resident
in an HBM bank larger than the Infinity Cache) -> 2-ear FFT convolution with the env's source clip -> truncate to
1 s -> STFT(512/160/400) -> |.| -> 4x4 mean pool -> log1p -> spectrogram [65, T4, 2] in device memory
(reference: soundspaces/simulator.py:608-701 + soundspaces/tasks/nav.py:86-100, cache-miss path).
Workload = BASELINE.json's metric shape: 128 envs, 16 kHz, 1-s clips, 2-channel RIRs of 1 s.

  python bench.py [--gpus N --steps K --warmup W]
      N>1 without a torch.distributed environment: bench.py re-launches itself as N ranks (python -m
      torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1, one rank per GPU over RCCL - the
      reference's own launcher shape, ss_baselines/av_nav/single_node.sh:8-11); under torchrun it checks WORLD_SIZE == N.

`value` (round 6) is the DEPENDENT-step figure: one env group; per step ss_ctx_observe renders into
rollouts.observations['spectrogram'][step + 1], a one-workgroup kernel on the caller's stream consumes those rows (the policy's
stand-in, csrc/ss_bench_token.hip) and the next step is ordered behind it - the loop a trainer runs
(ss_baselines/av_nav/ppo/ppo_trainer.py:133-194).  `pipelined` = the r1-r5 protocol (independent steps on 2-3 internal streams:
needs that many independent env groups), `dependent.two_groups` = a double-buffered sampler, `dependent.host_sync` = the host also
waits for every step.

N>1: every rank renders its own 128 envs (weak scaling, units are independent; --scaling strong splits 128 envs over the
ranks, BASELINE configs[3]), each rank's consumer reads its own slab (DD-PPO) and the per-rank spectrogram slabs are all-gathered
over RCCL on a side stream (the exchange step BASELINE.json names) in chunks of --gather-every steps; the same run also times
`exchange_none` (no collective: the reference's arrangement), `allgather`, `gather` (peer copies to one learner) and the per-step
gather and reports them beside the headline.  Rank 0 prints ONE JSON line.

The line also carries, measured in the same run: per-step GPU time distribution (HIP events per step), the convolution
kernel alone on both bank forms (`roofline_conv_only`, `roofline_conv_only_spectral`: the figure BASELINE's >= 40 % target is
defined on), the other RIR-bank format, the PLUGIN PATH (simulator state -> planning -> descriptor upload -> launch -> rollout
rows, all inside the timed region) and the CPU oracle on the host cores this process may use.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sound-spaces_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def bytes_per_unit(sr, rir_len, t4):
    """SURVEY.md 8(d) algorithmic bytes per env-step."""
    b_rir = 2 * rir_len * 4
    b_spec = 65 * t4 * 2 * 4
    return {"fused": b_rir + b_spec, "conv": b_rir + 2 * sr * 4, "spec": 2 * sr * 4 + b_spec}


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (scipy.signal.fftconvolve x2 + numpy restatement of librosa.stft/block_reduce),
# one process per core like habitat.VectorEnv, timed for a bounded wall-clock budget.  Runs BEFORE CUDA init.
def _cpu_worker(args):
    """One reference-style env process: the oracle's restatement of simulator.py:608-666 + nav.py:86-100 on the CONFIG's own
    shape - `savi`: clips of 1-20 s (every windowing branch), a distractor on every step (two more fftconvolve calls + add);
    `feats`: the extension features of BASELINE configs[4] (textbook log-mel / GCC-PHAT of the oracle) on top."""
    seed, sr, seconds, workload, feats = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import ss_oracle as O
    rng = np.random.default_rng(seed)
    savi = workload == "savi"
    if savi:
        secs = [1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20]
        src = [O.synth_sources(rng, sr, k=1, seconds=sec)[0] for sec in secs]
    else:
        src = O.synth_sources(rng, sr, k=4)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=8)]
    n, t_conv, t_spec, t_feat = 0, 0.0, 0.0, 0.0
    t0 = time.perf_counter()
    t_end = t0 + seconds
    while time.perf_counter() < t_end:
        ta = time.perf_counter()
        if savi:
            k = int(rng.integers(0, len(src)))
            idx = int(rng.integers(0, secs[k]))
            a = O.compute_audiogoal(src[k], rirs[n % 8], sr, audio_index=idx, distractor=src[n % 3],
                                    distractor_rir=rirs[(n + 3) % 8])
        else:
            a = O.compute_audiogoal(src[n % 4], rirs[n % 8], sr)
        tb = time.perf_counter()
        O.compute_spectrogram(a)
        tc = time.perf_counter()
        if "logmel" in feats:
            O.compute_logmel(a, sr)
        if "gccphat" in feats:
            O.compute_gcc_phat(a)
        td = time.perf_counter()
        t_conv += tb - ta
        t_spec += tc - tb
        t_feat += td - tc
        n += 1
    return n, time.perf_counter() - t0, t_conv, t_spec, t_feat


def usable_cores():
    """Cores this process may actually run on: the scheduler affinity mask, capped by the cgroup CPU quota
    (os.cpu_count() reports the machine, not the container: r1's '256 cores' scaled only 20x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(sr, seconds, workload="audiogoal", feats=(), rotations=1):
    cores = usable_cores()
    feats = tuple(feats)
    one = _cpu_worker((0, sr, min(4.0, seconds), workload, feats))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(100 + i, sr, seconds, workload, feats) for i in range(cores)])
    total = sum(r[0] / r[1] for r in res)
    one_rate = one[0] / one[1]
    n_all = sum(r[0] for r in res)
    shape = ("savi: clips of 1-20 s, distractor on every step (4 fftconvolve calls + add)" if workload == "savi"
             else "1-s clip, 1-s RIR")
    stages_one = {"fftconvolve_x2": round(1e3 * one[2] / one[0], 3), "spectrogram": round(1e3 * one[3] / one[0], 3)}
    stages_all = {"fftconvolve_x2": round(1e3 * sum(r[2] for r in res) / n_all, 3),
                  "spectrogram": round(1e3 * sum(r[3] for r in res) / n_all, 3)}
    if workload == "savi":
        stages_one["fftconvolve_x4"] = stages_one.pop("fftconvolve_x2")
        stages_all["fftconvolve_x4"] = stages_all.pop("fftconvolve_x2")
    if feats:
        stages_one["features"] = round(1e3 * one[4] / one[0], 3)
        stages_all["features"] = round(1e3 * sum(r[4] for r in res) / n_all, 3)
    return {"value": round(total, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "one_core": round(one_rate, 1), "scaling_efficiency": round(total / (one_rate * cores), 3),
            "stage_ms_one_core": stages_one, "stage_ms_all_cores": stages_all,
            "sample": f"oracle (scipy fftconvolve + numpy STFT/pool/log1p"
                      f"{' + ' + '/'.join(feats) if feats else ''}), {cores} processes (sched affinity / cgroup "
                      f"quota; os.cpu_count() = {os.cpu_count()}) x {seconds:.0f} s, sr={sr}, {shape}, caches off"
                      f"{'; a unit = one (env, rotation) observation, the ' + str(rotations) + ' rotations of an env are ' + str(rotations) + ' units' if rotations > 1 else ''}"
                      f"; 1 core alone: {one_rate:.1f} env-steps/s"}


# ---------------------------------------------------------------------------------------------------------
def synth_rir_bank_device(torch, n, sr, length, device, seed):
    """SURVEY 8(d) synthetic RIR bank generated on the GPU: decaying Gaussian noise + direct-path impulse,
    peak-normalised to 0.5; float32 planar [n, 2, length]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = torch.arange(length, device=device, dtype=torch.float32)
    out = torch.empty((n, 2, length), dtype=torch.float32, device=device)
    chunk = 256
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        rt60 = torch.empty((m, 1, 1), device=device).uniform_(0.2, 0.8, generator=g)
        gain = torch.empty((m, 2, 1), device=device).uniform_(0.5, 1.0, generator=g)
        delay = torch.randint(0, int(0.0107 * sr), (m, 2, 1), device=device, generator=g)
        h = torch.randn((m, 2, length), device=device, generator=g) * torch.exp(-6.9 * k / (rt60 * sr)) * 0.1 * gain
        h = torch.where(k.view(1, 1, -1) < delay, torch.zeros_like(h), h)
        h.scatter_add_(2, delay, gain)
        h *= 0.5 / h.abs().amax(dim=(1, 2), keepdim=True)
        out[lo:lo + m] = h
    return out


# ---------------------------------------------------------------------------------------------------------
# Plugin path: what a vector env pays per step when it goes through the Habitat-side boundary instead of handing the
# kernels pre-planned descriptors -- simulator state -> units -> planning -> descriptor upload -> launch -> rollout
# rows, ALL inside the timed region (the headline above pre-plans its descriptors).
class SyntheticSim:
    """Stand-in for SoundSpacesSim with the attributes its audio code reads (soundspaces/simulator.py:110-117,
    303-305) and a `move()` that changes them the way `step()` does (:500-516: rotate by +-90 or hop to a neighbour)."""

    def __init__(self, sounds, n_nodes, rng):
        self._source_sound_dict = sounds
        self._current_sound = "sound%d" % rng.integers(0, len(sounds))
        self._current_distractor_sound = None
        self._episode_step_count, self._duration = 0, 500
        self._receiver_position_index = int(rng.integers(0, n_nodes))
        self._source_position_index = int(rng.integers(0, n_nodes))
        self._distractor_position_index = 0
        self._rotation_angle = int(rng.integers(0, 4)) * 90
        self._audio_index = 0

    def move(self, action, node):
        if action == 0:
            self._receiver_position_index = node
        elif action == 1:
            self._rotation_angle = (self._rotation_angle + 90) % 360
        else:
            self._rotation_angle = (self._rotation_angle - 90) % 360
        self._episode_step_count += 1


def measure_plugin_path(torch, np, dev, sr, n_envs, bank, n_sounds, sources, steps, warmup, spectra=None):
    """-> dict for the JSON line.  One 'scene' of n_nodes x n_nodes (receiver, source) pairs whose 4 azimuths sit in 4
    adjacent rows of `bank` (the same HBM-resident bank as the headline); n_envs stand-in simulators bound to a
    VectorSimState; per step: agents move (attribute writes on the simulators), FastVectorAudioObserver.observe_into()
    renders all envs into the rollout rows of the next insert()."""
    import types
    from ss_amd.context import AudioContext
    from ss_amd.rollout import RolloutStorage
    from ss_amd.vector import FastVectorAudioObserver, RirIndex, VectorSimState
    from ss_amd import planning as P
    R = bank.shape[0]
    n_nodes = int(np.sqrt(R // 4))
    rng = np.random.default_rng(5)
    sounds = {"sound%d" % i: c for i, c in enumerate(sources)}
    ctx = AudioContext(sr, max_window_sets=max(256, 2 * n_sounds))
    ctx.set_rir_bank(bank, torch.full((R,), bank.shape[2], dtype=torch.int32, device=dev))
    if spectra is not None:
        ctx.set_rir_spectra(spectra)                      # same bank format as the headline loop
    index = RirIndex(4)
    sid = index.add_scene("synthetic", n_nodes)
    rr, ss = np.meshgrid(np.arange(n_nodes), np.arange(n_nodes), indexing="ij")
    index.set(sid, rr.reshape(-1), ss.reshape(-1), (4 * (rr * n_nodes + ss)).reshape(-1).astype(np.int32))
    sims = [SyntheticSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    state = VectorSimState(n_envs)
    for i, sim in enumerate(sims):
        state.bind(sim, i)
    state.scene[:] = sid
    obs = FastVectorAudioObserver(ctx, state, index, sr)
    T = 16
    space = types.SimpleNamespace(spaces={"spectrogram": types.SimpleNamespace(shape=P.spectrogram_shape(sr))})

    class ActionSpace:                      # RolloutStorage tests the class name, like the reference
        pass
    rollouts = RolloutStorage(T, n_envs, space, ActionSpace(), 8, device=dev)
    total = warmup + steps
    acts = rng.integers(0, 3, (total, n_envs))
    nodes = rng.integers(0, n_nodes, (total, n_envs))

    def run(k0, k1, move_sims, host_us=None):
        for k in range(k0, k1):
            if move_sims:                   # the simulators' own work: attribute writes land in the columns
                a, nd = acts[k], nodes[k]
                for i, sim in enumerate(sims):
                    sim.move(a[i], int(nd[i]))
            else:                           # the same motion applied to the columns directly (vectorised env)
                a = acts[k]
                state.recv[:] = np.where(a == 0, nodes[k], state.recv)
                state.rot[:] = np.where(a == 1, (state.rot + 90) % 360, np.where(a == 2, (state.rot - 90) % 360, state.rot))
                state.step_count += 1
            t0 = time.perf_counter()
            obs.observe_into(rollouts)
            rollouts.step = (rollouts.step + 1) % T      # the other fields of insert() are the trainer's business
            if host_us is not None:
                host_us.append(1e6 * (time.perf_counter() - t0))

    out = {}
    for name, move_sims, native in (("columns", False, True), ("columns_numpy_host", False, False), ("bound_sims", True, True)):
        obs.native = native
        run(0, warmup, move_sims)
        torch.cuda.synchronize()
        host_us = []
        t0 = time.perf_counter()
        run(warmup, total, move_sims, host_us)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        h = np.sort(np.asarray(host_us))
        out[name] = {"env_steps_per_s": round(n_envs * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 5),
                     "observe_into_host_us": {"median": round(float(np.median(h)), 2),
                                              "p10": round(float(h[len(h) // 10]), 2),
                                              "p90": round(float(h[(9 * len(h)) // 10]), 2)}}
    out["note"] = ("simulator state columns -> ss_ctx_observe_sims (C++: silence / clip window / azimuth / RIR table lookup, "
                   "planner + window cache + pinned descriptor ring, launch) -> spectrograms written into "
                   "rollouts.observations['spectrogram'][step+1]; planning and descriptor upload inside the timed region.  "
                   "'columns': agent motion applied to the state columns (vectorised env); 'columns_numpy_host': the same "
                   "with the state -> unit columns step in numpy (ss_ctx_observe); 'bound_sims': motion applied through attribute writes on %d Python simulator objects "
                   "bound to the columns (their Python loop is the simulators' cost, not the audio path's)" % n_envs)
    out["cache"] = ctx.stats()
    return out


def measure_deferred_path(torch, np, dev, sr, n_envs, bank, sources, steps, warmup):
    """The reference's DEFAULT env arrangement (habitat.VectorEnv: one worker process per env,
    ss_baselines/common/env_utils.py:91-107) through ss_amd.deferred: every env's sensor returns an AudioRequest (worker
    half: runs inside the env processes, N-way parallel in real use), the trainer turns the N requests of the step into unit
    columns and ONE ss_ctx_observe into the rollout rows (trainer half: what the learner process pays).  RIR 'files' are rows
    of the headline's HBM bank served through a reader on first touch; steady-state steps find every pose resident."""
    import types
    from ss_amd import planning as P
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.renderer import AudioEngine
    from ss_amd.rollout import RolloutStorage
    R = bank.shape[0]
    n_nodes = min(int(np.sqrt(R // 4)), 24)                   # 24 x 24 pairs x 4 azimuths = 2304 poses: 288 MiB of store
    rng = np.random.default_rng(6)
    sounds = {"sound%d" % i: c for i, c in enumerate(sources)}
    NS = types.SimpleNamespace

    class DSim(SyntheticSim):
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = "bank"
        azimuth_angle = property(lambda self: -(self._rotation_angle + 0) % 360)
        current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])
        _audio_length = property(lambda self: self.current_source_sound.shape[0] // sr)

    def reader(path):                                         # "bank/<azimuth>/<recv>_<src>.wav" -> a row of the device bank
        _, az, name = path.split("/")
        r_, s_ = name[:-4].split("_")
        row = 4 * (int(r_) * n_nodes + int(s_)) + int(az) // 90
        return bank[row].t().contiguous().cpu().numpy()
    eng = AudioEngine(sr, device=dev, rir_slots=4 * n_nodes * n_nodes)
    res = DeferredResolver(eng, rir_reader=reader, fast=True)
    sims = [DSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    for s_ in sims:
        s_._duration = 10 ** 9
    for i, sim in enumerate(sims):
        attach_deferred(sim, env_rank=i)
    T = 16
    space = NS(spaces={"spectrogram": NS(shape=P.spectrogram_shape(sr))})

    class ActionSpace:
        pass
    rollouts = RolloutStorage(T, n_envs, space, ActionSpace(), 8, device=dev)
    total = warmup + steps
    acts = rng.integers(0, 3, (total, n_envs))
    nodes = rng.integers(0, n_nodes, (total, n_envs))
    # scene load: every pose once (wav read + H2D per pose in real use), so that the timed steps are steady-state ones
    poses = [(r_, s_, az) for r_ in range(n_nodes) for s_ in range(n_nodes) for az in range(4)]
    for lo in range(0, len(poses), n_envs):
        part = poses[lo:lo + n_envs]
        for sim, (r_, s_, az) in zip(sims, part):
            sim._receiver_position_index, sim._source_position_index, sim._rotation_angle = r_, s_, 90 * az
            sim._episode_step_count += 1                  # (a new simulator state: a new request)
        res.resolve([sim.get_current_spectrogram_observation(None) for sim in sims[:len(part)]])
    torch.cuda.synchronize()
    out = {}
    for name, replace in (("trainer_rollout_rows", False), ("trainer_rollout_rows_and_per_env_views", True)):
        w_us, t_us = [], []
        for k in range(total):
            if k == warmup:
                torch.cuda.synchronize()
                t_start = time.perf_counter()
            a, nd = acts[k], nodes[k]
            for i, sim in enumerate(sims):
                sim.move(a[i], int(nd[i]))
            t0 = time.perf_counter()
            observations = [{"spectrogram": sim.get_current_spectrogram_observation(None)} for sim in sims]
            t1 = time.perf_counter()
            res.resolve_observations(observations, rollouts, replace=replace)
            rollouts.step = (rollouts.step + 1) % T
            t2 = time.perf_counter()
            if k >= warmup:
                w_us.append(1e6 * (t1 - t0)); t_us.append(1e6 * (t2 - t1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_start
        tm = float(np.median(t_us))
        out[name] = {"trainer_half_us_per_step": round(tm, 2), "trainer_half_env_steps_per_s": round(n_envs / (tm * 1e-6), 1),
                     "worker_half_us_per_step_all_envs": round(float(np.median(w_us)), 1),
                     "both_halves_serial_env_steps_per_s": round(n_envs * steps / dt, 1)}
    out["resident_pairs"] = int(res._pair_keys.shape[0])
    out["column_steps"], out["walk_steps"] = res.column_steps, res.walk_steps
    # batched in-process mode over the SAME engine (the reference's SyncVectorEnv arrangement, simulators untouched):
    # VectorAudioObserver reads the simulators with attrgetter, packs the step into the same records and makes the same C call
    from ss_amd import sim_audio
    sims_b = [DSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    for s_ in sims_b:
        s_._duration = 10 ** 9
    vobs = sim_audio.VectorAudioObserver(eng, [sim_audio.attach(s_, eng, rir_reader=reader) for s_ in sims_b])
    for lo in range(0, len(poses), n_envs):                   # scene load for THIS observer's pair table (rows are resident)
        for sim, (r_, s_, az) in zip(sims_b, poses[lo:lo + n_envs]):
            sim._receiver_position_index, sim._source_position_index, sim._rotation_angle = r_, s_, 90 * az
        vobs.observe()
    torch.cuda.synchronize()
    h_us = []
    for k in range(total):
        if k == warmup:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        a, nd = acts[k], nodes[k]
        for i, sim in enumerate(sims_b):
            sim.move(a[i], int(nd[i]))
        t0 = time.perf_counter()
        vobs.observe_into(rollouts)
        rollouts.step = (rollouts.step + 1) % T
        if k >= warmup:
            h_us.append(1e6 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    hm = float(np.median(h_us))
    out["batched_in_process"] = {"observe_into_host_us": round(hm, 2), "env_steps_per_s_with_sim_moves": round(n_envs * steps / dt, 1),
                                 "record_steps": vobs.record_steps, "walk_steps": vobs.walk_steps}
    out["note"] = ("AudioRequest per env (packed record: CRC keys of sound / RIR directory, receiver, source, clip window) -> "
                   "DeferredResolver._columns (numpy: searchsorted over resident pairs) -> AudioEngine.observe_columns -> "
                   "ss_ctx_observe (C++ planner, window cache, descriptor ring, launch) into rollouts.observations['spectrogram']"
                   "[step+1].  The worker half runs in the env processes (here: serially, in this process); "
                   "'..._and_per_env_views' also replaces the request in each env's observation dict by its row view.")
    return out


def dist_of(ms):
    """median / p10 / p90 of per-step durations (ms)."""
    import numpy as np
    v = np.sort(np.asarray(ms, dtype=np.float64))
    return {"median": round(float(np.median(v)), 5), "p10": round(float(v[len(v) // 10]), 5),
            "p90": round(float(v[(9 * len(v)) // 10]), 5)}


def measured_traffic(units, sr, kernel):
    """HBM bytes per launch from the committed PMC pass of THIS build (profiles/r6/traffic.json written by
    scripts/gpu_profile_r6.sh with the hash of the kernel sources): null when the sources have changed since.  Both
    counters carry the factor measured in the same pass on a known byte count in the kernel's dominant access pattern
    (scripts/calib_traffic.hip)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r6", "traffic.json")))
        have = open(os.path.join(ROOT, "sound-spaces_amd", "csrc", ".libss_hip.kernelhash")).read().strip()
        e = tj["kernels"].get("%s@%d" % (kernel, units)) or tj["kernels"][kernel]
        if tj["source_hash"] == have and e["units_per_launch"] == units and e["sampling_rate"] == sr:
            fc, wc = float(e.get("fetch_correction", 1.0)), float(e.get("write_correction", 1.0))
            return {"bytes": int(fc * e["fetch_bytes"] + wc * e["write_bytes"]), "fetch_bytes_raw": int(e["fetch_bytes"]),
                    "fetch_correction": fc, "write_bytes_raw": int(e["write_bytes"]), "write_correction": wc,
                    "calibrated_on": {"fetch": e.get("fetch_pattern"), "write": e.get("write_pattern")},
                    "tcc_hit_rate": e.get("tcc_hit_rate"),
                    "traffic_source": "profiles/r6 rocprofv3 --pmc pass of this build (source hash matches), not measured in this run",
                    "note": e.get("note", "")}
    except Exception:
        pass
    return None


def dry_run(args, rank, world):
    """The multi-rank flow of this script without a GPU: ranks, barrier + max-over-ranks timing, the chunked slab exchange
    (ss_amd.dist over gloo, CPU tensors standing in for the renderer's output) and the one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from ss_amd import planning as P
    from ss_amd.dist import ChunkedSlabExchange
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_env = args.envs // world if args.scaling == "strong" else args.envs
    N = n_env * args.rotations
    shape = P.spectrogram_shape(args.sr)
    seen = []
    cx = ChunkedSlabExchange(N, shape, args.gather_every, device="cpu",
                             gathered=lambda full, n: seen.append((tuple(full.shape), n, float(full[(world - 1) * args.gather_every * N, 0, 0, 0]))))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        cx.step_rows().fill_(float(rank))          # stands in for the kernels' output rows
        cx.step_done()
    cx.flush()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = all(sh == (world * args.gather_every * N,) + shape and last == float(world - 1) for sh, _, last in seen)
    if rank == 0:
        print(json.dumps({"metric": "audio env-steps/sec (RIR-convolve+spectrogram) per node, 128 envs Replica 16 kHz",
                          "dry_run": True, "value": None, "unit": "env-steps/s", "n_gpus": world, "rccl_ranks": 0,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 5),
                          "scaling": args.scaling, "gathers": cx.gathers, "gather_ok": bool(ok),
                          "config": {"workload": "dry run: no kernels", "envs_per_gpu": n_env, "units_per_gpu": N}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-steps", type=int, default=-1,
                    help="untimed device spin-up before the warm-up steps (GPU clocks take milliseconds to come up; a "
                         "20-step timed region lasts half a millisecond); -1 = 1500 steps at <= 128 units per GPU, fewer for "
                         "larger launches; 0 disables")
    ap.add_argument("--envs", type=int, default=128, help="envs per GPU per step (weak scaling) / in total (--scaling strong)")
    ap.add_argument("--rotations", type=int, default=1, help="agent rotations rendered per env and step (BASELINE configs[2]: "
                                                             "4; the azimuths of a pair sit in adjacent bank rows)")
    ap.add_argument("--sr", type=int, default=16000)
    ap.add_argument("--bank-mib", type=int, default=1024, help="time-domain RIR bank size per GPU (4x the 256 MiB Infinity "
                                                              "Cache: the rows of the timed region are first touches)")
    ap.add_argument("--sounds", type=int, default=102)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --envs per GPU (default); strong: --envs in total, split over the ranks (BASELINE configs[3]: "
                         "128 envs = 16 per GPU on 8 GPUs)")
    ap.add_argument("--exchange", choices=["allgather", "peercopy", "gather", "none"], default="allgather",
                    help="allgather: RCCL all_gather_into_tensor on a side stream; peercopy: every rank copies its slab into the "
                         "peers' buffers over xGMI (HIP IPC mappings, SDMA / blit path: no CUs taken from the kernels), RCCL "
                         "only for the 4-byte release / completion barriers; gather: peer copies into the LEARNER rank (0) only "
                         "(1/world of the fabric traffic); none: no exchange (DD-PPO: each rank consumes its own slab)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="steps per all-gather: the learner consumes rollouts, so per-rank slabs are exchanged in chunks of "
                         "this many steps (fewer, larger collectives suit the point-to-point xGMI fabric); the per-step "
                         "gather (1) is timed as well and reported beside it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test "
                                                      "the multi-rank flow on a 1-GPU box)")
    ap.add_argument("--streams", type=int, default=0,
                    help="internal streams (lanes) of the product path's overlap mode: 1 .. 4; 0 = auto: 2, or 3 for steps "
                         "that fill at most half the chip (<= 128 (unit, ear) rows)")
    ap.add_argument("--regions", type=int, default=0,
                    help="timed regions of --steps steps each for the headline pass (value = the median region); 0 = auto: 25 "
                         "when --steps <= 50 (a 20-step region is 0.5 ms), else 1")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="seconds of the headline loop run once more, untimed for `value`, after the timed regions (reported as "
                         "`sustained`: long enough for once-a-second utilisation samplers to see the load); 0 = off")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plugin-path", action="store_true", help="skip the plugin-path (boundary) measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the conv-only / spectral-bank / 2-stream side measurements")
    ap.add_argument("--rir-bank", choices=["auto", "spectral", "time"], default="auto",
                    help="format of the HBM-resident RIR bank the headline loop reads: 'time' = the reference's time-domain "
                         "samples (the format SURVEY 8(d) defines the metric and its algorithmic bytes on); 'spectral' = block "
                         "spectra computed once at bank load (ss_rir_spectra_f32), no forward FFT per step, 2x the bytes per "
                         "RIR.  The other format is timed in the same run and reported beside it.  'auto' (default) = what "
                         "AudioEngine(rir_spectral=None) does: both forms resident (when they fit its HBM share), every launch "
                         "reads the spectral rows - measured faster at every step size and rate (profiles/r6/"
                         "kbench_bank_form_16k.txt).  SURVEY 8(d) defines the metric's ALGORITHMIC bytes on the time-domain rows: "
                         "`roofline` keeps that definition whatever the launch reads (`actual_bytes_per_unit` says what it reads)")
    ap.add_argument("--config", choices=["headline", "cfg1", "cfg2", "cfg3", "cfg4"], default="headline",
                    help="BASELINE.json configs[] presets: cfg1 = 32 envs @16 kHz; cfg2 = 128 envs x 4 rotations @44.1 kHz "
                         "(512 units / launch); cfg4 = savi: 256 envs, 21 sounds of 1-20 s, distractor, audiogoal + "
                         "spectrogram + the fused log-mel / GCC-PHAT sensor; cfg3 = av_nav DD-PPO: 8 x 16 envs @16 kHz - with "
                         "--gpus 8 the node's run (16 envs per rank, all-gather of the spectrograms), with --gpus 1 (default) ONE "
                         "rank's shard of it: 16 envs per step on one GPU.  A preset sets --envs/--sr/--rotations/--workload "
                         "(cfg3 also --scaling)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no kernels: the multi-rank launch / exchange / JSON flow on CPU tensors over gloo (tests only; "
                         "the line says dry_run and its value is not a measurement)")
    ap.add_argument("--with-audiogoal", action="store_true", help="also materialise the [N,2,sr] waveform")
    ap.add_argument("--features", default=None,
                    help="comma list of logmel,gccphat: the STFT-derived extension features of BASELINE configs[4] (savi: "
                         "'GCC-PHAT + log-mel fused sensor'), produced INSIDE the timed region by one k_features launch per step "
                         "on the step's stream (ss_ctx_observe_features); default: both for --workload savi, none otherwise; "
                         "'none' switches them off")
    ap.add_argument("--workload", choices=["audiogoal", "savi"], default="audiogoal",
                    help="audiogoal: the headline shape (1-s clips, no distractor).  savi: BASELINE configs[4] (semantic_audionav): "
                         "21 sounds of 1-20 s (all windowing branches of simulator.py:629-647), a distractor on every env "
                         "(two convolutions + add), audiogoal AND spectrogram written; use with --envs 256")
    args = ap.parse_args()
    if args.config == "cfg1":
        args.envs, args.sr, args.rotations = 32, 16000, 1
    elif args.config == "cfg2":
        args.envs, args.sr, args.rotations = 128, 44100, 4
    elif args.config == "cfg3":
        # BASELINE configs[3]: 8 x 16 envs sharded over 8 GPUs.  --gpus 8: 128 envs split over the ranks; fewer GPUs: 16 envs
        # per rank (the per-GPU shard of the node's run - what one GPU of the node does per step)
        args.sr, args.rotations = 16000, 1
        if args.gpus == 8:
            args.envs, args.scaling = 128, "strong"
        else:
            args.envs, args.scaling = 16, "weak"
    elif args.config == "cfg4":
        args.envs, args.sr, args.rotations, args.workload = 256, 16000, 1, "savi"

    # (after the presets: 'auto' needs the step's size)
    args.spectral = args.rir_bank in ("spectral", "auto")        # auto = AudioEngine(rir_spectral=None): both forms resident, the
                                                              # launches read the spectral rows (faster at every size: kbench_bank_form_16k.txt)
    feats = args.features if args.features is not None else ("logmel,gccphat" if args.workload == "savi" else "none")
    feats = [f for f in feats.split(",") if f and f != "none"]
    assert all(f in ("logmel", "gccphat") for f in feats), "--features: logmel, gccphat"
    # --gpus N is a request, not a label: without a torch.distributed environment this process becomes the launcher of N
    # ranks (one per GPU; ss_baselines/av_nav/single_node.sh:8-11 does the same with torch.distributed.launch)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}"
    if args.dry_run:
        return dry_run(args, rank, world)
    sr = args.sr
    if args.scaling == "strong":
        assert args.envs % world == 0, "--scaling strong: --envs must divide by the number of ranks"
        n_env = args.envs // world
    else:
        n_env = args.envs
    N = n_env * args.rotations                               # units per GPU per step

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:                # rank 0 of ANY world size: an N > 1 line carries it too (the
        cpu = cpu_baseline(sr, args.cpu_seconds, args.workload, feats, args.rotations)   # other ranks wait at the first barrier);
                                                          # before any CUDA context exists (fork-safe); the CONFIG's own shape

    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import ss_oracle as O
    from ss_amd import planning as P
    from ss_amd.dist import ChunkedSlabExchange, PeerCopyExchange
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    from ss_amd import ops as ops_mod

    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no CPU fallback"
    import ctypes
    bench_so = os.path.join(ROOT, "sound-spaces_amd", "csrc", "libss_bench.so")
    assert os.path.exists(bench_so), "libss_bench.so missing: run `python sound-spaces_amd/build.py` (or __graft_entry__.build())"
    bench_lib = ctypes.CDLL(bench_so)
    bench_lib.ssb_policy_token.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    bench_lib.ssb_policy_token.restype = ctypes.c_int
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    # ---- synthetic, HBM-resident inputs --------------------------------------------------------------
    rng = np.random.default_rng(1000 + rank)
    L = sr                                                   # RIR length = fallback shape (simulator.py:621)
    R = max(64, (args.bank_mib << 20) // (2 * L * 4))
    R -= R % 4                                               # whole azimuth groups
    r = BatchedAudioRenderer(sr, device=dev)
    savi = args.workload == "savi"
    if savi:
        args.sounds = 21
        secs = [1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20]
        for i, sec in enumerate(secs):
            r.add_source(f"sound{i}", O.synth_sources(rng, sr, k=1, seconds=sec)[0])
        args.with_audiogoal = True
    else:
        for i, clip in enumerate(O.synth_sources(rng, sr, k=args.sounds)):
            r.add_source(f"sound{i}", clip)
    bank = synth_rir_bank_device(torch, R, sr, L, dev, seed=7 + rank)
    r.set_rir_bank(RirBank(bank, torch.full((R,), L, dtype=torch.int32, device=dev)))
    # Short timed regions (the driver's --steps 20 is half a millisecond of GPU time, with ~50 us of launch / sync latency at
    # its two ends: twelve runs of one build spread +-3.5 %): the headline times REGIONS regions of EXACTLY --steps steps
    # each - every region bracketed by barrier + synchronize like the one region of the contract, fresh unit columns per
    # region - and reports the MEDIAN region (value_spread: the others).  --steps > 50: one region, as before.
    REGIONS = args.regions if args.regions > 0 else (25 if args.steps <= 50 else 1)
    total = args.warmup + REGIONS * args.steps
    rot = args.rotations

    # ---- the steps: unit columns {sound, t0, rir, (distractor)} per step, drawn once; the SAME columns feed the product
    # path (ss_ctx_observe: planning inside the timed region) and the pre-planned descriptors of the kernel-rate passes
    from ss_amd.context import AudioContext
    from ss_amd.renderer import UnitRequest

    def draw_cols():
        if savi:
            snd = rng.integers(0, 21, n_env)
            idx = np.array([rng.integers(0, secs[s]) for s in snd])
            return dict(sound=snd.astype(np.int32), t0=np.array([0 if secs[s] == 1 else int(i) * sr for s, i in zip(snd, idx)], np.int32),
                        rir=rng.integers(0, R, n_env).astype(np.int32), dis_sound=rng.integers(0, 3, n_env).astype(np.int32),
                        dis_rir=rng.integers(0, R, n_env).astype(np.int32))
        snd = rng.integers(0, args.sounds, n_env)
        base = rng.integers(0, R // rot, n_env) * rot          # rotations > 1: first row of an azimuth group of `rot` rows
        return dict(sound=np.repeat(snd, rot).astype(np.int32), t0=np.zeros(N, np.int32),
                    rir=(np.repeat(base, rot) + np.tile(np.arange(rot), n_env)).astype(np.int32))

    def plan_of(c):
        if savi:
            return r.plan([UnitRequest(int(s), int(t), int(h), dis_sound=int(d), dis_rir=int(hd))
                           for s, t, h, d, hd in zip(c["sound"], c["t0"], c["rir"], c["dis_sound"], c["dis_rir"])])
        return r.plan_arrays(c["sound"][::rot], np.zeros(n_env, np.int64), c["rir"][::rot], rotations=rot)
    cols = [draw_cols() for _ in range(total)]
    descs = [plan_of(c) for c in cols]
    # the untimed spin-up has steps OF ITS OWN: cycling over the timed region's steps would re-read the timed region's bank
    # rows dozens of times before the clock starts ("8(d)'s bank-larger-than-the-Infinity-Cache guarantee is void")
    spin_cols = [draw_cols() for _ in range(16)]
    spin_descs = [plan_of(c) for c in spin_cols]
    t4 = r.spectrogram_shape[1]
    fused = sr <= 3 * P.KB                                   # one launch per step (16 kHz: k_conv; 44.1 kHz: k_obs_rows)
    want_ag = args.with_audiogoal or not fused
    lengths_dev = torch.full((R,), L, dtype=torch.int32, device=dev)
    ctx = AudioContext(sr, max_window_sets=max(256, 4 * args.sounds))
    for i in range(len(r.sources)):
        ctx.add_source(f"sound{i}", r.sources._host[i])
    ctx.set_rir_bank(bank, lengths_dev)
    preps, spin_preps = [ctx.prepare(**c) for c in cols], [ctx.prepare(**c) for c in spin_cols]
    feat_sets = None
    if feats:
        assert fused and sr <= P.KB, "--features: rows of one block (16 kHz)"
        args.with_audiogoal = want_ag = True                   # the features read the step's waveform
        n_mels = 64
        ms_np, mw_np, _ = P.mel_filterbank_sparse(sr, n_mels)
        mel_start, mel_w = torch.from_numpy(ms_np).to(dev), torch.from_numpy(mw_np).to(dev)
        T_fr = 1 + sr // 160
        feat_sets = []
        for _ in range(4):                                     # one output set per lane of the overlap mode
            lm = torch.empty((N, n_mels, T_fr, 2), dtype=torch.float32, device=dev) if "logmel" in feats else None
            gc = torch.empty((N, 65, T_fr), dtype=torch.float32, device=dev) if "gccphat" in feats else None
            feat_sets.append(ctx.features(lm, mel_start, mel_w, 1e-6, gc, 32, 1e-8))

    spin_steps = args.spinup_steps if args.spinup_steps >= 0 else max(64, 1500 * 128 // max(N, 128) * 16000 // sr)

    def run_loop(S, gather_every, spectral, per_step_events=False, lanes=0, regions=1, sustain_s=0.0, dependent=0,
                 host_sync=False, exchange=None):
        """warm-up + EXACTLY args.steps timed steps bracketed by barrier + synchronize -> (elapsed s, GPU ms per step from
        HIP events on the launch stream: the region average, or the per-step list with per_step_events; note).
        lanes = 0: pre-planned descriptors through the stateless entry points on S torch streams (kernel-rate passes; with
        S = 1 the event average IS the kernel's average launch duration, the figure the rocprofv3 kernel trace reports).
        lanes >= 1: the product path, ss_ctx_observe on the step's unit columns (planning, window cache, descriptor ring in
        the timed region); lanes >= 2: its overlap mode (consecutive steps on that many internal streams, joined when a slab is
        gathered and at the end of the region).
        Per-step event records put a marker packet between consecutive launches (measured: +2-3 us per step), so the
        headline loop records only the two ends and the distribution comes from a separate pass.
        dependent = G >= 1: the TRAINER-SHAPED loop.  In the reference step k+1's observation cannot be asked
        for before the policy has consumed step k's (ss_baselines/av_nav/ppo/ppo_trainer.py:133-150: actor_critic.act on
        rollouts.observations[step], envs.step, batch_obs, rollouts.insert).  Here: the env set is G groups of N / G envs; per
        step and group ss_ctx_observe renders into the group's rows of rollouts.observations['spectrogram'][step + 1]
        (a [T+1, N, 65, T4, 2] tensor, T = 16 slots cycled), ss_ctx_join makes the group's caller stream see the rows, and a
        ONE-workgroup kernel on that stream reads them and writes a token (libss_bench.so: the stand-in for the policy).  The
        group's next step is issued on the same caller stream, i.e. ordered behind the token (the library fences a lane behind a
        caller stream that has work pending).  G = 1: one chain, nothing overlaps (one internal stream).  G = 2: the double-buffered
        sampler - group B renders while group A's policy runs: a context and a caller stream PER GROUP (each chain is plain
        in-order work on its own stream: no cross-stream events at all; ss_ctx_set_chip_share(2) makes each group's launch take
        half the chip so the two run side by side).  host_sync: the host also WAITS
        for every token before it issues the next step of that group (a trainer that reads its actions back: actions.item())."""
        r.rirs.spectra = spectra if spectral else None
        if dependent:
            lanes = 1                                              # (two groups: two single-stream contexts, see below)
        ex_mode = args.exchange if exchange is None else exchange
        use_ctx = lanes >= 1
        if use_ctx:
            ctx.set_rir_bank(bank, lengths_dev)                    # (drops the spectral form)
            if spectral:
                ctx.set_rir_spectra(spectra)
            ctx.set_overlap(lanes)
        cx = None
        if world > 1 and gather_every > 0 and ex_mode != "none":
            kw = {"learners": [0]} if ex_mode == "gather" else {}
            cx = ChunkedSlabExchange(N, r.spectrogram_shape, gather_every, device=dev,
                                     exchange_cls=PeerCopyExchange if ex_mode in ("peercopy", "gather") else None, **kw)
        ex = cx.exchange if cx is not None else None
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
        n_out = 16 if lanes == 3 else max(2, S, lanes)             # (3 lanes: lane = ring slot % 3, the ring has 16 slots)
        sg_buf = [torch.empty((N,) + r.spectrogram_shape, dtype=torch.float32, device=dev) for _ in range(n_out)]
        ag_bufs = ([torch.empty((N, 2, sr), dtype=torch.float32, device=dev) for _ in range(n_out)] if want_ag
                   else [None] * n_out)

        main_stream = torch.cuda.current_stream(dev).cuda_stream

        def render(k, plans, columns, rows, ag):
            if use_ctx and feat_sets is not None:
                ctx.observe_prepared_features(columns[k], rows.data_ptr(), ag.data_ptr(), main_stream, feat_sets[k % 4])
            elif use_ctx:                                          # (unit columns converted to the C struct once, above)
                ctx.observe_prepared(columns[k], rows.data_ptr(), None if ag is None else ag.data_ptr(), main_stream)
            elif feat_sets is not None:                            # same two launches as the product path, pre-planned units:
                r.render_audiogoal(plans[k], out=ag)               # convolution (waveform written), then k_features, which
                fk = feat_sets[k % 4]["keep"]                      # also pools the spectrogram from the spectra it holds
                ops_mod.audio_features_into(ag, rows, fk[0], fk[3], fk[1], fk[2])
            else:
                r.render(plans[k], spectrogram_out=rows, audiogoal_out=ag)

        unit_floats = int(np.prod(r.spectrogram_shape))
        tok_dev = torch.zeros((4,), dtype=torch.float32, device=dev)
        tok_host = torch.zeros((4,), dtype=torch.float32).pin_memory() if host_sync else None

        def policy_token(rows_ptr, n_units, g, stream):
            """the stand-in for actor_critic.act on this group's rows, on the group's caller stream"""
            tp = (tok_host.data_ptr() if host_sync else tok_dev.data_ptr()) + 4 * g
            rc = bench_lib.ssb_policy_token(rows_ptr, n_units * unit_floats, tp, stream)
            assert rc == 0, f"ssb_policy_token: hip error {rc}"

        def step(k, plans=descs, columns=preps):
            st = streams[k % S]
            with torch.cuda.stream(st):
                if cx is not None and not state["no_exchange"]:
                    rows = cx.step_rows(streams)
                    render(k, plans, columns, rows, ag_bufs[k % len(ag_bufs)])
                    if dependent:                                  # (one chain per rank: each rank's policy consumes its own slab,
                        policy_token(rows.data_ptr(), N, 0, st.cuda_stream)   #  ddppo_trainer.py:140-142; the gather serves the learner)
                        if host_sync:
                            st.synchronize()
                    if use_ctx and cx.will_gather():
                        ctx.join()                                 # the slab is complete once every lane has drained
                    cx.step_done(streams)                          # all-gather of the chunk once it is full
                else:
                    render(k, plans, columns, sg_buf[k % len(sg_buf)], ag_bufs[k % len(ag_bufs)])

        if dependent and cx is None:
            # rollouts.observations['spectrogram']: [T + 1, N, 65, T4, 2]; step k writes slot 1 + k % T (rollout_storage.py:78-102)
            G, T_roll = dependent, 16
            roll = torch.empty((T_roll + 1, N) + r.spectrogram_shape, dtype=torch.float32, device=dev)
            roll_ag = torch.empty((2, N, 2, sr), dtype=torch.float32, device=dev) if want_ag else None
            cs = [main_stream] if G == 1 else [torch.cuda.Stream(device=dev) for _ in range(G)]
            cs_raw = [c_ if isinstance(c_, int) else c_.cuda_stream for c_ in cs]
            bounds = [(g * N // G, (g + 1) * N // G) for g in range(G)]
            row_b, ag_b = unit_floats * 4, 2 * sr * 4

            ctxs = [ctx]
            if G > 1:
                for _ in range(G - 1):                             # one context per group: same sounds, same bank
                    c2 = AudioContext(sr, max_window_sets=max(256, 4 * args.sounds))
                    for i in range(len(r.sources)):
                        c2.add_source(f"sound{i}", r.sources._host[i])
                    c2.set_rir_bank(bank, lengths_dev)
                    if spectral:
                        c2.set_rir_spectra(spectra)
                    ctxs.append(c2)
                for c_ in ctxs:
                    c_.set_chip_share(G)

            def split(cl):
                return [[ctxs[g].prepare(**{kk: vv[lo:hi] for kk, vv in c.items()}) for g, (lo, hi) in enumerate(bounds)] for c in cl]
            dep_preps = preps if G == 1 else split(cols)
            dep_spin = spin_preps if G == 1 else split(spin_cols)
            assert G == 1 or feat_sets is None, "dependent groups > 1: without the extension features"

            def step(k, plans=descs, columns=preps):               # noqa: F811
                cg = dep_spin if columns is spin_preps else dep_preps
                slot = roll.data_ptr() + (1 + k % T_roll) * N * row_b
                agp0 = None if roll_ag is None else roll_ag.data_ptr() + (k & 1) * N * ag_b
                for g in range(G):
                    lo, hi = bounds[g]
                    sgp = slot + lo * row_b
                    agp = None if agp0 is None else agp0 + lo * ag_b
                    if G == 1:
                        if feat_sets is not None:
                            ctx.observe_prepared_features(cg[k], sgp, agp, cs_raw[0], feat_sets[k % 4])
                        else:
                            ctx.observe_prepared(cg[k], sgp, agp, cs_raw[0])
                    else:
                        ctxs[g].observe_prepared(cg[k][g], sgp, agp, cs_raw[g])
                    policy_token(sgp, hi - lo, g, cs_raw[g])
                    if host_sync:                                  # the trainer reads the group's actions back before envs.step
                        if G == 1:
                            torch.cuda.current_stream(dev).synchronize()
                        else:
                            cs[g].synchronize()
        elif use_ctx and cx is None:                               # the lean loop: one bound call per step, nothing else
            sg_ptrs = [b_.data_ptr() for b_ in sg_buf]
            ag_ptrs = [None if b_ is None else b_.data_ptr() for b_ in ag_bufs]
            n_sg, n_ag = len(sg_ptrs), len(ag_ptrs)

            def step(k, plans=descs, columns=preps):               # noqa: F811
                if feat_sets is not None:
                    ctx.observe_prepared_features(columns[k], sg_ptrs[k % n_sg], ag_ptrs[k % n_ag], main_stream, feat_sets[k % 4])
                else:
                    ctx.observe_prepared(columns[k], sg_ptrs[k % n_sg], ag_ptrs[k % n_ag], main_stream)

        def fence():
            # (torch.cuda.synchronize() is a DEVICE synchronise: it covers the context's internal streams as well; joining
            # them into the launch stream first only puts two cross-stream hops in front of it)
            if world > 1:
                if use_ctx:
                    ctx.join()
                dist.barrier()
            torch.cuda.synchronize()

        def flush():
            if cx is not None and not state["no_exchange"]:
                if use_ctx:
                    ctx.join()
                cx.flush(streams)                                  # partial last chunk + wait

        note = None
        state = {"no_exchange": False}
        if ex is not None:                                         # RCCL communicator / channel set-up (seconds, lazy on
            try:                                                   # the first collective) must not land in the timed region
                for _ in range(2):
                    ex.next_local(streams)
                    ex.gather(streams)
                ex.wait(streams)
                torch.cuda.synchronize()
            except Exception as e:                                 # keep the run alive: the shards do not depend on it
                note = f"all-gather unavailable ({type(e).__name__}: {e}); ran without exchange"
                print("[bench] " + note, file=sys.stderr, flush=True)
                state["no_exchange"] = True
        torch.cuda.synchronize()
        # device spin-up (clocks, TLBs, instruction caches): untimed, reported in the JSON line; the W warm-up steps follow.
        # A fixed NUMBER of steps (every rank issues the same collectives), sized for ~40 ms at the headline shape.
        for k in range(spin_steps):
            step(k % len(spin_descs), spin_descs, spin_preps)
            if k % 64 == 63:
                if use_ctx:
                    ctx.join()
                torch.cuda.synchronize()                           # keep the launch queue shallow
        if spin_steps:
            flush()
            if use_ctx:
                ctx.join()
            torch.cuda.synchronize()
        for k in range(args.warmup):
            step(k)
        flush()
        fence()
        region_s, region_ev = [], []
        timed_events = S == 1 and not (use_ctx and lanes > 1)      # (overlapped lanes: wall clock only - a closing event
        for rg in range(regions):                                  #  would need the lanes joined into this stream)
            k0 = args.warmup + rg * args.steps
            n_ev = args.steps + 1 if per_step_events else 2
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] if timed_events else None
            t_start = time.perf_counter()
            if evs:
                evs[0].record(streams[0])
            for k in range(k0, k0 + args.steps):
                step(k)
                if evs and per_step_events:
                    evs[k - k0 + 1].record(streams[0])
            if use_ctx and lanes == 1:
                ctx.join()
            if evs and not per_step_events:
                evs[1].record(streams[0])
            flush()
            fence()                                                # barrier + device synchronise: also the cut between regions
            region_s.append(time.perf_counter() - t_start)
            if evs:
                region_ev.append([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)] if per_step_events
                                 else [evs[0].elapsed_time(evs[1]) / args.steps])
        if use_ctx and cx is None and world == 1:
            # host time of ONE product-path call (planner + window cache + descriptor ring + launch inside ss_ctx_observe, plus
            # the bound Python call): each call timed on its own, launch queue kept shallow.  Small steps are bound by THIS,
            # not by the kernel (a 32-env launch on two lanes is 10 us of GPU time per step: profiles/r5/kbench_lanes.txt)
            hs = []
            for k in range(256):
                t_h = time.perf_counter()
                step(k % len(descs))
                hs.append(time.perf_counter() - t_h)
                if k % 32 == 31:
                    ctx.join()
                    torch.cuda.synchronize()
            state["host_us_per_call"] = round(1e6 * float(np.median(hs)), 2)
            if os.environ.get("SS_AB_HOST_PROFILE") and hasattr(ctx.lib, "ss_ab_host_profile"):
                # -DSS_AB libraries only (scripts/gpu_host_profile.sh): the call's host time by segment, queue kept shallow
                import ctypes
                ns, calls = (ctypes.c_double * 8)(), ctypes.c_longlong(0)
                ctx.lib.ss_ab_host_profile(ns, ctypes.byref(calls))
                for k in range(512):
                    step(k % len(descs))
                    if k % 32 == 31:
                        ctx.join()
                        torch.cuda.synchronize()
                ctx.lib.ss_ab_host_profile(ns, ctypes.byref(calls))
                names = ("input_fence", "lane_waits", "ring_slot", "planner", "windows", "launch", "group_event", "-")
                print("host_profile_us " + json.dumps({"calls": calls.value, **{nm: round(v / max(calls.value, 1) / 1e3, 2)
                                                                                for nm, v in zip(names, ns)}}), file=sys.stderr, flush=True)
        if sustain_s > 0 and world == 1:
            # a run long enough for an outside observer (rocm-smi samples once a second): the same steps, cycled over every
            # prepared step of the run (500+ distinct steps over a 1-GiB bank), for ~sustain_s seconds between two fences
            n_prep = len(descs)
            chunk, done = 256, 0
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < sustain_s:
                for k in range(done, done + chunk):
                    step(k % n_prep)
                done += chunk
                if use_ctx and done % 4096 == 0:                   # keep the launch queue shallow (as the spin-up does)
                    ctx.join()
                    torch.cuda.synchronize()
            flush()
            fence()
            dt_s = time.perf_counter() - t_s
            state["sustained"] = {"value": round(N * done / dt_s, 1), "unit": "env-steps/s", "seconds": round(dt_s, 3), "steps": done,
                                  "ms_per_step": round(1e3 * dt_s / done, 5),
                                  "note": "the headline loop cycled over every prepared step of this run for seconds (outside "
                                          "observers sample utilisation once a second); not the reported value"}
        if world > 1:
            tmax = torch.tensor(region_s, device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)            # per region: the slowest rank
            region_s = [float(v) for v in tmax.tolist()]
        elapsed = float(np.median(region_s))
        per_step = None
        if region_ev:
            per_step = region_ev[0] if per_step_events else [float(np.median([e[0] for e in region_ev]))]
        if use_ctx:
            ctx.set_overlap(1)
            ctx.set_chip_share(0)
        state["regions"] = region_s
        last_regions[:] = region_s
        last_sustained.clear()
        last_sustained.update(state.get("sustained") or {})
        last_host[:] = [state.get("host_us_per_call")]
        return elapsed, per_step, note

    last_regions = []
    last_sustained = {}
    last_host = [None]
    spectra = None
    if args.spectral or not args.no_secondary:
        r.rirs.build_spectra()
        spectra = r.rirs.spectra
    exchanging = world > 1 and args.exchange != "none"
    G_head = args.gather_every if exchanging else 0
    # lanes of the overlap mode: 2 (the head of step k+1 under the tail of step k); 3 for steps that fill at most half the chip
    # (<= 128 (unit, ear) rows: cfg1, cfg3, the reference's 5-10 envs) - those are throughput-bound by how many rows are in
    # flight, not by one launch's latency (same box: +10-30 % at 5-32 envs, +2-10 % at 64, -6 % at 128: profiles/r5/NOTES.md
    # section 1f).  Not 4: the runtime multiplexes its streams onto 4 hardware queues by default and a fourth lane then shares
    # one (slower than 2; with GPU_MAX_HW_QUEUES=8 in the environment --streams 4 is the fastest)
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    LANES = (3 if 2 * 2 * N <= n_cus else 2) if args.streams == 0 else max(1, min(4, args.streams))

    def rate(e):
        return {"value": round(world * N * args.steps / e, 1), "ms_per_step": round(1e3 * e / args.steps, 5)}
    # ---- headline (round 6): the product path in the TRAINER-SHAPED loop - one env group, step k+1 ordered behind the consumer
    # of step k's observation (run_loop: dependent), writing into rollouts.observations['spectrogram'][step + 1]; exchange on
    # when there are ranks.  The pipelined figure (independent steps on LANES internal streams: what r1-r5 reported as `value`)
    # is measured right after it and reported as `pipelined`.
    elapsed, head_ev, exchange_note = run_loop(1, G_head, args.spectral, regions=REGIONS, sustain_s=args.sustain, dependent=1)
    head_regions = list(last_regions)
    head_sustained = dict(last_sustained)
    head_host_us = last_host[0]
    side = {}
    e_p, _, _ = run_loop(1, G_head, args.spectral, lanes=LANES, regions=REGIONS)
    pipe_regions = list(last_regions)
    side["pipelined"] = dict(rate(e_p), lanes=LANES, independent_groups_presumed=LANES, host_us_per_call=last_host[0],
                             note=f"consecutive steps issued with NO dependency between them on {LANES} internal streams of the "
                                  f"library (ss_ctx_set_overlap): realisable only by a caller that has {LANES} independent env "
                                  "groups of this size in flight; r1-r5 reported this figure as `value`",
                             **({} if len(pipe_regions) < 2 else
                                {"spread": {k_: round(world * N * args.steps / float(np.quantile(pipe_regions, q_)), 1)
                                            for k_, q_ in (("min", 1.0), ("p10", 0.9), ("p90", 0.1), ("max", 0.0))}}))
    dep = {"groups": 1, "envs_per_group": n_env, "consumer": "one-workgroup kernel per step and group on the caller's stream "
           "(libss_bench.so: stands in for actor_critic.act, ppo_trainer.py:133-150); the next step is ordered behind it",
           "writes_into": "rollouts.observations['spectrogram'][step + 1] ([17, N, 65, T4, 2], 16 slots cycled)"}
    if world == 1 and not args.no_secondary:
        if N % 2 == 0 and not feats:
            e2, _, _ = run_loop(1, 0, args.spectral, regions=REGIONS, dependent=2)
            dep["two_groups"] = dict(rate(e2), envs_per_group=n_env // 2 if rot == 1 else f"{N // 2} units",
                                     note="double-buffered sampler: two groups of half the envs, each a dependent chain on its own "
                                          "caller stream; group B renders while group A's policy runs")
        eh, _, _ = run_loop(1, 0, args.spectral, regions=REGIONS, dependent=1, host_sync=True)
        dep["host_sync"] = dict(rate(eh), note="as `value`, and the host also waits for every step's token before it issues the "
                                               "next step (a trainer that reads its actions back: actions.item())")
    side["dependent"] = dep
    step_dist = None
    # ---- the kernel's own rate: pre-planned descriptors, ONE stream - per-launch durations are separable only without
    # overlap, and this is the average the rocprofv3 kernel trace of the same command reports for the kernel
    per_step = None
    kernel_pass = "preplanned_single_stream"
    if rank == 0 or world > 1:
        e_k, per_step, _ = run_loop(1, 0, args.spectral, regions=REGIONS)
        side["preplanned_single_stream"] = dict(rate(e_k), note="stateless entry point, descriptors planned outside the timed "
                                                "region, one stream: the kernel rate (r1 / r2 headline protocol)")
    if world == 1 and not args.no_secondary:
        _, ps, _ = run_loop(1, 0, args.spectral, per_step_events=True)
        step_dist = dist_of(ps)
        step_dist["note"] = "separate single-stream pass with one HIP event record per step (the records themselves add 2-3 us per step)"
        e1, ps_ctx, _ = run_loop(1, 0, args.spectral, lanes=1)
        side["ctx_single_stream"] = dict(rate(e1), note="ss_ctx_observe without overlap (a caller that joins every step)")
        # Short kernels (small steps: 14-16 us): the pre-planned pass goes through the Python op layer (~19 us of host per launch)
        # and its event average is then the HOST's rate, not the kernel's.  The context pass issues the same kernel from one C
        # call (~7 us of host): the smaller of the two event averages is the one that is bound by the kernel
        if per_step and ps_ctx and float(np.mean(ps_ctx)) < float(np.mean(per_step)):
            per_step, kernel_pass = ps_ctx, "ctx_single_stream"
    if world > 1 and not args.no_secondary:
        #  every exchange arrangement from ONE run of `bench.py --gpus N`, same dependent-step protocol:
        #   exchange_none  no collective (the reference's DD-PPO arrangement: every rank's learner consumes its own slab)
        #   allgather      RCCL all_gather_into_tensor of the step slabs, a chunk of --gather-every steps at a time
        #   gather         peer copies to ONE learner rank (PeerCopyExchange(learners=[0]): xGMI writes, no collective)
        # (the peer-copy arrangement last: everything that only needs RCCL is measured before the IPC transport is set up)
        for key, mode, ge in (("exchange_none", "none", 0), ("allgather", "allgather", args.gather_every),
                              ("exchange_per_step_gather", args.exchange, 1), ("gather", "gather", args.gather_every)):
            if mode == args.exchange and ge == G_head:
                side[key] = dict(rate(elapsed), note="= the headline of this run")
                continue
            try:
                e_x, _, note_x = run_loop(1, ge, args.spectral, dependent=1, exchange=mode)
                side[key] = dict(rate(e_x), **({"note": note_x} if note_x else {}))
            except Exception as ex_:                               # one arrangement failing must not take the line down
                side[key] = {"value": None, "note": f"{type(ex_).__name__}: {ex_}"}
    if world == 1 and not args.no_secondary:
        eo, _, _ = run_loop(1, 0, not args.spectral, lanes=LANES)  # the other RIR bank format: pipelined, dependent, kernel rate
        ed, _, _ = run_loop(1, 0, not args.spectral, regions=REGIONS, dependent=1)
        _, ps_o, _ = run_loop(1, 0, not args.spectral)
        side["spectral_bank" if not args.spectral else "time_domain_bank"] = dict(
            rate(eo), dependent=rate(ed), avg_launch_ms=round(float(np.mean(ps_o)), 5),
            note=("`value` = pipelined (as `pipelined` of the line), `dependent` = the line's own protocol.  " +
                  ("RIR bank stored as block spectra (ss_rir_spectra_f32): no forward FFT per step, 2x the bytes per RIR "
                   "(actual reads per unit: 2*2*L*4 + window spectrum from L2)" if not args.spectral else
                   "launches read the time-domain rows: the format SURVEY 8(d) defines the algorithmic bytes on, and what "
                   "rounds 1-5 reported the headline on")))
    r.rirs.spectra = spectra if args.spectral else None

    # ---- secondary measurement: the convolution kernel alone (audiogoal written), same inputs -------------
    conv_ms = None
    conv_ms_other = None
    if rank == 0 and not args.no_secondary:
        ag = torch.empty((N, 2, sr), dtype=torch.float32, device=dev)

        def conv_alone():
            for k in range(min(10, total)):
                r.render_audiogoal(descs[k], out=ag)
            torch.cuda.synchronize()
            reps = min(100, args.steps)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(reps):
                r.render_audiogoal(descs[args.warmup + k], out=ag)
            e1.record()
            torch.cuda.synchronize()
            return [e0.elapsed_time(e1) / reps]
        conv_ms = conv_alone()
        if spectra is not None:                                # the other bank form, same steps
            r.rirs.spectra = None if args.spectral else spectra
            conv_ms_other = conv_alone()
            r.rirs.spectra = spectra if args.spectral else None

    if rank == 0:
        kernel_ms = float(np.mean(per_step)) if per_step else 1e3 * elapsed / args.steps
        b = bytes_per_unit(sr, L, t4)
        kk = "k_conv_spec" if args.spectral else "k_conv"
        if sr > P.KB:
            # rows of 2-3 blocks: small steps (every output block of every row on a CU of its own) take k_obs_blocks
            small = 2 * N * P.ceil_div(sr, P.KB) <= n_cus and sr <= 3 * P.KB
            kname = f"{'k_obs_blocks' if small else 'k_obs_rows'}<SPECTRAL={'true' if args.spectral else 'false'}>"
        else:
            kname = f"{kk}<FUSE=true>" if fused else f"{kk}<FUSE=false>+k_spectrogram"
        bpu = (b["fused"] + (2 * sr * 4 if args.with_audiogoal else 0)) if fused else b["conv"] + b["spec"]
        if savi:
            bpu = 2 * (2 * L * 4) + 2 * sr * 4 + 65 * t4 * 2 * 4        # two RIRs read, waveform and spectrogram written
            kname = kname.replace("<FUSE=true>", "<FUSE=true,loop>")    # distractor terms: the two-term loop instantiation
        if feats:                                             # k_features: the waveform read once more, the features written;
            # the pooled spectrogram comes from k_features too (it holds every frame's spectrum), the convolution launch drops
            # its fused STFT phase: k_conv<FUSE=false,loop> + k_features<spectrogram,...>
            kname = kname.replace("<FUSE=true", "<FUSE=false")
        traffic_kernels = [kname] + (["k_features<" + ",".join(["spectrogram"] + feats) + ">"] if feats else [])
        if feats:
            T_fr = 1 + sr // 160
            bpu += 2 * sr * 4 + (64 * T_fr * 2 * 4 if "logmel" in feats else 0) + (65 * T_fr * 4 if "gccphat" in feats else 0)
            kname += " + k_features<" + ",".join(["spectrogram"] + feats) + "> (two launches per step: avg_launch_ms is their sum)"
        ach = bpu * N / (kernel_ms * 1e-3) / 1e9
        workload = (("savi semantic_audionav shape: 21 sounds of 1-20 s, a distractor on every env (2 convolutions + add), "
                     "audiogoal AND spectrogram written" + (", + " + " + ".join(feats) + " (64 mels / 65 lags per frame) from one "
                                                            "k_features launch per step, which also pools the spectrogram" if feats else "") + "; " if savi else "") +
                    f"{n_env} envs/GPU x {rot} rotation(s) = {N} units/launch, sr={sr}, " +
                    ("" if savi else f"1-s source clips ({args.sounds} sounds), ") +
                    f"2-ch RIR L={L}, RIR bank {R} entries ({R * 2 * L * 4 >> 20} MiB/GPU, HBM-resident"
                    f"{', stored as block spectra: ' + str(R * 2 * P.ceil_div(L, P.KB) * P.SPEC_FLOATS * 4 >> 20) + ' MiB' if args.spectral else ''}), "
                    "cache-miss path, spectrogram [65,%d,2] f32 out" % t4)
        out = {
            "metric": "audio env-steps/sec (RIR-convolve+spectrogram) per node, 128 envs Replica 16 kHz",
            "value": round(world * N * args.steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": world, "rccl_ranks": world if (world > 1 and args.backend == "nccl") else 0, "steps": args.steps, "warmup": args.warmup, "spinup_steps_untimed": spin_steps,
            "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "timed_regions": len(head_regions),
            "value_spread": (None if len(head_regions) < 2 else
                             {"regions": len(head_regions), "each": f"{args.steps} steps between barrier + synchronize",
                              "value": "median region",
                              **{k_: round(world * N * args.steps / float(np.quantile(head_regions, q_)), 1)
                                 for k_, q_ in (("min", 1.0), ("p10", 0.9), ("p90", 0.1), ("max", 0.0))}}),
            "sustained": head_sustained or None,
            "host_us_per_call": head_host_us,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": n_env, "rotations": rot, "units_per_gpu": N,
                       "sampling_rate": sr, "rir_len": L, "rir_bank": "spectral" if args.spectral else "time-domain",
                       "actual_bytes_per_unit": ((2 * P.ceil_div(L, P.KB) * P.SPEC_FLOATS * 4 if args.spectral else 2 * L * 4)
                                                 + 65 * t4 * 2 * 4 + (0 if fused else 2 * 2 * sr * 4)
                                                 + (2 * sr * 4 if (fused and want_ag) else 0)),
                       "exchange": (exchange_note or ((args.exchange + f" every {args.gather_every} steps") if exchanging else "none")),
                       "path": "DEPENDENT steps (one env group): AudioContext.observe_prepared -> ss_ctx_observe (planner + window "
                               "cache + descriptor ring inside the timed region; the steps' unit columns were converted to the C "
                               "struct ss_units OUTSIDE it, AudioContext.prepare: a vector env that owns its columns) into "
                               "rollouts.observations['spectrogram'][step + 1], then a one-workgroup consumer of those rows on the "
                               "caller's stream (the policy's stand-in); step k+1 is ordered behind it.  One stream, nothing "
                               f"overlaps.  `pipelined` = the r1-r5 protocol ({LANES} internal streams, independent steps)",
                       "streams": 1, "pipelined_streams": LANES, "kernel": kname, "features": feats or None},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": kname,
                         "bytes_per_unit": bpu, "units_per_launch": N, "avg_launch_ms": round(kernel_ms, 5),
                         "pass": f"single-stream pass of the same run over the same steps ({kernel_pass}): per-launch durations are only "
                                 "separable without overlap; HIP events around its timed region (for kernels shorter than the host's "
                                 "issue rate - small steps - the pass with the leaner host side is taken)",
                         "dependent_achieved": round(bpu * N / (elapsed / args.steps) / 1e9, 1),
                         "dependent_frac": round(bpu * N / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "pipeline_achieved": round(bpu * N / (e_p / args.steps) / 1e9, 1),
                         "pipeline_frac": round(bpu * N / (e_p / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        trs = [measured_traffic(N, sr, k_) for k_ in traffic_kernels]
        if all(t_ is not None for t_ in trs):
            out["roofline"]["traffic"] = sum(t_["bytes"] for t_ in trs)
            out["roofline"]["traffic_detail"] = trs[0] if len(trs) == 1 else dict(zip(traffic_kernels, trs))
        if step_dist:
            out["gpu_ms_per_step"] = step_dist
        if fused and sr <= P.KB:
            # FFT convolution sits at the FP32 ridge (~20 FLOP/B), so the vector-FP32 roofline is reported
            # beside the declared HBM one.  Algorithmic FLOPs per env-step (radix-2 real-FFT count 2.5 N log2 N):
            # conv 5.1 MFLOP + STFT 2.3 MFLOP @16 kHz; peak = 157.3 TFLOP/s dense FP32 vector (packed FMA) on MI355X.
            flops = 7.4e6 * N
            tf = flops / (kernel_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "valu-fp32", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s",
                                    "frac": round(tf / 157.3, 4), "flops_per_unit": 7.4e6,
                                    "note": "FFT butterflies are adds (no FMA pairing): the add-rate ceiling is 78.6"}
        if conv_ms is not None:
            cm = float(np.mean(conv_ms))
            a2 = b["conv"] * N / (cm * 1e-3) / 1e9
            out["roofline_conv_only"] = {"bound": "hbm", "achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": round(a2 / HBM_PEAK_GBS, 4),
                                         "kernel": "k_conv_spec<FUSE=false>" if args.spectral else "k_conv<FUSE=false>",
                                         "bytes_per_unit": b["conv"], "avg_launch_ms": round(cm, 5)}
            if conv_ms_other is not None:
                # the same convolution from the OTHER bank form.  Spectral rows are 2 x the bytes (block spectra: 2 x 128 KiB per
                # RIR block and unit) for no forward FFT: `achieved` stays on SURVEY 8(d)'s algorithmic bytes, `actual_*` are the
                # bytes that kernel really reads + writes (profiles/r6/conv_roofline.txt has the counter view of both)
                co = float(np.mean(conv_ms_other))
                other_spectral = not args.spectral
                act = ((2 * P.ceil_div(L, P.KB) * P.SPEC_FLOATS * 4) if other_spectral else 2 * L * 4) + 2 * sr * 4
                a3, a4 = b["conv"] * N / (co * 1e-3) / 1e9, act * N / (co * 1e-3) / 1e9
                out["roofline_conv_only_spectral" if other_spectral else "roofline_conv_only_time_domain"] = {
                    "bound": "hbm", "achieved": round(a3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a3 / HBM_PEAK_GBS, 4),
                    "kernel": "k_conv_spec<FUSE=false>" if other_spectral else "k_conv<FUSE=false>", "bytes_per_unit": b["conv"],
                    "avg_launch_ms": round(co, 5), "actual_bytes_per_unit": act, "actual_achieved": round(a4, 1),
                    "actual_frac": round(a4 / HBM_PEAK_GBS, 4)}
        out.update(side)
        if fused and sr > P.KB:
            flops = (23.1e6 + 6.4e6) * N                      # SURVEY 8(d) @44.1 kHz: conv 23.1 MFLOP + STFT 6.4 MFLOP per unit
            tf = flops / (kernel_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "valu-fp32", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s",
                                    "frac": round(tf / 157.3, 4), "flops_per_unit": 29.5e6}
        if world == 1 and not args.no_plugin_path and sr <= P.KB and rot == 1 and not savi:
            srcs = [r.sources._host[i] for i in range(len(r.sources))]
            out["plugin_path"] = measure_plugin_path(torch, np, dev, sr, n_env, bank, args.sounds, srcs,
                                                     min(args.steps, 400), min(args.warmup, 50),
                                                     spectra if args.spectral else None)
            out["plugin_path"]["rir_bank"] = "spectral" if args.spectral else "time-domain"
            # the figure to quote as "the plugin path": reference-style Python simulator objects (their attribute writes
            # included); `columns` - a struct-of-arrays vector env the reference does not have - is the ceiling
            out["plugin_path"]["headline"] = dict(out["plugin_path"]["bound_sims"], mode="bound_sims")
            out["plugin_path"]["ceiling"] = dict(out["plugin_path"]["columns"], mode="columns")
            # the reference's default arrangement: worker processes + one resolver in the trainer (ss_amd/deferred.py)
            out["plugin_path"]["deferred"] = measure_deferred_path(torch, np, dev, sr, n_env, bank, srcs, 200, 20)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_all_cores"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
