#!/usr/bin/env python
"""The RIR MISS path, measured.  The reference reads one wav per cache-missing step from an
867 GB set (soundspaces/simulator.py:615-624, soundspaces/README.md:9); every other number of this repository assumes the
poses are resident.  Two measurements on float32 stereo wav files on tmpfs (/dev/shm):

  (a) scene load: ``load_scene_rirs`` over <scene>/<azimuth>/<recv>_<src>.wav at 16 kHz and 44.1 kHz -> files/s and GB/s,
      through the library's native reader (ss_wav_read_rirs_f32 -> pinned block -> one H2D per 256 rows) and through the
      previous path (scipy.io.wavfile on a thread pool), next to the box's own memcpy and pinned H2D rates;
  (b) steps with pose misses: deferred mode (habitat.VectorEnv arrangement, trainer half) and batched in-process mode at
      1 / 5 / 25 % of the envs standing on a never-seen pose per step, 128 envs, 16 kHz.

usage: python scripts/bench_loader.py [--files 2048] [--steps 60] [--out gpurun_out/loader.json]"""
import argparse
import json
import os
import shutil
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np
import torch
from scipy.io import wavfile

from bench import SyntheticSim
from oracle import ss_oracle as O
from ss_amd import planning as P
from ss_amd.renderer import AudioEngine, RirStore, load_scene_rirs
from ss_amd.sim_audio import wav_rir_reader


def sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def make_scene(root, sr, n_files, n_names, rng):
    """n_names files <root>/<az>/<r>_<s>.wav, hard links onto n_files distinct float32 stereo files of sr frames"""
    if os.path.isdir(root):
        shutil.rmtree(root)
    n_nodes = int(np.ceil(np.sqrt(n_names / 4)))
    for az in (0, 90, 180, 270):
        os.makedirs(os.path.join(root, str(az)))
    base = os.path.join(root, "_distinct")
    os.makedirs(base)
    for i in range(n_files):
        wavfile.write(os.path.join(base, f"{i}.wav"), sr, (rng.standard_normal((sr, 2)) * 0.05).astype(np.float32))
    k = 0
    for r in range(n_nodes):
        for s in range(n_nodes):
            for az in (0, 90, 180, 270):
                os.link(os.path.join(base, f"{k % n_files}.wav"), os.path.join(root, str(az), f"{r}_{s}.wav"))
                k += 1
    return n_nodes


def box_rates(dev):
    a = torch.ones((64 << 20,), dtype=torch.float32)
    b = torch.empty_like(a)
    b.copy_(a)
    t = time.perf_counter(); b.copy_(a); memcpy = a.numel() * 4 / (time.perf_counter() - t) / 1e9
    if dev.type != "cuda":
        return {"host_memcpy_GBps_one_thread": round(memcpy, 2)}
    p = torch.ones((64 << 20,), dtype=torch.float32, pin_memory=True)
    d = torch.empty_like(p, device=dev)
    d.copy_(p, non_blocking=True); sync()
    t = time.perf_counter(); d.copy_(p, non_blocking=True); sync()
    h2d = p.numel() * 4 / (time.perf_counter() - t) / 1e9
    return {"host_memcpy_GBps_one_thread": round(memcpy, 2), "pinned_h2d_GBps": round(h2d, 2)}


def scene_load(dev, root, sr, n, native, workers):
    store = RirStore(n, sr, dev, truncate_to=sr)
    reader = wav_rir_reader if native else (lambda p: wav_rir_reader(p))      # (a wrapper is not "the stock reader")
    load_scene_rirs(store, root, reader, limit=256, workers=workers)        # staging blocks / thread pool exist
    store.clear()
    sync()
    t = time.perf_counter()
    got = load_scene_rirs(store, root, reader, limit=n, workers=workers)
    sync()
    dt = time.perf_counter() - t
    assert got == n and store.misses >= n
    return {"files": n, "seconds": round(dt, 4), "files_per_s": round(n / dt, 1), "GBps": round(n * sr * 8 / dt / 1e9, 3)}


def miss_steps(dev, root, sr, n_nodes, n_envs, rate, steps, native, mode, sources, profile=False, full_store=False, in_call=True):
    """trainer half of a vector step with round(rate * n_envs) envs on a never-seen pose; full_store: a store that holds the
    resident set and little more, so every miss evicts (the steady state against a data set larger than the HBM set aside)"""
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.rollout import RolloutStorage
    from ss_amd import sim_audio
    NS = types.SimpleNamespace
    rng = np.random.default_rng(3)
    sounds = {"sound%d" % i: c for i, c in enumerate(sources)}

    class DSim(SyntheticSim):
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = root
        azimuth_angle = property(lambda self: -(self._rotation_angle + 0) % 360)
        current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])
        _audio_length = property(lambda self: self.current_source_sound.shape[0] // sr)

    reader = wav_rir_reader if native else (lambda p: wav_rir_reader(p))
    m = max(1, int(round(rate * n_envs)))
    warm = 10
    poses = [(r, s, az) for r in range(n_nodes) for s in range(n_nodes) for az in range(4)]
    rng.shuffle(poses)
    n_res = 4 * n_envs                                         # poses resident before the clock starts
    assert len(poses) >= n_res + (steps + warm) * m, "scene too small for this miss rate"
    eng = AudioEngine(sr, device=dev, rir_slots=(n_res + 2 * m + 8) if full_store else n_res + (steps + warm + 1) * m + 64)
    sims = [DSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    for s_ in sims:
        s_._duration = 10 ** 9
    space = NS(spaces={"spectrogram": NS(shape=P.spectrogram_shape(sr))})

    class ActionSpace:
        pass
    rollouts = RolloutStorage(16, n_envs, space, ActionSpace(), 8, device=dev)
    if mode == "deferred":
        res = DeferredResolver(eng, rir_reader=reader, fast=True)
        res.native_miss_path = bool(in_call)                     # False: report -> load_files -> call again (r5's three calls)
        for i, sim in enumerate(sims):
            attach_deferred(sim, env_rank=i)

        def step():
            obs = [{"spectrogram": sim.get_current_spectrogram_observation(None)} for sim in sims]
            t0 = time.perf_counter()
            res.resolve_observations(obs, rollouts, replace=False)
            return time.perf_counter() - t0
    else:
        vobs = sim_audio.VectorAudioObserver(eng, [sim_audio.attach(s_, eng, rir_reader=reader) for s_ in sims])

        def step():
            t0 = time.perf_counter()
            vobs.observe_into(rollouts)
            return time.perf_counter() - t0

    def place(sim, pose):
        sim._receiver_position_index, sim._source_position_index, sim._rotation_angle = pose[0], pose[1], 90 * pose[2]
        sim._episode_step_count += 1
    resident = poses[:n_res]
    for lo in range(0, n_res, n_envs):                          # the resident set
        for sim, pose in zip(sims, resident[lo:lo + n_envs]):
            place(sim, pose)
        step()
    fresh = iter(poses[n_res:])
    sync()
    host = []
    import cProfile
    pr = cProfile.Profile() if profile else None
    for k in range(warm + steps):
        if k == warm:
            sync()
            t_start = time.perf_counter()
            if pr:
                pr.enable()
        movers = set(rng.choice(n_envs, m, replace=False).tolist())
        for i, sim in enumerate(sims):
            place(sim, next(fresh) if i in movers else resident[int(rng.integers(0, len(resident)))])
        dt = step()
        rollouts.step = (rollouts.step + 1) % 16
        if k >= warm:
            host.append(dt)
    sync()
    wall = time.perf_counter() - t_start
    if pr:
        import io, pstats
        pr.disable()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(30)
        print(st.getvalue()[:7000], flush=True)
    hm = float(np.median(host))
    lib_loaded = getattr(res, "library_loaded", 0) if mode == "deferred" else getattr((vobs._rec or {}).get("res"), "library_loaded", 0)
    return {"mode": mode, "reader": ("native" if in_call else "native, three calls (r5)") if native else "scipy", "miss_rate": rate,
            "new_poses_per_step": m, "poses_loaded_inside_the_call": int(lib_loaded),
            "store": f"full: {eng.store.slots} slots, every miss evicts" if full_store else "roomy: no evictions",
            "trainer_half_us_per_step_median": round(1e6 * hm, 1), "trainer_half_us_per_step_mean": round(1e6 * float(np.mean(host)), 1),
            "env_steps_per_s_trainer_half": round(n_envs / float(np.mean(host)), 1),
            "env_steps_per_s_wall_incl_worker_half": round(n_envs * steps / wall, 1),
            "store_misses": eng.store.misses}


def walk_steps(dev, root, sr, n_nodes, n_envs, steps, prefetch, sources):
    """Agent-like motion through a scene that is NOT resident: every env starts an episode at a random (receiver, source)
    and per step turns left / right (same node, next azimuth) or moves forward to a neighbouring receiver, one action in
    three each (habitat's MOVE_FORWARD / TURN_LEFT / TURN_RIGHT); a new episode every 40 steps.  Deferred mode, trainer half;
    with and without the sibling-azimuth prefetch of DeferredResolver."""
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.rollout import RolloutStorage
    NS = types.SimpleNamespace
    rng = np.random.default_rng(5)
    sounds = {"sound%d" % i: c for i, c in enumerate(sources)}

    class DSim(SyntheticSim):
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = root
        azimuth_angle = property(lambda self: -(self._rotation_angle + 0) % 360)
        current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])
        _audio_length = property(lambda self: self.current_source_sound.shape[0] // sr)

    eng = AudioEngine(sr, device=dev, rir_slots=4 * n_nodes * n_nodes)
    res = DeferredResolver(eng, rir_reader=wav_rir_reader, fast=True, prefetch_azimuths=prefetch)
    sims = [DSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    for i, sim in enumerate(sims):
        sim._duration = 10 ** 9
        attach_deferred(sim, env_rank=i)
    space = NS(spaces={"spectrogram": NS(shape=P.spectrogram_shape(sr))})

    class ActionSpace:
        pass
    rollouts = RolloutStorage(16, n_envs, space, ActionSpace(), 8, device=dev)
    host, warm = [], 5
    for k in range(warm + steps):
        if k == warm:
            sync()
            m0, ms0 = eng.store.misses, res.miss_steps
        for sim in sims:
            if k % 40 == 0:                                     # episode start
                sim._receiver_position_index, sim._source_position_index = int(rng.integers(0, n_nodes)), int(rng.integers(0, n_nodes))
                sim._rotation_angle = 90 * int(rng.integers(0, 4))
            else:
                a = int(rng.integers(0, 3))
                if a == 0:
                    sim._receiver_position_index = int(np.clip(sim._receiver_position_index + (1 if rng.integers(0, 2) else -1), 0, n_nodes - 1))
                else:
                    sim._rotation_angle = (sim._rotation_angle + (90 if a == 1 else 270)) % 360
            sim._episode_step_count += 1
        obs = [{"spectrogram": sim.get_current_spectrogram_observation(None)} for sim in sims]
        t0 = time.perf_counter()
        res.resolve_observations(obs, rollouts, replace=False)
        dt = time.perf_counter() - t0
        rollouts.step = (rollouts.step + 1) % 16
        if k >= warm:
            host.append(dt)
    sync()
    return {"mode": "deferred, agent-like walk through a scene that is not resident", "prefetch_azimuths": prefetch, "envs": n_envs,
            "steps": steps, "trainer_half_us_per_step_mean": round(1e6 * float(np.mean(host)), 1),
            "trainer_half_us_per_step_median": round(1e6 * float(np.median(host)), 1),
            "env_steps_per_s_trainer_half": round(n_envs / float(np.mean(host)), 1),
            "files_read": eng.store.misses - m0, "steps_with_misses": res.miss_steps - ms0, "files_prefetched_in_all": res.prefetched}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2048, help="distinct files per rate (hard-linked to the scene's names)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--out", default="")
    ap.add_argument("--profile-miss", type=float, default=0.0, help="only: cProfile of deferred-mode steps at this miss rate")
    ap.add_argument("--device", default="cuda:0", help="cpu: the scene-load half only, into a host store (functional check)")
    a = ap.parse_args()
    dev = torch.device(a.device)
    tmp = "/dev/shm/ss_loader_bench" if os.path.isdir("/dev/shm") else "/tmp/ss_loader_bench"
    rng = np.random.default_rng(0)
    out = {"tmpfs": tmp, "cores": len(os.sched_getaffinity(0)), "box": box_rates(dev), "scene_load": [], "miss_steps": []}
    if a.profile_miss > 0:
        sr = 16000
        root = os.path.join(tmp, f"scene{sr}")
        need = 4 * a.envs + (a.steps + 12) * max(1, int(round(a.profile_miss * a.envs)))
        n_nodes = make_scene(root, sr, min(a.files, 512), max(need + 64, 4096), rng)
        r = miss_steps(dev, root, sr, n_nodes, a.envs, a.profile_miss, a.steps, True, "deferred", O.synth_sources(rng, sr, k=8), profile=True)
        print(json.dumps(r), flush=True)
        shutil.rmtree(tmp, ignore_errors=True)
        return
    for sr, n in ((16000, 2 * a.files), (44100, a.files)):
        root = os.path.join(tmp, f"scene{sr}")
        make_scene(root, sr, a.files if sr == 16000 else a.files // 2, n, rng)
        for native in (True, False, True, False):
            r = scene_load(dev, root, sr, n, native, a.workers)
            r.update(sr=sr, reader="native" if native else "scipy", workers=a.workers)
            out["scene_load"].append(r)
            print(json.dumps(r), flush=True)
        if sr != 16000:
            shutil.rmtree(root)
    if dev.type != "cuda":
        shutil.rmtree(tmp, ignore_errors=True)
        return
    sr = 16000
    root = os.path.join(tmp, f"scene{sr}")
    need = 4 * a.envs + (a.steps + 12) * max(1, int(round(0.25 * a.envs)))
    n_nodes = make_scene(root, sr, a.files, max(need + 64, 4096), rng)
    sources = O.synth_sources(rng, sr, k=8)
    for mode in ("deferred", "batched"):
        for rate in (0.01, 0.05, 0.25):
            for native, in_call in ((True, True), (True, False), (False, True)):
                if mode == "batched" and not in_call:
                    continue
                r = miss_steps(dev, root, sr, n_nodes, a.envs, rate, a.steps, native, mode, sources, in_call=in_call)
                out["miss_steps"].append(r)
                print(json.dumps(r), flush=True)
    for rep in range(2):                                        # ... and with a FULL store (every miss evicts an old pose)
        r = miss_steps(dev, root, sr, n_nodes, a.envs, 0.05, a.steps, True, "deferred", sources, full_store=True)
        out["miss_steps"].append(r)
        print(json.dumps(r), flush=True)
    for envs in (8, a.envs):                                    # the reference's per-GPU env count, and the headline's
        for prefetch in (True, False, True, False):
            r = walk_steps(dev, root, sr, n_nodes, envs, 320 if envs < 32 else 160, prefetch, sources)
            out.setdefault("walk", []).append(r)
            print(json.dumps(r), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
