#!/usr/bin/env python
"""The reference's default arrangement END TO END: a multi-process vector env (habitat.VectorEnv: worker processes own the
simulators, observations come back through pipes - ss_baselines/common/env_utils.py:91-107) stepped by a trainer process
that puts the audio observations into its rollout storage.  Same box, same workers, same agent motion, two ways:

  reference  every worker computes its envs' spectrograms itself, on its core, the way SoundSpacesSim does on a cache miss
             (wavfile.read of the pose's RIR + scipy fftconvolve + STFT/pool/log1p: simulator.py:608-666, nav.py:86-100 - the
             oracle's restatement of those functions stands in for the reference here, as in bench.py's cpu_baseline) and sends
             the [65, 26, 2] arrays through the pipe; the trainer stacks them and copies the batch to the GPU (batch_obs,
             ss_baselines/common/utils.py:126-153).
  deferred   the workers' sensors return AudioRequests (ss_amd.deferred.attach_deferred: a few hundred bytes each); the
             trainer renders the whole step in one launch into the rollout rows (DeferredResolver.resolve_observations).

Reported: whole-loop env-steps/s (wall clock around K vector steps: commands out, worker work, pipes, trainer half, the
rollout write finished on the GPU), and the trainer's own time per step.  Poses follow a random walk over the scene
(turn / turn / move, one in three each); the scene's RIRs are resident in the HBM store before the clock starts unless
--no-preload (then the first visits go through the miss path, as they would at the start of a training run).
usage: bench_vector_env.py [--workers 16] [--envs-per-worker 1] [--steps 200] [--no-preload] [--out file.jsonl]"""
import argparse, json, multiprocessing as mp, os, shutil, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "scripts")]
import numpy as np

SR = 16000
NS = types.SimpleNamespace


def make_sims(root, n_nodes, n_envs, seed, sounds):
    from bench import SyntheticSim

    class DSim(SyntheticSim):
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=SR, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = root
        azimuth_angle = property(lambda self: -(self._rotation_angle + 0) % 360)
        current_source_sound = property(lambda self: self._source_sound_dict[self._current_sound])
        _audio_length = property(lambda self: self.current_source_sound.shape[0] // SR)
    rng = np.random.default_rng(seed)
    sims = [DSim(sounds, n_nodes, rng) for _ in range(n_envs)]
    for s in sims:
        s._duration = 10 ** 9
    return sims, rng


def worker_main(rank, conn, mode, root, n_nodes, n_envs, sounds):
    os.environ["OMP_NUM_THREADS"] = "1"
    sims, rng = make_sims(root, n_nodes, n_envs, 1000 + rank, sounds)
    if mode == "deferred":
        from ss_amd.deferred import attach_deferred
        for i, s in enumerate(sims):
            attach_deferred(s, env_rank=rank * n_envs + i)

        def observe(s):
            return s.get_current_spectrogram_observation(None)
    else:
        from scipy.io import wavfile
        from oracle import ss_oracle as O                      # the reference's per-step functions, restated (see the header)

        def observe(s):
            path = os.path.join(root, str(s.azimuth_angle), f"{s._receiver_position_index}_{s._source_position_index}.wav")
            _, rir = wavfile.read(path)                            # simulator.py:615-618: on every cache-missing step
            a = O.compute_audiogoal(s.current_source_sound, rir, SR)
            return O.compute_spectrogram(a).astype(np.float32)
    while True:
        cmd = conn.recv()
        if cmd is None:
            break
        for s, (act, node) in zip(sims, cmd):
            s.move(act, node)
        conn.send([{"spectrogram": observe(s)} for s in sims])
    conn.close()


def run(mode, a, root, n_nodes, sounds):
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for rank in range(a.workers):                                  # fork BEFORE this process touches the GPU
        p_conn, c_conn = ctx.Pipe()
        p = ctx.Process(target=worker_main, args=(rank, c_conn, mode, root, n_nodes, a.envs_per_worker, sounds), daemon=True)
        p.start()
        c_conn.close()
        pipes.append(p_conn)
        procs.append(p)
    import torch
    from ss_amd import planning as P
    from ss_amd.rollout import RolloutStorage
    dev = torch.device("cuda:0")
    N = a.workers * a.envs_per_worker
    space = NS(spaces={"spectrogram": NS(shape=P.spectrogram_shape(SR))})

    class ActionSpace:
        pass
    T = 16
    rollouts = RolloutStorage(T, N, space, ActionSpace(), 8, device=dev)
    res = None
    if mode == "deferred":
        from ss_amd.deferred import DeferredResolver
        from ss_amd.renderer import AudioEngine
        from ss_amd.sim_audio import wav_rir_reader
        eng = AudioEngine(SR, device=dev, rir_slots=4 * n_nodes * n_nodes + 64)
        res = DeferredResolver(eng, rir_reader=wav_rir_reader, fast=True)
        if not a.no_preload:
            res.preload_scene(root)
    rng = np.random.default_rng(7)
    trainer_us = []
    try:
        for k in range(a.warmup + a.steps):
            if k == a.warmup:
                torch.cuda.synchronize()
                t_start = time.perf_counter()
            acts = rng.integers(0, 3, N)
            nodes = rng.integers(0, n_nodes, N)
            for w, pipe in enumerate(pipes):
                lo = w * a.envs_per_worker
                pipe.send([(int(acts[lo + i]), int(nodes[lo + i])) for i in range(a.envs_per_worker)])
            observations = []
            for pipe in pipes:
                assert pipe.poll(120), "worker died or hung"
                observations += pipe.recv()
            t0 = time.perf_counter()
            if mode == "deferred":
                res.resolve_observations(observations, rollouts, replace=False)
            else:                                                   # batch_obs: stack on the host, one copy to the device
                batch = torch.from_numpy(np.stack([o["spectrogram"] for o in observations]))
                rollouts.observation_slot("spectrogram").copy_(batch, non_blocking=False)
            if k >= a.warmup:
                trainer_us.append(1e6 * (time.perf_counter() - t0))
            rollouts.step = (rollouts.step + 1) % T
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_start
    finally:
        for pipe in pipes:
            try:
                pipe.send(None)
            except Exception:
                pass
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.terminate()
    out = {"mode": mode, "workers": a.workers, "envs_per_worker": a.envs_per_worker, "envs": N, "steps": a.steps,
           "env_steps_per_s_whole_loop": round(N * a.steps / wall, 1), "ms_per_vector_step": round(1e3 * wall / a.steps, 3),
           "trainer_half_us_per_step_median": round(float(np.median(trainer_us)), 1),
           "preloaded": mode == "deferred" and not a.no_preload}
    if res is not None:
        out.update(store_misses=int(eng.store.misses), miss_steps=int(res.miss_steps), native_steps=int(res.native_steps))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--envs-per-worker", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=24, help="scene of nodes x nodes (receiver, source) pairs x 4 azimuths")
    ap.add_argument("--no-preload", action="store_true")
    ap.add_argument("--modes", default="reference,deferred,reference,deferred")
    ap.add_argument("--out", default="")
    ap.add_argument("--scene", default="", help="(internal) run ONE mode against this scene directory and print its line")
    a = ap.parse_args()
    if a.scene:                                                    # one mode per process: workers are forked before the GPU is touched
        from oracle import ss_oracle as O                          # synthetic clips only
        sounds = {"sound%d" % i: c for i, c in enumerate(O.synth_sources(np.random.default_rng(1), SR, k=8))}
        print(json.dumps(run(a.modes, a, a.scene, a.nodes, sounds)), flush=True)
        return
    import subprocess
    import bench_loader as BL
    tmp = "/dev/shm/ss_vector_env_bench" if os.path.isdir("/dev/shm") else "/tmp/ss_vector_env_bench"
    root = os.path.join(tmp, "scene")
    n_nodes = BL.make_scene(root, SR, 512, 4 * a.nodes * a.nodes, np.random.default_rng(0))
    try:
        for mode in a.modes.split(","):
            cmd = [sys.executable, os.path.abspath(__file__), "--scene", root, "--modes", mode, "--workers", str(a.workers),
                   "--envs-per-worker", str(a.envs_per_worker), "--steps", str(a.steps), "--warmup", str(a.warmup),
                   "--nodes", str(n_nodes)] + (["--no-preload"] if a.no_preload else [])
            r = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=900)
            lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                print(json.dumps({"mode": mode, "error": r.returncode}), flush=True)
                continue
            print(lines[-1], flush=True)
            if a.out:
                os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
                with open(a.out, "a") as f:
                    f.write(lines[-1] + "\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
