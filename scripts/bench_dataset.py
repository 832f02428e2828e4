#!/usr/bin/env python
"""Items/s of ``ss_amd.datasets.AudioGoalDataset`` (the savi pre-training set) against the reference's way:
``DataLoader(dataset, batch_size=1024, num_workers=8)`` over per-item CPU rendering (ss_baselines/savi/pretraining/
audiogoal_trainer.py:61-67, audiogoal_dataset.py:97-155).  The CPU side is the ORACLE's restatement of compute_audiogoal +
compute_spectrogram behind scipy's wav reader on `--workers` processes (the reference itself cannot be imported here: librosa /
skimage are absent).  Synthetic tree on tmpfs: one scene, `--nodes` fully connected nodes, float32 RIR files of 0.2-1.6 s.
  python scripts/bench_dataset.py [--nodes 24 --batch 1024 --passes 3 --workers 8]   -> one JSON line"""
import argparse
import json
import multiprocessing as mp
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np  # noqa: E402


def build_tree(root, nodes, sr=16000):
    import networkx as nx
    from scipy.io import wavfile
    from oracle import ss_oracle as O
    g = nx.complete_graph(nodes)
    rng = np.random.default_rng(0)
    for n in g.nodes:
        g.nodes[n]["point"] = (float(rng.integers(-9, 10)), 0.0, float(rng.integers(-9, 10)))
    snd = os.path.join(root, "sounds")
    os.makedirs(snd)
    cats = {"chair": 0, "table": 1, "picture": 2, "cabinet": 3, "cushion": 4, "sofa": 5, "bed": 6}
    for i, name in enumerate(cats):
        wavfile.write(os.path.join(snd, name + ".wav"), sr, O.synth_sources(np.random.default_rng(50 + i), sr, k=1, seconds=3 + i)[0])
    n_files = 0
    for a in (0, 90, 180, 270):
        d = os.path.join(root, "rirs", "scene", str(a))
        os.makedirs(d)
        for s in g.nodes:
            for r in g.nodes:
                L = int(rng.integers(int(0.2 * sr), int(1.6 * sr)))
                h = O.synth_rir(np.random.default_rng(1000 * a + 100 * s + r), sr, length=L, n=1)[0]
                wavfile.write(os.path.join(d, f"{r}_{s}.wav"), sr, np.ascontiguousarray(h.T))
                n_files += 1
    return {"scene": g}, cats, n_files


def _cpu_item(args):
    path, clip, index = args
    from scipy.io import wavfile
    from oracle import ss_oracle as O
    _, h = wavfile.read(path)
    a = O.compute_audiogoal_savi_dataset(clip, h if h.shape[0] else O.zero_rir(16000), 16000, index)
    return O.compute_spectrogram(a.astype(np.float32)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--cpu-items", type=int, default=512)
    a = ap.parse_args()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        graphs, cats, n_files = build_tree(td, a.nodes)
        # ---- the reference's way on the CPU (before any CUDA context exists: fork-safe)
        from scipy.io import wavfile
        clips = {f: wavfile.read(os.path.join(td, "sounds", f))[1] for f in sorted(os.listdir(os.path.join(td, "sounds")))}
        rr = random.Random(3)
        files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(td, "rirs")) for f in fs)
        jobs = []
        for _ in range(a.cpu_items):
            snd = rr.choice(sorted(clips))
            jobs.append((rr.choice(files), clips[snd], rr.randint(0, clips[snd].shape[0] // 16000 - 2)))
        with mp.get_context("fork").Pool(a.workers) as pool:
            pool.map(_cpu_item, jobs[:a.workers * 2])
            t0 = time.perf_counter()
            pool.map(_cpu_item, jobs, chunksize=max(1, len(jobs) // (4 * a.workers)))
            cpu_s = time.perf_counter() - t0
        import torch
        from ss_amd.datasets import AudioGoalDataset
        random.seed(1)
        ds = AudioGoalDataset(graphs, ["scene"], "train", binaural_rir_dir=os.path.join(td, "rirs"),
                              source_sound_dir=os.path.join(td, "sounds"), category_index=cats, device="cuda:0", rir_slots=4096)
        res = {"items": len(ds), "rir_files": n_files, "batch": a.batch}
        ld = ds.loader(batch_size=a.batch, seed=5)
        per_pass = []
        for p in range(a.passes):                                  # pass 0 loads every RIR file (cold store), later passes hit
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for inputs, gts in ld:
                n += inputs[0].shape[0]
            torch.cuda.synchronize()
            per_pass.append(n / (time.perf_counter() - t0))
        res["batched_items_per_s"] = {"cold_pass_files_loaded": round(per_pass[0], 1), "warm_passes": [round(v, 1) for v in per_pass[1:]]}
        # a store far smaller than the file set: every batch reloads (the 867-GB data set does not fit any store)
        ds2 = AudioGoalDataset(graphs, ["scene"], "train", binaural_rir_dir=os.path.join(td, "rirs"),
                               source_sound_dir=os.path.join(td, "sounds"), category_index=cats, device="cuda:0",
                               rir_slots=256)                # (576 distinct files per pass through 256 entries: every batch evicts)
        ld2 = ds2.loader(batch_size=min(a.batch, 256), seed=5, shuffle=True)
        miss_rates = []
        for p in range(a.passes):                                  # (pass 0 also grows the bank's rows to the longest RIR)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = sum(inputs[0].shape[0] for inputs, _ in ld2)
            torch.cuda.synchronize()
            miss_rates.append(round(n / (time.perf_counter() - t0), 1))
        res["batched_items_per_s"]["always_missing_store"] = miss_rates[-1]
        res["batched_items_per_s"]["always_missing_store_passes"] = miss_rates
        res["batched_items_per_s"]["always_missing_store_files_read"] = int(ds2.engine.store.misses)
        t0 = time.perf_counter()
        k = min(256, len(ds))
        for i in range(k):
            ds[i]
        torch.cuda.synchronize()
        res["per_item_getitem_items_per_s"] = round(k / (time.perf_counter() - t0), 1)
        res["cpu_reference_way"] = {"items_per_s": round(len(jobs) / cpu_s, 1), "workers": a.workers, "items": len(jobs),
                                    "kind": "port (oracle restatement + scipy wavfile.read per item on worker processes)"}
        res["speedup_warm"] = round(float(np.median(per_pass[1:] or per_pass)) / res["cpu_reference_way"]["items_per_s"], 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
