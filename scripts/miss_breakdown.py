#!/usr/bin/env python
"""Where a miss step's host time goes: perf_counter around the stages of the trainer half (deferred mode, 128 envs, a fixed
share of the envs on a never-seen pose per step; scripts/bench_loader.py has the headline numbers).
usage: miss_breakdown.py [--rate 0.05] [--steps 60]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "scripts")]
import numpy as np, torch
import bench_loader as BL
from ss_amd import renderer as R, deferred as D, _lib
from ss_amd.context import AudioContext
acc = {}
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[tag] = acc.get(tag, 0) + time.perf_counter() - t
            acc[tag + "#"] = acc.get(tag + "#", 0) + 1
    setattr(obj, name, g)
wrap(D.DeferredResolver, "_serve_misses", "serve_misses")
wrap(D.DeferredResolver, "_load_pairs", "load_pairs")
wrap(D.DeferredResolver, "_request_tables", "request_tables")
wrap(D.DeferredResolver, "resolve_records", "resolve_records")
wrap(R.RirStore, "load_files", "load_files")
wrap(R.AudioEngine, "observe_requests", "observe_requests")
wrap(_lib, "wav_read_rirs", "wav_read_rirs")
wrap(R.RirStore, "_scatter_staged", "scatter_staged")
wrap(R.RirStore, "_take_slots", "take_slots")
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--rate", type=float, default=0.05); ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--full-store", action="store_true", help="a store that holds the resident set and little more: every miss evicts")
a = ap.parse_args()
from oracle import ss_oracle as O
sr, envs = 16000, 128
dev = torch.device("cuda:0")
tmp = "/dev/shm/ss_miss_breakdown"
rng = np.random.default_rng(0)
root = os.path.join(tmp, "scene")
m = max(1, int(round(a.rate * envs)))
need = 4 * envs + (a.steps + 12) * m
n_nodes = BL.make_scene(root, sr, 512, max(need + 64, 4096), rng)
sources = O.synth_sources(rng, sr, k=8)
runs = []
for rep in range(3):                                        # (the first run also pays first-use costs: the last one is printed)
    acc.clear()
    r = BL.miss_steps(dev, root, sr, n_nodes, envs, a.rate, a.steps, True, "deferred", sources, full_store=a.full_store)
    runs.append((r, dict(acc)))
r, acc = runs[-1]
n = acc.get("serve_misses#", 1)
print("%s, %d new poses per 128-env step: median %.1f us per step over %s (runs: %s); us per miss step:" % (
    "FULL store (every miss evicts)" if a.full_store else "roomy store", m, r["trainer_half_us_per_step_median"],
    "%d miss steps" % n, ", ".join("%.0f" % q[0]["trainer_half_us_per_step_median"] for q in runs)), flush=True)
for k in ("resolve_records", "observe_requests", "request_tables", "serve_misses", "load_pairs", "load_files", "wav_read_rirs",
          "take_slots", "scatter_staged"):
    if k in acc:
        print("   %-18s %8.1f   (%d calls)" % (k, 1e6 * acc[k] / n, acc.get(k + "#", 0)), flush=True)
import shutil; shutil.rmtree(tmp, ignore_errors=True)
