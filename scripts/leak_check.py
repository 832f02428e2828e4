#!/usr/bin/env python
"""Device-memory leak check: free HBM before / after (a) 30 000 overlapped steps through one context, (b) 60 create / use / destroy
cycles of contexts with overlap lanes at 44.1 kHz (k_obs_rows' per-stream stash: ADVICE r3), (c) 1 500 deferred steps with live
SoundSpaces-2.0 RIRs (batched pinned uploads).  Prints the deltas; exits 1 if any exceeds 64 MiB."""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd.context import AudioContext
from ss_amd.renderer import AudioEngine, RirBank
from ss_amd import _lib

dev = "cuda:0"
rng = np.random.default_rng(0)


def free_mib():
    torch.cuda.synchronize()
    gc.collect()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


bad = False
# (a)
sr, N = 16000, 128
src = O.synth_sources(rng, sr, k=4)
bank = RirBank(torch.from_numpy(O.synth_rir(rng, sr, n=512)).to(dev), torch.full((512,), sr, dtype=torch.int32, device=dev))
ctx = AudioContext(sr)
for i, s_ in enumerate(src):
    ctx.add_source(f"s{i}", s_)
ctx.set_rir_bank(bank.data, bank.lengths)
ctx.set_overlap(2)
sg = [torch.empty((N,) + ctx.spectrogram_shape, device=dev) for _ in range(2)]
cols = [(rng.integers(0, 4, N), np.zeros(N, np.int64), rng.integers(0, 512, N)) for _ in range(16)]
for k in range(200):
    ctx.observe(*cols[k % 16], spectrogram_out=sg[k & 1])
ctx.join()
f0 = free_mib()
for k in range(30000):
    ctx.observe(*cols[k % 16], spectrogram_out=sg[k & 1])
    if k % 512 == 511:
        ctx.join(); torch.cuda.synchronize()
ctx.join()
d = f0 - free_mib()
print(f"(a) 30000 overlapped steps: free memory changed by {d:+.1f} MiB")
bad |= d > 64
ctx.close()
# (b)
sr2 = 44100
src2 = O.synth_sources(rng, sr2, k=2)
bank2 = RirBank(torch.from_numpy(O.synth_rir(rng, sr2, n=32)).to(dev), torch.full((32,), sr2, dtype=torch.int32, device=dev))
sg2 = torch.empty((64, 65, 69, 2), device=dev)


def cycle():
    c = AudioContext(sr2)
    for i, s_ in enumerate(src2):
        c.add_source(f"s{i}", s_)
    c.set_rir_bank(bank2.data, bank2.lengths)
    c.set_overlap(2)
    for _ in range(4):
        c.observe(rng.integers(0, 2, 64), np.zeros(64, np.int64), rng.integers(0, 32, 64), spectrogram_out=sg2)
    c.join(); torch.cuda.synchronize()
    c.close()


for _ in range(3):
    cycle()
f0 = free_mib()
for _ in range(60):
    cycle()
d = f0 - free_mib()
print(f"(b) 60 context create / observe at 44.1 kHz with overlap / destroy cycles: free memory changed by {d:+.1f} MiB")
bad |= d > 64
# (c)
from fakes import FakeContinuousSim
from ss_amd.deferred import DeferredResolver, attach_deferred
import pickle
sounds = {"s.wav": O.synth_sources(rng, sr, k=1)[0]}
n_env = 32
pools = [[np.ascontiguousarray(h) for h in O.synth_rir(np.random.default_rng(100 + i), sr, length=6000, n=6)] for i in range(n_env)]
sims = [FakeContinuousSim(sr, sounds, lambda k, pool=pools[i]: pool[k % 6], step_time=0.25, crossfade=True) for i in range(n_env)]
for i, s_ in enumerate(sims):
    s_._duration = 10 ** 9
    attach_deferred(s_, env_rank=i, continuous=True)
eng = AudioEngine(sr, device=dev, rir_slots=4 * n_env, step_time=0.25, wrap=True)
res = DeferredResolver(eng)
def step():
    for s_ in sims:
        s_.step()
    res.resolve([s_.get_current_spectrogram_observation(None) for s_ in sims])
for _ in range(50):
    step()
f0 = free_mib()
for _ in range(1500):
    step()
d = f0 - free_mib()
print(f"(c) 1500 deferred SS2.0 steps x {n_env} envs (live RIRs, cross-fade): free memory changed by {d:+.1f} MiB")
bad |= d > 64
sys.exit(1 if bad else 0)
