#!/usr/bin/env python
"""Length-bucketed RIR bank vs one bank at the longest capacity: a 70 / 30 mix of short
(0.3-1.0 s) and 3-s RIRs, 3-s source clips (the steady branch hears the whole tail), 128 envs @16 kHz, spectrogram only.
Prints ONE JSON line: env-steps/s and HBM bytes of the bank for (a) BucketedRirBank [short bucket cap = sr, long bucket
cap = 3 sr], (b) the same RIRs in one bank whose every row has the long capacity, (c) a step whose units all sit in the
short bucket (keeps the loop-free kernel: SS_FLAG_FIRST_BUCKET).
usage: python scripts/bench_buckets.py [--envs 128 --steps 200 --long-frac 0.3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd import ops, planning as P
from ss_amd.renderer import BatchedAudioRenderer, BucketedRirBank, RirBank, UnitRequest

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=128)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--long-frac", type=float, default=0.3)
ap.add_argument("--n-short", type=int, default=1400)
ap.add_argument("--n-long", type=int, default=600)
a = ap.parse_args()
dev, sr = torch.device("cuda:0"), 16000
rng = np.random.default_rng(5)
lens = [int(rng.uniform(0.3, 1.0) * sr) for _ in range(a.n_short)] + [3 * sr] * a.n_long
g = torch.Generator(device=dev); g.manual_seed(1)
rirs = []
for L in lens:                                        # decaying noise, generated on the device (2000 RIRs of up to 3 s)
    k = torch.arange(L, device=dev, dtype=torch.float32)
    h = torch.randn((2, L), device=dev, generator=g) * torch.exp(-6.9 * k / (0.4 * L)) * 0.1
    h[:, 5] += 0.5
    rirs.append(h)
def bank_of(idx, cap):
    data = torch.zeros((len(idx), 2, cap), device=dev)
    for r, i in enumerate(idx):
        data[r, :, :lens[i]] = rirs[i]
    return RirBank(data, torch.tensor([lens[i] for i in idx], dtype=torch.int32, device=dev))
short_idx, long_idx = list(range(a.n_short)), list(range(a.n_short, a.n_short + a.n_long))
b_short, b_long = bank_of(short_idx, sr), bank_of(long_idx, 3 * sr)
lengths = torch.cat([b_short.lengths, b_long.lengths])
b_short.lengths, b_long.lengths = lengths[:a.n_short], lengths[a.n_short:]
bucketed = BucketedRirBank([b_short, b_long], lengths)
one = bank_of(short_idx + long_idx, 3 * sr)
src = O.synth_sources(rng, sr, k=8, seconds=3)

def renderer(bank):
    r = BatchedAudioRenderer(sr, device=dev)
    for i, s in enumerate(src):
        r.add_source(f"s{i}", s)
    r.set_rir_bank(bank)
    return r
def steps(long_frac):
    out = []
    for _ in range(a.steps + 20):
        is_long = rng.uniform(size=a.envs) < long_frac
        rir = np.where(is_long, a.n_short + rng.integers(0, a.n_long, a.envs), rng.integers(0, a.n_short, a.envs))
        out.append([UnitRequest(int(rng.integers(0, 8)), int(rng.integers(0, 3)) * sr, int(h)) for h in rir])
    return out
def timeit(r, plans):
    sg = torch.empty((a.envs,) + r.spectrogram_shape, device=dev)
    for k in range(600):                              # clocks
        r.render(plans[k % 20], spectrogram_out=sg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(a.steps):
        r.render(plans[20 + k], spectrogram_out=sg)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.steps * 1e3
    return {"env_steps_per_s": round(a.envs / us * 1e6, 1), "us_per_step": round(us, 2)}
rb, r1 = renderer(bucketed), renderer(one)
mix, shorts = steps(a.long_frac), steps(0.0)
pm_b, pm_1 = [rb.plan(u) for u in mix], [r1.plan(u) for u in mix]
ps_b, ps_1 = [rb.plan(u) for u in shorts], [r1.plan(u) for u in shorts]
chk_b, chk_1 = rb.render(pm_b[0])[1], r1.render(pm_1[0])[1]
assert float((chk_b - chk_1).abs().max()) <= 2e-6 * float(chk_1.abs().max())
mb = lambda *banks: round(sum(b.data.numel() * 4 for b in banks) / 2 ** 20, 1)
print(json.dumps({
    "what": f"{a.envs} envs @16 kHz, 3-s clips, RIRs: {a.n_short} of 0.3-1.0 s + {a.n_long} of 3 s; units draw a 3-s RIR with p = {a.long_frac}",
    "bucketed_mix": dict(timeit(rb, pm_b), bank_mib=mb(b_short, b_long), flags=pm_b[0].flags),
    "one_bank_mix": dict(timeit(r1, pm_1), bank_mib=mb(one), flags=pm_1[0].flags),
    "bucketed_short_only_step": dict(timeit(rb, ps_b), flags=ps_b[0].flags, kernel="loop-free (SS_FLAG_FIRST_BUCKET)"),
    "one_bank_short_only_step": dict(timeit(r1, ps_1), flags=ps_1[0].flags, kernel="loop kernel (row capacity 3 blocks)")}))
