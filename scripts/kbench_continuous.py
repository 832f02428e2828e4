#!/usr/bin/env python
"""Kernel-level timing of a cross-faded SS2.0 step at 44.1 kHz (bank resident, descriptors planned once):
   python scripts/kbench_continuous.py [units] [with_audiogoal 0|1] [crossfade 0|1] [split 0|1] [sr 44100|16000]  -> us per step (HIP events, 200 steps)
split = 1: the two-kernel formulation (convolution kernel with the waveform written, then k_spectrogram) instead of ss_audio_obs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with_ag = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
xfade = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
split = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
from ss_amd import ops
sr, dev = (int(sys.argv[5]) if len(sys.argv) > 5 else 44100), "cuda:0"
rng = np.random.default_rng(0)
r = BatchedAudioRenderer(sr, device=dev, step_time=0.25, wrap=True)
for i, c in enumerate(O.synth_sources(rng, sr, k=8)):
    r.add_source(f"s{i}", O.tile_short_source(c, sr))
R = 512
bank = torch.from_numpy(O.synth_rir(rng, sr, n=R)).to(dev)
r.set_rir_bank(RirBank(bank, torch.full((R,), sr, dtype=torch.int32, device=dev)))
plans = []
for k in range(8):
    idx = rng.integers(sr, 2 * sr, N)
    plans.append(r.plan([UnitRequest(int(rng.integers(0, 8)), int(idx[i]), int(rng.integers(0, R)), wrap=True,
                                     last_rir=int(rng.integers(0, R)) if xfade else -1, last_wrap=True) for i in range(N)]))
sg = torch.empty((N,) + r.spectrogram_shape, device=dev)
ag = torch.empty((N, 2, sr), device=dev) if (with_ag or split) else None


def go(k):
    if split:
        r.render_audiogoal(plans[k % 8], out=ag)
        ops.spectrogram_into(ag, sg, r.pad_mode)
    else:
        r.render(plans[k % 8], spectrogram_out=sg, audiogoal_out=ag)


for k in range(40):
    go(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(200):
    go(k)
e1.record()
torch.cuda.synchronize()
print(f"N={N} sr={sr} crossfade={xfade} split={split} with_audiogoal={with_ag or split}: {1e3 * e0.elapsed_time(e1) / 200:.1f} us per step")
