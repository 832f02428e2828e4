"""A/B for tests/test_wav_loader.py::test_store_writes_go_behind_steps_in_flight_in_overlap_mode: the same scenario with the
store's `before_device_write` hook removed must corrupt the queued steps (otherwise the test proves nothing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd")]
import numpy as np, torch
from oracle import ss_oracle as O
from ss_amd.renderer import AudioEngine

for hook in (False, True):
    sr, n_units, n_steps = 16000, 512, 30
    rng = np.random.default_rng(22)
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=16000, n=1)[0].T) for _ in range(4)]
    clip = O.synth_sources(rng, sr, k=1)[0]
    eng = AudioEngine(sr, device="cuda:0", rir_slots=3)
    sid = eng.source_id("s", clip)
    slots = []
    for k in range(3):
        eng.begin_batch()
        slots.append(eng.rir_slot(("pose", k), lambda k=k: rirs[k]))
    ctx = eng.context()
    if not hook:
        eng.store.before_device_write = None
    cols = dict(sound=np.full(n_units, sid), t0=np.zeros(n_units, np.int64), rir=np.full(n_units, slots[0]))
    ref = torch.empty((n_units, 65, 26, 2), device="cuda:0")
    eng.observe_columns(cols, spectrogram_out=ref)
    torch.cuda.synchronize()
    ctx.set_overlap(2)
    out = torch.zeros((n_steps, n_units, 65, 26, 2), device="cuda:0")
    torch.cuda.synchronize()
    for k in range(n_steps):
        eng.observe_columns(cols, spectrogram_out=out[k])
    eng.begin_batch()
    new = eng.rir_slot(("pose", 3), lambda: rirs[3])
    sg = torch.empty((1, 65, 26, 2), device="cuda:0")
    eng.observe_columns(dict(sound=np.array([sid]), t0=np.zeros(1, np.int64), rir=np.array([new])), spectrogram_out=sg)
    ctx.join()
    torch.cuda.synchronize()
    bad = [k for k in range(n_steps) if not torch.equal(out[k], ref)]
    print(f"hook={hook}: {len(bad)} of {n_steps} queued steps corrupted", bad[:8])
