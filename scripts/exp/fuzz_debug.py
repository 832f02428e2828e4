"""Debug helper for scripts/gpu_fuzz.py: re-run ONE trial and print every unit's error, for unit subsets."""
import sys, os, warnings
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import gpu_fuzz as m
from oracle import ss_oracle as O
from ss_amd.renderer import BatchedAudioRenderer, RirBank

seed, trial = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, trial])
sr, srcs, rirs, units, keys = m.draw_trial(rng)
print("sr", sr, "sources", [len(s) for s in srcs], "rirs", [len(h) for h in rirs], "units", len(units))


def errs(sel, spectral):
    r = BatchedAudioRenderer(sr, device="cuda:0")
    for i, s in enumerate(srcs):
        r.add_source(f"s{i}", s)
    r.set_rir_bank(RirBank.from_arrays(rirs, "cuda:0"))
    if spectral:
        r.rirs.build_spectra()
    us = [units[n] for n in sel]
    plan = r.plan(us)
    ag = r.render(plan, want_audiogoal=True)[0].cpu().numpy()
    bad = []
    for j, n in enumerate(sel):
        s, idx, h, silent, ds, dh = keys[n]
        if silent:
            continue
        a = O.compute_audiogoal(srcs[s], rirs[h], sr, idx, silent, srcs[ds] if ds >= 0 else None, rirs[dh] if ds >= 0 else None)
        e = O.relerr(ag[j], a)
        if e > 1e-4:
            bad.append((n, keys[n], float(e), plan.desc[j].cpu().numpy().tolist()))
    return bad, plan


all_units = list(range(len(units)))
for spectral in (False, True):
    bad, plan = errs(all_units, spectral)
    print("spectral", spectral, "all units: bad", len(bad), "flags", plan.flags)
    for b in bad[:12]:
        print("   ", b)
    if bad:
        n = bad[0][0]
        for sel in ([n], all_units[:n + 1], [u for u in all_units if keys[u][4] < 0 or u == n]):
            b2, _ = errs(sel, spectral)
            print("   subset of", len(sel), "-> bad", [(x[0], x[2]) for x in b2][:6])
