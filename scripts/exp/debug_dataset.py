import os, random, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sound-spaces_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, tempfile
import dataset_tree as T
from oracle import ss_oracle as O
from ss_amd.datasets import AudioGoalDataset
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "dataset_vectors.npz"))
real = os.listdir
os.listdir = lambda p=".": sorted(real(p))
td = tempfile.mkdtemp()
graphs = T.build(td)
def make(**kw):
    random.seed(7)
    return AudioGoalDataset(graphs, list(T.SCENES), T.SPLIT, binaural_rir_dir=os.path.join(td, "data", "binaural_rirs", "mp3d"),
                            source_sound_dir=os.path.join(td, "data", "sounds", "semantic_splits", T.SPLIT),
                            category_index=T.CATEGORY_INDEX, device="cuda:0", **kw)
ref = GOLD["items/spectrogram"]; idx = GOLD["items/index"]
for bs in (1, 2, 7, 26):
    ds = make()
    got = torch.cat([i[0] for i, _ in ds.loader(batch_size=bs, seed=11)]).cpu().numpy()
    errs = [O.relerr(got[i], ref[i]) for i in range(26)]
    eng = ds.engine
    print("bs", bs, "cap", eng.store.cap, "bad", [(i, round(float(e), 3), int(eng.store.host_len[eng.store._slot_of[ds.files[i][0]]]), int(idx[i])) for i, e in enumerate(errs) if e > 1e-4])
# the same items in one launch, but files loaded one by one first
ds = make()
for i in range(26):
    ds.engine.store.load_files([ds.files[i][0]], [ds.files[i][0]])
got = ds.render(list(range(26)), idx).cpu().numpy()
print("preloaded singly, one launch: bad", [i for i in range(26) if O.relerr(got[i], ref[i]) > 1e-4], "cap", ds.engine.store.cap)
ds = make()
ds.engine.store.load_files([f for f, _ in ds.files], [f for f, _ in ds.files])
got = np.concatenate([ds.render([i], [idx[i]]).cpu().numpy() for i in range(26)])
print("preloaded in one call, single launches: bad", [i for i in range(26) if O.relerr(got[i], ref[i]) > 1e-4])
