"""A tiny synthetic stand-in for the data tree ``AudioGoalDataset`` reads (ss_baselines/savi/pretraining/audiogoal_dataset.py:
26-27: data/binaural_rirs/mp3d/<scene>/<angle>/<recv>_<src>.wav, data/sounds/semantic_splits/<split>/<category>.wav) plus the
scene graphs it is given (networkx graphs whose nodes carry 'point').  Deterministic from its arguments: the golden generator
(tests/golden/make_golden_dataset.py, which runs the REFERENCE's class on it) and the tests build the same tree."""
import os

import networkx as nx
import numpy as np
from scipy.io import wavfile

from oracle import ss_oracle as O

SR = 16000
SCENES = ("sceneA", "sceneB")
SPLIT = "train"
SOUNDS = {"chair": 3, "table": 4, "sofa": 5, "bed": 2}          # category -> seconds (multi-second clips: :126 needs >= 2 s)
CATEGORY_INDEX = {"chair": 0, "table": 1, "sofa": 5, "bed": 6}  # the reference's ids of these categories (mp3d_utils.py:33-55)
ANGLES = (0, 90, 180, 270)


def scene_graph(scene: str) -> nx.Graph:
    """two connected components (3 + 2 nodes) with node ids that are not 0..n-1, points (x, y, z) on a 1-m grid"""
    rng = np.random.default_rng(abs(hash_str(scene)) % (1 << 31))
    g = nx.Graph()
    ids = [3, 7, 12, 20, 21]
    for n in ids:
        g.add_node(n, point=(float(rng.integers(-5, 6)), 0.0, float(rng.integers(-5, 6))))
    g.add_edges_from([(3, 7), (7, 12), (20, 21)])
    return g


def hash_str(s: str) -> int:
    h = 0
    for ch in s:
        h = (h * 131 + ord(ch)) & 0x7FFFFFFF
    return h


def rir_length(scene: str, angle: int, r: int, s: int) -> int:
    """ragged lengths on both sides of one second: 0 (an empty file: the zero RIR), short, about 1 s, up to 1.6 s"""
    k = hash_str(f"{scene}/{angle}/{r}_{s}") % 11
    return [0, 2500, 9000, 15999, 16000, 16001, 17000, 20000, 23456, 25600, 12345][k]


def build(root: str):
    """writes the tree under `root` (used as the working directory: the reference's paths are relative) -> scene graphs"""
    graphs = {sc: scene_graph(sc) for sc in SCENES}
    snd_dir = os.path.join(root, "data", "sounds", "semantic_splits", SPLIT)
    os.makedirs(snd_dir, exist_ok=True)
    for i, (name, secs) in enumerate(sorted(SOUNDS.items())):
        clip = O.synth_sources(np.random.default_rng(100 + i), SR, k=1, seconds=secs)[0]
        wavfile.write(os.path.join(snd_dir, name + ".wav"), SR, clip.astype(np.float32))
    for sc in SCENES:
        g = graphs[sc]
        for comp in nx.connected_components(g):
            for s in comp:
                for r in comp:
                    for a in ANGLES:
                        d = os.path.join(root, "data", "binaural_rirs", "mp3d", sc, str(a))
                        os.makedirs(d, exist_ok=True)
                        L = rir_length(sc, a, r, s)
                        seed = hash_str(f"{sc}/{a}/{r}_{s}")
                        h = O.synth_rir(np.random.default_rng(seed), SR, length=L, n=1)[0] if L else np.zeros((2, 0), np.float32)
                        wavfile.write(os.path.join(d, f"{r}_{s}.wav"), SR, np.ascontiguousarray(h.T).astype(np.float32))
    return graphs
