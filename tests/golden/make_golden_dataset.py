#!/usr/bin/env python
"""tests/golden/dataset_vectors.npz: the REFERENCE's ``AudioGoalDataset`` run here on the synthetic tree of
tests/dataset_tree.py.  The class is extracted from the reference checkout with ``ast`` at generation time and executed whole
(``__init__``, ``_compute_goal_xy``, ``__getitem__``, ``compute_audiogoal``, ``compute_spectrogram``: ss_baselines/savi/
pretraining/audiogoal_dataset.py:21-155) with real networkx / scipy / torch and stand-ins only for what the image lacks:
``librosa.load`` (float32 16-kHz wavs: a plain read), ``librosa.stft`` / ``block_reduce`` (the oracle's restatements - recorded
in the npz meta; bound to the REAL libraries when they are importable), ``tqdm``.  ``CATEGORY_INDEX_MAPPING`` and ``to_tensor``
are likewise extracted from the reference's own modules.

Stored: the item list (files, goals) for ``random.seed(7)``, and - for ``random.seed(11)`` followed by item-by-item access in
order - every item's drawn second index, audiogoal (strided) and spectrogram.
Run from the repo root in the build container:  python tests/golden/make_golden_dataset.py"""
import ast
import json
import os
import random
import sys
import tempfile
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import ss_oracle as O  # noqa: E402
import dataset_tree as T  # noqa: E402

REF = os.environ.get("SS_REFERENCE", "/root/reference")


def node_source(path, name, kind):
    src = open(os.path.join(REF, path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, kind) and getattr(n, "name", None) == name)
    return textwrap.dedent("\n".join(src.splitlines()[node.lineno - 1: node.end_lineno]))


def assigned_literal(path, name):
    src = open(os.path.join(REF, path)).read()
    for n in ast.parse(src).body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in n.targets):
            return ast.literal_eval(n.value)
    raise KeyError(name)


def main():
    import networkx as nx
    import torch
    from scipy.io import wavfile
    from scipy.signal import fftconvolve
    used = {}
    try:
        import librosa
        stft, load = librosa.stft, librosa.load
        used["librosa"] = "real " + librosa.__version__
    except ImportError:
        stft = lambda sig, n_fft, hop_length, win_length: O.stft(sig)                          # noqa: E731
        load = lambda path, sr: (wavfile.read(path)[1].astype(np.float32), sr)                 # noqa: E731
        used["librosa"] = "absent: stft = oracle restatement, load = plain float32 wav read"
    try:
        from skimage.measure import block_reduce
        used["skimage"] = "real"
    except ImportError:
        block_reduce = lambda a, block_size, func: O.block_reduce_mean(a, block_size)          # noqa: E731
        used["skimage"] = "absent: block_reduce = oracle restatement"
    # os.listdir's order is the file system's: the item list the reference builds depends on it.  The fixture is generated - and
    # the tests run - with a SORTED listing so that it is the same on every box
    os_sorted = types.SimpleNamespace(listdir=lambda p_: sorted(os.listdir(p_)), path=os.path)
    ns = {"np": np, "os": os_sorted, "random": random, "nx": nx, "wavfile": wavfile, "fftconvolve": fftconvolve, "logging": __import__("logging"),
          "product": __import__("itertools").product, "copy": __import__("copy"), "pickle": __import__("pickle"),
          "tqdm": lambda x: x, "Dataset": object, "torch": torch,
          "librosa": types.SimpleNamespace(stft=stft, load=load), "block_reduce": block_reduce,
          "CATEGORY_INDEX_MAPPING": assigned_literal("soundspaces/mp3d_utils.py", "CATEGORY_INDEX_MAPPING")}
    exec(node_source("ss_baselines/common/utils.py", "to_tensor", ast.FunctionDef), ns)
    exec(node_source("ss_baselines/savi/pretraining/audiogoal_dataset.py", "AudioGoalDataset", ast.ClassDef), ns)
    RefDataset = ns["AudioGoalDataset"]
    assert all(ns["CATEGORY_INDEX_MAPPING"][k] == v for k, v in T.CATEGORY_INDEX.items())
    out, meta = {}, {"stand_ins": used}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        graphs = T.build(td)
        os.chdir(td)
        try:
            for polar in (False, True):
                random.seed(7)
                ds = RefDataset(scene_graphs=graphs, scenes=list(T.SCENES), split=T.SPLIT, use_polar_coordinates=polar, use_cache=False)
                tag = "polar" if polar else "xy"
                out[f"{tag}/goals"] = np.stack([g.numpy() for g in ds.goals])
                meta[f"{tag}/files"] = [list(f) for f in ds.files]
                meta[f"{tag}/goal_dtype"] = str(ds.goals[0].dtype)
            random.seed(7)
            ds = RefDataset(scene_graphs=graphs, scenes=list(T.SCENES), split=T.SPLIT, use_polar_coordinates=False, use_cache=True)
            meta["len"] = len(ds)
            # the drawn indices are not observable from outside: wrap random.randint for the recording
            drawn = []
            real_randint = random.randint

            def rec(a, b):
                v = real_randint(a, b)
                drawn.append(v)
                return v
            random.seed(11)
            random.randint = rec
            try:
                items = [ds[i] for i in range(len(ds))]
                again = ds[3]                                                  # use_cache: the SAME object, no new draw
            finally:
                random.randint = real_randint
            assert again is items[3] and len(drawn) == len(ds)
            out["items/index"] = np.asarray(drawn, np.int64)
            out["items/spectrogram"] = np.stack([it[0][0].numpy() for it in items]).astype(np.float32)
            meta["items/spectrogram_dtype"] = str(items[0][0][0].dtype)
            # audiogoals of a few items (full rate), recomputed with the recorded index
            for i in (0, 5, 17):
                random.randint = lambda a, b, i=i: drawn[i]
                try:
                    out[f"items/audiogoal_{i}"] = np.asarray(ds.compute_audiogoal(*ds.files[i]), np.float32)
                finally:
                    random.randint = real_randint
        finally:
            os.chdir(cwd)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "dataset_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", meta["len"], "items;", used)


if __name__ == "__main__":
    main()
