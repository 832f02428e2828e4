"""Rebuild the seeded inputs of tests/golden/reference_vectors.npz (see make_golden.py)."""
import json
import os

import numpy as np

from oracle import ss_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
_G = None


def golden():
    global _G
    if _G is None:
        z = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))
        meta = json.loads(bytes(z["meta"]).decode())
        _G = (z, meta["cases"], meta["params"])
    return _G


def case_inputs(name):
    """-> dict(sr, source [S] f32, rir [L,2] f32 wav layout, + optional distractor, last_rir ...)"""
    z, _, params = golden()
    p = params[name]
    sr = p["sr"]
    d = dict(p)
    if p.get("src") == "singing_16k":
        d["source"] = z["singing_16k"]
    elif "src_seed" in p:
        if "seconds" in p:
            d["source"] = O.synth_sources(np.random.default_rng(p["src_seed"]), sr, k=1, seconds=p["seconds"])[0]
        else:
            bank = O.synth_sources(np.random.default_rng(p["src_seed"]), sr, k=p.get("src_k", 1), seconds=1)
            d["source"] = bank[p.get("src_sel", 0)]
            if "dis_sel" in p:
                d["distractor"] = bank[p["dis_sel"]]
    if "rir_seed" in p:
        L = p.get("rir_len", sr)
        bank = O.synth_rir(np.random.default_rng(p["rir_seed"]), sr, length=None if L == sr else L, n=p.get("rir_n", 1))
        d["rir"] = np.ascontiguousarray(bank[p.get("rir_sel", 0)].T)
        if "dis_rir_sel" in p:
            d["distractor_rir"] = np.ascontiguousarray(bank[p["dis_rir_sel"]].T)
        if "rir_decay" in p:
            bank = bank * np.exp(-np.arange(L) / float(p["rir_decay"]))[None, None, :].astype(np.float32)
            d["rir"] = np.ascontiguousarray(bank[p.get("rir_sel", 0)].T)
        if "last_rir_sel" in p:
            d["last_rir"] = np.ascontiguousarray(bank[p["last_rir_sel"]].T)
        if "last_rir_seed" in p:
            Ll = p["last_rir_len"]
            lb = O.synth_rir(np.random.default_rng(p["last_rir_seed"]), sr, length=Ll, n=p["last_rir_n"])
            lb = lb * np.exp(-np.arange(Ll) / float(p["rir_decay_last"]))[None, None, :].astype(np.float32)
            d["last_rir"] = np.ascontiguousarray(lb[p["last_rir_pick"]].T)
    return d


def case_outputs(name):
    z, _, params = golden()
    return z[name + "/audiogoal"], z[name + "/spectrogram"], params[name]["audiogoal_stride"]
