"""``ss_amd.datasets.AudioGoalDataset`` against the REFERENCE's class run on the same synthetic tree
(tests/golden/make_golden_dataset.py executes ss_baselines/savi/pretraining/audiogoal_dataset.py:21-155 as it stands ->
tests/golden/dataset_vectors.npz): same item list and labels for the same state of ``random``, same second-index draws item by
item AND through the batched loader, the oracle pinned to the reference-run audiogoals; on the GPU (-m gpu) every item's
spectrogram within 1e-4 of the reference-run one, per item and per mini-batch."""
import json
import os
import random

import numpy as np
import pytest
import torch

import dataset_tree as T
from oracle import ss_oracle as O
from ss_amd import planning as P
from ss_amd.datasets import AudioGoalDataset

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_vectors.npz"))
META = json.loads(bytes(GOLD["meta"]).decode())


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = tmp_path_factory.mktemp("savi_tree")
    graphs = T.build(str(root))
    return str(root), graphs


@pytest.fixture(autouse=True)
def sorted_listdir(monkeypatch):
    """the fixture was generated with a sorted directory listing (os.listdir's order is the file system's)"""
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p_=".": sorted(real(p_)))


def make(tree, device="cpu", **kw):
    root, graphs = tree
    kw.setdefault("category_index", T.CATEGORY_INDEX)
    return AudioGoalDataset(graphs, list(T.SCENES), T.SPLIT, binaural_rir_dir=os.path.join(root, "data", "binaural_rirs", "mp3d"),
                            source_sound_dir=os.path.join(root, "data", "sounds", "semantic_splits", T.SPLIT),
                            device=device, **kw)


def rel_files(ds, root):
    return [[os.path.relpath(f, root), s] for f, s in ds.files]


@pytest.mark.parametrize("polar", [False, True])
def test_item_list_and_goal_labels_equal_the_reference_run(tree, polar):
    random.seed(7)
    ds = make(tree, use_polar_coordinates=polar)
    tag = "polar" if polar else "xy"
    assert len(ds) == META["len"] == 26
    assert rel_files(ds, tree[0]) == META[f"{tag}/files"]
    goals = torch.stack(ds.goals)
    assert str(goals.dtype) == META[f"{tag}/goal_dtype"] == "torch.float64"
    assert np.array_equal(goals.numpy(), GOLD[f"{tag}/goals"])          # bit-exact: same float32 arithmetic, same casts
    assert ds.audio_length("sofa.wav") == 5 and ds.rir_sampling_rate == 16000
    assert sorted(ds.source_sound_dict) == sorted(n + ".wav" for n in T.SOUNDS)


def test_goal_xy_rotations_follow_the_reference_table():
    # audiogoal_dataset.py:71-95: -Y forward, X rightward
    f = AudioGoalDataset._compute_goal_xy
    assert f(2.0, 3.0, 0, False).tolist() == [2.0, 3.0]
    assert f(2.0, 3.0, 90, False).tolist() == [3.0, -2.0]
    assert f(2.0, 3.0, 180, False).tolist() == [-2.0, -3.0]
    assert f(2.0, 3.0, 270, False).tolist() == [-3.0, 2.0]
    th, d = f(2.0, 3.0, 90, True).tolist()
    assert abs(th - np.arctan2(-2.0, 3.0)) < 1e-6 and abs(d - np.hypot(2.0, 3.0)) < 1e-6


def test_second_index_draws_item_by_item_and_batched_equal_the_reference_run(tree, monkeypatch):
    random.seed(7)
    ds = make(tree)
    seen = []

    def fake_render(items, indices, out=None):
        seen.append((list(items), [int(i) for i in indices]))
        return torch.zeros((len(items),) + P.spectrogram_shape(16000))
    monkeypatch.setattr(ds, "render", fake_render)
    random.seed(11)
    for i in range(len(ds)):
        sg, goal = ds[i]
        assert isinstance(sg, list) and tuple(sg[0].shape) == (65, 26, 2) and goal is ds.goals[i]
    assert [idx[0] for _, idx in seen] == GOLD["items/index"].tolist()
    # the batched loader, seeded: the same draws in item order, a whole mini-batch per render() call
    seen.clear()
    batches = list(ds.loader(batch_size=8, seed=11))
    assert [len(it) for it, _ in seen] == [8, 8, 8, 2] and len(ds.loader(batch_size=8)) == 4 and len(ds.loader(8, drop_last=True)) == 3
    assert sum((idx for _, idx in seen), []) == GOLD["items/index"].tolist()
    assert sum((it for it, _ in seen), []) == list(range(26))
    inputs, gts = batches[1]
    assert isinstance(inputs, list) and tuple(inputs[0].shape) == (8, 65, 26, 2) and tuple(gts.shape) == (8, 3)
    assert np.array_equal(gts.numpy(), GOLD["xy/goals"][8:16])
    # a second pass draws anew (seed + 1), a shuffled pass permutes the items and keeps labels attached
    seen.clear()
    list(ds.loader(batch_size=26, seed=11, shuffle=True))
    assert sorted(seen[0][0]) == list(range(26)) and seen[0][0] != list(range(26))


def test_use_cache_keeps_the_first_rendering(tree, monkeypatch):
    random.seed(7)
    ds = make(tree, use_cache=True)
    calls = []
    monkeypatch.setattr(ds, "render", lambda items, indices, out=None: calls.append(list(items)) or torch.full((len(items), 65, 26, 2), float(len(calls))))
    a = ds[3]
    assert ds[3] is a and calls == [[3]]                    # audiogoal_dataset.py:100-110: the cached tuple itself, no new draw
    got = list(ds.loader(batch_size=4, order=[2, 3, 4, 5]))
    assert calls[-1] == [2, 4, 5] and float(got[0][0][0][1, 0, 0, 0]) == 1.0 and float(got[0][0][0][0, 0, 0, 0]) == 2.0
    list(ds.loader(batch_size=4, order=[2, 3, 4, 5]))
    assert len(calls) == 2                                  # everything cached: nothing rendered


def test_unit_columns_and_the_oracle_against_the_reference_run_audiogoals(tree):
    """CPU store: the batch's files become resident through load_files, t0 follows audiogoal_dataset.py:127-138; the oracle's
    restatement of compute_audiogoal reproduces the reference-run waveforms from the same files."""
    from scipy.io import wavfile
    from fakes import OracleColumnEngine
    random.seed(7)
    ds = make(tree, engine=OracleColumnEngine(16000, slots=64))      # a real RirStore on the CPU; the arithmetic is the oracle's
    idx = GOLD["items/index"]
    cols = ds.unit_columns(list(range(26)), idx)
    eng = ds.engine
    assert eng.store.truncate_to is None                   # multi-second clips: whole RIR rows
    for i in range(26):
        path, snd = ds.files[i]
        _, h = wavfile.read(path)
        L = h.shape[0]
        slot = int(cols["rir"][i])
        assert int(eng.store.host_len[slot]) == L
        row = eng.store.bank.data[slot, :, :L].numpy()
        assert np.array_equal(row, h.T)
        Lref = L if L else 16000                            # the zero RIR of an empty file (:121-123)
        assert int(cols["t0"][i]) == P.window_start_savi_dataset(L, 16000, int(idx[i])) or L == 0
        if L == 0:
            assert int(cols["t0"][i]) in (int(idx[i]) * 16000, int(idx[i]) * 16000 - 1)
        assert eng.sources[int(cols["sound"][i])].shape[0] == ds.source_sound_dict[snd].shape[0]
    for i in (0, 5, 17):
        path, snd = ds.files[i]
        _, h = wavfile.read(path)
        ref = GOLD[f"items/audiogoal_{i}"]
        got = O.compute_audiogoal_savi_dataset(ds.source_sound_dict[snd], h if h.shape[0] else O.zero_rir(16000), 16000, int(idx[i]))
        assert O.relerr(got, ref) < 1e-6
        assert O.relerr(O.compute_spectrogram(ref), GOLD["items/spectrogram"][i]) < 1e-6
    # ... and the whole host side end to end (files -> store rows -> columns -> one step per mini-batch) with the oracle standing
    # in for the kernels: every item of the seeded loader against the reference-run spectrograms
    got = torch.cat([inputs[0] for inputs, _ in ds.loader(batch_size=9, seed=11)]).numpy()
    assert max(O.relerr(got[i], GOLD["items/spectrogram"][i]) for i in range(26)) < 1e-5
    # a store of 8 entries under a mini-batch of 26 items (24 distinct files): rendered in several steps, same numbers
    random.seed(7)
    small = OracleColumnEngine(16000, slots=8)
    ds_s = make(tree, engine=small)
    got_s = torch.cat([inputs[0] for inputs, _ in ds_s.loader(batch_size=26, seed=11)]).numpy()
    assert small.column_calls >= 3 and max(O.relerr(got_s[i], GOLD["items/spectrogram"][i]) for i in range(26)) < 1e-5


def test_missing_labels_and_one_second_clips_fail_like_the_reference(tree):
    random.seed(7)
    with pytest.raises(KeyError):
        make(tree, category_index={"chair": 0})            # CATEGORY_INDEX_MAPPING[sound_file[:-4]] (:43)
    ds = make(tree)
    ds.source_sound_dict["chair.wav"] = ds.source_sound_dict["chair.wav"][:16000]
    with pytest.raises(ValueError):
        ds.draw_index("chair.wav")                         # random.randint(0, -1) (:126)


@pytest.mark.gpu
def test_items_and_batches_on_the_gpu_equal_the_reference_run(tree):
    random.seed(7)
    ds = make(tree, device="cuda:0", use_cache=True)
    ref = GOLD["items/spectrogram"]
    random.seed(11)
    worst = 0.0
    for i in range(len(ds)):
        sg, goal = ds[i]
        assert sg[0].is_cuda and tuple(sg[0].shape) == (65, 26, 2)
        worst = max(worst, O.relerr(sg[0].cpu().numpy(), ref[i]))
    assert worst < 1e-4, worst
    assert ds[4] is ds[4]
    random.seed(7)                                          # (the item list is drawn from `random` at construction)
    ds2 = make(tree, device="cuda:0")
    got = torch.cat([inputs[0] for inputs, _ in ds2.loader(batch_size=7, seed=11)]).cpu().numpy()
    assert got.shape == ref.shape
    assert max(O.relerr(got[i], ref[i]) for i in range(26)) < 1e-4
    # an item whose RIR file is empty is the zero RIR: exact zeros (:121-123)
    empties = [i for i, (f, _) in enumerate(ds2.files) if os.path.getsize(f) <= 64]
    assert empties and all(not got[i].any() for i in empties)
    # a store smaller than the data set: rows are evicted and reloaded, same numbers
    random.seed(7)
    ds3 = make(tree, device="cuda:0", rir_slots=8)
    got3 = torch.cat([inputs[0] for inputs, _ in ds3.loader(batch_size=5, seed=11)]).cpu().numpy()
    assert np.array_equal(got3, got) or max(O.relerr(got3[i], ref[i]) for i in range(26)) < 1e-4
    # ... and a mini-batch with more distinct files than the store has entries: several launches, same numbers
    random.seed(7)
    ds4 = make(tree, device="cuda:0", rir_slots=8)
    got4 = torch.cat([inputs[0] for inputs, _ in ds4.loader(batch_size=26, seed=11)]).cpu().numpy()
    assert max(O.relerr(got4[i], ref[i]) for i in range(26)) < 1e-4
