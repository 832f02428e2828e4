"""Drop-in for ``AudioGoalDataset`` (ss_baselines/savi/pretraining/audiogoal_dataset.py:21-155), the offline set the savi
belief predictor is pre-trained on (consumed at ss_baselines/savi/pretraining/audiogoal_trainer.py:53-66).

The reference renders every item on the CPU inside DataLoader workers: ``wavfile.read`` of the item's RIR file, two
``fftconvolve`` calls on a randomly drawn second of a multi-second clip, ``librosa.stft`` + ``block_reduce`` + ``log1p``
(:114-155).  Here the same items come from the HIP path:

* same constructor, same ``files`` / ``goals`` lists for the same state of Python's ``random`` module (the reference draws
  with ``random.shuffle`` / ``random.choice`` in a fixed order, :41-45), same ``_compute_goal_xy`` (:71-95), ``__len__``,
  ``audio_length``, ``use_cache`` semantics (:100-112);
* ``__getitem__(i)`` -> ``([spectrogram [65, 26, 2]], goal [3])`` as the reference returns it (one launch per item: for
  compatibility, e.g. ``DataLoader(dataset, num_workers=0)``), the second index drawn with ``random.randint(0, n - 2)``
  from the module-level generator exactly like :126;
* ``loader(batch_size, ...)``: what replaces ``DataLoader(dataset, batch_size=1024, num_workers=8)`` at
  audiogoal_trainer.py:61 - an iterable of ``(inputs, gts)`` with ``inputs = [spectrogram [B, 65, 26, 2]]`` (device tensor)
  and ``gts [B, 3]``, i.e. what the default collate function makes of the reference's items.  A whole mini-batch is ONE
  step of the engine: the batch's RIR files go through ``RirStore.load_files`` (the library's wav reader, one scatter launch
  per 256 rows), the unit columns {sound, t0, rir} through the C++ planner, the items through one launch.  The draws are
  reproducible from ``seed``: batch iteration with ``seed=S`` draws the same second per item as item-by-item access after
  ``random.seed(S)``.

Windowing (:126-138): ``index * sr - len(rir) < 0`` -> the clip up to second ``index`` convolved in full, second ``index``
kept (t0 = index * sr); else the slice starts ONE SAMPLE EARLIER than the simulator's and the last sample is dropped
(t0 = index * sr - 1: ``planning.window_start_savi_dataset``).  Unreadable (ValueError) and empty RIR files are the zero
RIR (:117-123).

What stays on the CPU, as in the reference: the scene graphs (networkx), the goal labels, ``librosa.load`` of the split's
sounds (once, :64-69)."""
from __future__ import annotations

import os
import random
from itertools import product
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import planning as P

try:                                                            # the real base class when torch's data package is there
    from torch.utils.data import Dataset as _TorchDataset
except Exception:                                               # pragma: no cover
    class _TorchDataset:                                        # type: ignore
        pass


def _category_index_mapping() -> Dict[str, int]:
    """``soundspaces.mp3d_utils.CATEGORY_INDEX_MAPPING`` when the reference package is importable (it is the label
    definition of the pre-training task); callers without it pass ``category_index=``."""
    from soundspaces.mp3d_utils import CATEGORY_INDEX_MAPPING   # noqa: WPS433 (optional dependency)
    return dict(CATEGORY_INDEX_MAPPING)


def default_sound_loader(path: str, sr: int) -> np.ndarray:
    """``librosa.load(path, sr=sr)[0]`` (:66-68) when librosa is installed; otherwise float / PCM wav files through scipy,
    mono-mixed, polyphase-resampled when the file's rate differs (NOT librosa's resampler: install librosa for parity on
    resampled clips)."""
    try:
        import librosa
        return librosa.load(path, sr=sr)[0]
    except ImportError:
        from scipy.io import wavfile
        fs, x = wavfile.read(path)
        if x.dtype.kind == "i":
            x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
        elif x.dtype.kind == "u":
            x = (x.astype(np.float32) - 128.0) / 128.0
        x = np.asarray(x, np.float32)
        if x.ndim == 2:
            x = x.mean(axis=1)
        if fs != sr:
            from scipy.signal import resample_poly
            g = np.gcd(int(fs), int(sr))
            x = resample_poly(x, sr // g, fs // g).astype(np.float32)
        return np.ascontiguousarray(x, np.float32)


def _goal_tensor(index: int, goal_xy: torch.Tensor) -> torch.Tensor:
    # the reference builds the label as to_tensor(np.zeros(3)) = a float64 tensor and assigns into it (:52-55)
    goal = torch.from_numpy(np.zeros(3))
    goal[0] = index
    goal[1:] = goal_xy
    return goal


class AudioGoalDataset(_TorchDataset):
    """See the module docstring.  Positional arguments are the reference's (:22); everything behind ``*`` is what the
    reference hard-codes or imports (directories, label map, sound loader) plus the engine's knobs."""

    def __init__(self, scene_graphs, scenes, split, use_polar_coordinates=False, use_cache=False, filter_rule='', *,
                 binaural_rir_dir: str = 'data/binaural_rirs/mp3d', source_sound_dir: Optional[str] = None,
                 category_index: Optional[Dict[str, int]] = None, sound_loader: Callable[[str, int], np.ndarray] = default_sound_loader,
                 device="cuda", engine=None, rir_slots: int = 4096, pairs_per_scene: int = 50000, progress=None):
        self.use_cache = use_cache
        self.files: List[Tuple[str, str]] = []
        self.goals: List[torch.Tensor] = []
        self.binaural_rir_dir = binaural_rir_dir
        self.source_sound_dir = source_sound_dir if source_sound_dir is not None else f'data/sounds/semantic_splits/{split}'
        self.source_sound_dict: Dict[str, np.ndarray] = {}
        self.rir_sampling_rate = 16000
        self._sound_loader = sound_loader
        self.device = torch.device(device)
        labels = category_index if category_index is not None else _category_index_mapping()
        sound_files = os.listdir(self.source_sound_dir)
        import networkx as nx
        # the draws below are the reference's, in the reference's order (:36-45): one shuffle of the scene's (source,
        # receiver) pairs, then per kept pair one choice of a sound file and one of an azimuth - so a caller that seeds
        # `random` gets the reference's item list
        for scene in (progress(scenes) if progress is not None else scenes):
            graph = scene_graphs[scene]
            sr_pairs = []
            for component in nx.connected_components(graph):
                sr_pairs += list(product(component, component))
            random.shuffle(sr_pairs)
            for s, r in sr_pairs[:pairs_per_scene]:
                sound_file = random.choice(sound_files)
                angle = random.choice([0, 90, 180, 270])
                self.files.append((os.path.join(self.binaural_rir_dir, scene, str(angle), f"{r}_{s}.wav"), sound_file))
                ps, pr = graph.nodes[s]['point'], graph.nodes[r]['point']
                xy = self._compute_goal_xy(ps[0] - pr[0], ps[2] - pr[2], angle, use_polar_coordinates)
                self.goals.append(_goal_tensor(labels[sound_file[:-4]], xy))
        self.data: List = [None] * len(self.goals)
        self._engine = engine
        self._rir_slots = int(rir_slots)
        self._sound_ids: Dict[str, int] = {}
        self.load_source_sounds()

    # ---- the reference's small methods ---------------------------------------------------------------------------------
    def audio_length(self, sound):
        return self.source_sound_dict[sound].shape[0] // self.rir_sampling_rate

    def load_source_sounds(self):
        for sound_file in os.listdir(self.source_sound_dir):
            self.source_sound_dict[sound_file] = self._sound_loader(os.path.join(self.source_sound_dir, sound_file),
                                                                    self.rir_sampling_rate)

    @staticmethod
    def _compute_goal_xy(delta_x, delta_y, angle, use_polar_coordinates):
        """-Y is forward, X is rightward, agent faces -Y (:71-95): the world-frame offset rotated by the agent's azimuth."""
        rot = {0: (delta_x, delta_y), 90: (delta_y, -delta_x), 180: (-delta_x, -delta_y)}
        x, y = rot.get(angle, (-delta_y, delta_x))
        if use_polar_coordinates:
            return torch.tensor([np.arctan2(y, x), np.linalg.norm([y, x])], dtype=torch.float)
        return torch.tensor([x, y], dtype=torch.float)

    def __len__(self):
        return len(self.files)

    # ---- rendering -------------------------------------------------------------------------------------------------------
    @property
    def engine(self):
        """The ``AudioEngine`` behind the items (built on first use: whole RIR rows - the clips are multi-second -, one bank
        slot per distinct RIR file, LRU beyond ``rir_slots``)."""
        if self._engine is None:
            from .renderer import AudioEngine
            # (mini-batches are hundreds of items: the time-domain rows; no block spectra to keep in step with the loads)
            self._engine = AudioEngine(self.rir_sampling_rate, device=self.device, rir_slots=self._rir_slots, rir_spectral=False)
        if not self._sound_ids:
            for name in sorted(self.source_sound_dict):                   # (sorted: ids do not depend on os.listdir's order)
                self._sound_ids[name] = self._engine.source_id("savi-pretraining/" + name, self.source_sound_dict[name])
        return self._engine

    def draw_index(self, sound_file: str, rng=random) -> int:
        """``random.randint(0, self.audio_length(sound_file) - 2)`` (:126; a 1-s clip raises ValueError there as here)"""
        return rng.randint(0, self.audio_length(sound_file) - 2)

    def unit_columns(self, items: Sequence[int], indices: Sequence[int]) -> Dict[str, np.ndarray]:
        """{sound, t0, rir} of the given items with the given second indices: their RIR files are made resident (one
        ``load_files`` call: the library's reader + scatter; unreadable / empty files = the zero RIR, :117-123)."""
        eng = self.engine
        paths = [self.files[i][0] for i in items]
        slots = np.asarray(eng.store.load_files(paths, paths), np.int64)
        sr = self.rir_sampling_rate
        lens = np.asarray(eng.store.host_len)[slots].astype(np.int64)
        # a zero RIR is np.zeros((sr, 2)) in the reference (:120, :123): its LENGTH picks the branch, its samples are zeros -
        # either branch renders zeros, so the stored (empty) row's length need not be patched for the output to match
        idx = np.asarray(indices, np.int64)
        t0 = np.where(idx * sr - lens < 0, idx * sr, idx * sr - 1)            # planning.window_start_savi_dataset, vectorised
        sound = np.asarray([self._sound_ids[self.files[i][1]] for i in items], np.int64)
        return dict(sound=sound, t0=t0, rir=slots)

    def render(self, items: Sequence[int], indices: Sequence[int], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """spectrograms [len(items), 65, 26, 2] of the items at the given second indices: ONE step of the engine"""
        eng = self.engine
        if out is None:
            out = torch.empty((len(items),) + P.spectrogram_shape(self.rir_sampling_rate), dtype=torch.float32, device=self.device)
        room = int(getattr(eng.store, "slots", len(items)))
        lo = 0
        while lo < len(items):                                    # (a mini-batch with more distinct RIR files than the store has
            seen, hi = set(), lo                                  #  entries is rendered in as many launches as it takes)
            while hi < len(items) and (len(seen) < room or self.files[items[hi]][0] in seen):
                seen.add(self.files[items[hi]][0])
                hi += 1
            eng.observe_columns(self.unit_columns(items[lo:hi], indices[lo:hi]), spectrogram_out=out[lo:hi])   # (load_files opens the store's batch)
            lo = hi
        return out

    def __getitem__(self, item):
        if self.use_cache and self.data[item] is not None:
            return self.data[item]
        index = self.draw_index(self.files[item][1])
        spectrogram = self.render([item], [index])[0]
        inputs_outputs = ([spectrogram], self.goals[item])
        if self.use_cache:
            self.data[item] = inputs_outputs
        return inputs_outputs

    def loader(self, batch_size: int = 1024, shuffle: bool = False, drop_last: bool = False, seed: Optional[int] = None,
               order: Optional[Sequence[int]] = None) -> "AudioGoalBatches":
        """The batched replacement of ``DataLoader(self, batch_size=..., num_workers=8)`` (audiogoal_trainer.py:61-67)."""
        return AudioGoalBatches(self, batch_size, shuffle, drop_last, seed, order)


class AudioGoalBatches:
    """Iterable of ``(inputs, gts)`` mini-batches of an ``AudioGoalDataset``: ``inputs = [spectrogram [B, 65, 26, 2]]`` on
    the dataset's device, ``gts [B, 3]`` (float64 like the reference's labels; the trainer casts, :104).  Every pass draws
    fresh second indices (the reference redraws per ``__getitem__`` unless ``use_cache``); with ``seed`` the draws - and the
    shuffle - of pass k come from ``random.Random(seed + k)``, in item order, one ``randint`` per item exactly as item-by-item
    access would make them.  ``use_cache=True`` keeps the first rendering of every item (:100-110) - as one device tensor per
    batch position, not per-item objects."""

    def __init__(self, dataset: AudioGoalDataset, batch_size: int, shuffle: bool, drop_last: bool, seed: Optional[int],
                 order: Optional[Sequence[int]]):
        self.ds, self.batch_size, self.shuffle, self.drop_last, self.seed = dataset, int(batch_size), shuffle, drop_last, seed
        self.order = None if order is None else list(order)
        self._pass = 0
        self._goals = torch.stack(dataset.goals) if len(dataset.goals) else torch.zeros((0, 3), dtype=torch.float64)

    def __len__(self) -> int:
        n = len(self.ds) if self.order is None else len(self.order)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self) -> Iterator[Tuple[List[torch.Tensor], torch.Tensor]]:
        ds = self.ds
        rng = random.Random(self.seed + self._pass) if self.seed is not None else random
        self._pass += 1
        order = list(range(len(ds))) if self.order is None else list(self.order)
        if self.shuffle:
            rng.shuffle(order)
        for lo in range(0, len(order), self.batch_size):
            items = order[lo:lo + self.batch_size]
            if self.drop_last and len(items) < self.batch_size:
                break
            cached = ds.use_cache and all(ds.data[i] is not None for i in items)
            if cached:
                sg = torch.stack([ds.data[i][0][0] for i in items])
            else:
                need = [i for i in items if not (ds.use_cache and ds.data[i] is not None)]
                idx = [ds.draw_index(ds.files[i][1], rng) for i in need]
                fresh = ds.render(need, idx)
                if len(need) == len(items):
                    sg = fresh
                else:
                    pos = {i: k for k, i in enumerate(need)}
                    sg = torch.stack([fresh[pos[i]] if i in pos else ds.data[i][0][0] for i in items])
                if ds.use_cache:
                    for k, i in enumerate(need):
                        ds.data[i] = ([fresh[k]], ds.goals[i])
            yield [sg], self._goals[items]


class AudioGoalBatcher:
    """Lower-level helper kept from round 1: (sound, RIR bank row, second index) triples -> one launch on a
    ``BatchedAudioRenderer`` whose banks the caller filled."""

    def __init__(self, renderer):
        self.r = renderer

    def requests(self, sound_ids: Sequence[int], rir_ids: Sequence[int], rir_lens: Sequence[int], indices: Sequence[int]):
        from .renderer import UnitRequest
        sr = self.r.sr
        return [UnitRequest(int(s), P.window_start_savi_dataset(int(L), sr, int(i)), int(h))
                for s, h, L, i in zip(sound_ids, rir_ids, rir_lens, indices)]

    def spectrograms(self, sound_ids, rir_ids, rir_lens, indices, want_audiogoal: bool = False):
        """-> (audiogoal [N,2,sr] or None, spectrogram [N,65,T4,2]) on the device."""
        return self.r.render(self.r.plan(self.requests(sound_ids, rir_ids, rir_lens, indices)), want_audiogoal=want_audiogoal)

    def draw_indices(self, rng: np.random.Generator, sound_ids: Sequence[int]) -> np.ndarray:
        lens = np.array([self.r.sources.lengths[s] // self.r.sr for s in sound_ids])
        return np.array([rng.integers(0, max(1, n - 1)) for n in lens])
