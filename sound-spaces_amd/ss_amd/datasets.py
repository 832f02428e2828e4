"""GPU batch generator with the semantics of ``AudioGoalDataset`` (ss_baselines/savi/pretraining/audiogoal_dataset.py):
the offline belief-predictor pretraining set re-implements ``_compute_audiogoal`` + ``compute_spectrogram`` per item on
the CPU inside a DataLoader; here a whole batch of (rir, sound, second-index) items is one launch."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import planning as P
from .renderer import BatchedAudioRenderer, UnitRequest


class AudioGoalBatcher:
    def __init__(self, renderer: BatchedAudioRenderer):
        self.r = renderer

    def requests(self, sound_ids: Sequence[int], rir_ids: Sequence[int], rir_lens: Sequence[int],
                 indices: Sequence[int]):
        """indices[k] = the second index drawn by ``random.randint(0, audio_length - 2)`` (:124); the steady branch
        starts one sample earlier than the simulator's and drops the last sample (:134-138)."""
        sr = self.r.sr
        return [UnitRequest(int(s), P.window_start_savi_dataset(int(L), sr, int(i)), int(h))
                for s, h, L, i in zip(sound_ids, rir_ids, rir_lens, indices)]

    def spectrograms(self, sound_ids, rir_ids, rir_lens, indices, want_audiogoal: bool = False):
        """-> (audiogoal [N,2,sr] or None, spectrogram [N,65,T4,2]) on the device."""
        return self.r.render(self.r.plan(self.requests(sound_ids, rir_ids, rir_lens, indices)),
                             want_audiogoal=want_audiogoal)

    def draw_indices(self, rng: np.random.Generator, sound_ids: Sequence[int]) -> np.ndarray:
        """random.randint(0, audio_length - 2) per item (inclusive upper bound, :124)."""
        lens = np.array([self.r.sources.lengths[s] // self.r.sr for s in sound_ids])
        return np.array([rng.integers(0, max(1, n - 1)) for n in lens])
