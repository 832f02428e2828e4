"""Test doubles: a stand-in SoundSpacesSim exposing the state the audio path reads, and an oracle-backed engine
with the interface of ss_amd.renderer.AudioEngine (CPU tensors).  Used ONLY to test host-side logic (caches,
descriptor planning, plugin API) without a GPU; the product engine is the HIP one."""
import types

import numpy as np
import torch

from oracle import ss_oracle as O


class NS(types.SimpleNamespace):
    pass


class FakeSim:
    """The attributes of soundspaces.simulator.SoundSpacesSim that _compute_audiogoal and the caches touch."""

    def __init__(self, sr, sounds, rirs, has_distractor=False):
        self.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=has_distractor),
                         USE_RENDERED_OBSERVATIONS=True)
        self._source_sound_dict = dict(sounds)
        self.rir_files = rirs                       # path -> [L,2] array / None
        self.binaural_rir_dir = "rirs/replica/apartment_0"
        self._current_sound = next(iter(sounds))
        self._episode_step_count = 0
        self._duration = 500
        self._receiver_position_index = 3
        self._source_position_index = 7
        self._distractor_position_index = 11
        self._current_distractor_sound = None
        self._rotation_angle = 270
        self._audio_index = 0
        self._audiogoal_cache = dict()
        self._spectrogram_cache = dict()

    @property
    def azimuth_angle(self):                         # simulator.py:568-573
        return -(self._rotation_angle + 0) % 360

    @property
    def current_source_sound(self):
        return self._source_sound_dict[self._current_sound]

    @property
    def _audio_length(self):
        return self.current_source_sound.shape[0] // self.config.AUDIO.RIR_SAMPLING_RATE

    def reader(self, path):
        return self.rir_files.get(path)


class OracleEngine:
    def __init__(self, sr):
        self.sr = sr
        self.sources, self.names, self.rirs, self.keys = [], {}, [], {}
        self.calls = 0

    def source_id(self, name, clip):
        if name not in self.names:
            self.names[name] = len(self.sources)
            self.sources.append(np.asarray(clip, np.float32))
        return self.names[name]

    def rir_slot(self, key, loader, refresh=False):
        if key in self.keys and not refresh:
            return self.keys[key]
        r = loader()
        r = O.zero_rir(self.sr) if r is None else np.asarray(r, np.float32)
        if key in self.keys:
            self.rirs[self.keys[key]] = r
            return self.keys[key]
        self.keys[key] = len(self.rirs)
        self.rirs.append(r)
        return self.keys[key]

    def observe(self, units, want_audiogoal=False, want_spectrogram=True, spectrogram_out=None, audiogoal_out=None):
        self.calls += 1
        sr, ag, sg = self.sr, [], []
        for u in units:
            if u.silent or u.rir < 0:
                a = np.zeros((2, sr), np.float32)
            else:
                a = O.conv_window_fft(self.sources[u.sound], self.rirs[u.rir], u.t0, sr)
                if u.dis_rir >= 0:
                    a = a + O.conv_window_fft(self.sources[u.dis_sound], self.rirs[u.dis_rir], 0, sr)
            ag.append(a.astype(np.float32))
            sg.append(O.compute_spectrogram(a.astype(np.float32)).astype(np.float32))
        out = {}

        def place(dst, arr):                     # like the real engine: fill the caller's tensor when one is given
            t = torch.from_numpy(np.stack(arr))
            if dst is None:
                return t
            dst.copy_(t)
            return dst
        if want_audiogoal or audiogoal_out is not None or not want_spectrogram:
            out["audiogoal"] = place(audiogoal_out, ag)
        if want_spectrogram:
            out["spectrogram"] = place(spectrogram_out, sg)
        return out
