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

    def use_live_rirs(self, rir_fn):
        """simulator.py:625-626: USE_RENDERED_OBSERVATIONS False - the RIR of the current pose comes from the habitat_sim
        audio sensor ([2][L] nested lists) instead of a wav file; `rir_fn(call number)` plays the ray tracer."""
        self.config.USE_RENDERED_OBSERVATIONS = False
        self.live_calls = 0

        def get_sensor_observations():
            self.live_calls += 1
            return {"audio_sensor": rir_fn(self.live_calls - 1)}
        self._sim = NS(get_sensor_observations=get_sensor_observations)
        return self


class FakeContinuousSim:
    """The attributes of soundspaces.continuous_simulator.ContinuousSoundSpacesSim that its audio code touches
    (continuous_simulator.py:370-462); `step()` advances them exactly like the reference's `step` (:384-390)."""

    def __init__(self, sr, sounds, rir_fn, step_time=0.25, crossfade=True, start_index=0):
        self.config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, CROSSFADE=crossfade), STEP_TIME=step_time)
        self._source_sound_dict = {k: O.tile_short_source(v, sr) for k, v in sounds.items()}      # :408-410
        self._current_sound = next(iter(sounds))
        self._episode_step_count = 0
        self._duration = 500
        self._rir_fn = rir_fn                       # step number -> [2][L] nested lists, like the habitat_sim audio sensor
        self._k = 0
        self._prev_sim_obs = {"audio_sensor": rir_fn(0)}
        self._last_rir = None                       # reconfigure (:341)
        self._current_sample_index = start_index    # reconfigure draws randint(sr * STEP_TIME) (:342)

    @property
    def current_source_sound(self):
        return self._source_sound_dict[self._current_sound]

    def step(self):
        self._last_rir = np.transpose(np.array(self._prev_sim_obs["audio_sensor"]))               # :384
        self._k += 1
        self._prev_sim_obs = {"audio_sensor": self._rir_fn(self._k)}
        self._episode_step_count += 1
        self._current_sample_index = int(self._current_sample_index + self.config.AUDIO.RIR_SAMPLING_RATE *
                                         self.config.STEP_TIME) % self.current_source_sound.shape[0]   # :389-390

    def reference_audiogoal(self):
        """continuous_simulator.py:413-426 via the oracle's restatement."""
        sr = self.config.AUDIO.RIR_SAMPLING_RATE
        rir = np.transpose(np.array(self._prev_sim_obs["audio_sensor"]))
        return O.compute_audiogoal_continuous(self.current_source_sound, rir, sr, self._current_sample_index,
                                              self.config.STEP_TIME, last_rir=self._last_rir,
                                              use_crossfade=self.config.AUDIO.CROSSFADE,
                                              silent=self._episode_step_count > self._duration)


def window_conv(source, rir, t0, n, wrap):
    """out[c, t] = sum_k rir[k, c] x[t0 + t - k], t < n, with x = 0 for negative indices and, past the clip end, either
    0 or (wrap) the clip again from its start -- the UnitRequest semantics, evaluated directly with scipy."""
    from scipy.signal import fftconvolve
    S, L = source.shape[0], rir.shape[0]
    idx = np.arange(t0 - L + 1, t0 + n)
    seg = np.zeros(idx.shape[0], np.float64)
    inside = (idx >= 0) & (idx < S)
    seg[inside] = source[idx[inside]]
    if wrap:
        over = (idx >= S) & (idx < 2 * S)
        seg[over] = source[idx[over] - S]
    if L == 0:
        return np.zeros((2, n))
    return np.stack([fftconvolve(seg, rir[:, c].astype(np.float64), mode="valid") for c in range(2)])


class OracleEngine:
    def __init__(self, sr, step_time=None):
        self.sr = sr
        self.n_valid = sr if step_time is None else int(sr * step_time)
        self.wrap = step_time is not None
        self.sources, self.names, self.rirs, self.keys = [], {}, [], {}
        self.calls = 0
        self.uploads = 0

    def source_id(self, name, clip):
        if name not in self.names:
            self.names[name] = len(self.sources)
            self.sources.append(np.asarray(clip, np.float32))
        return self.names[name]

    def rir_slot(self, key, loader, refresh=False):
        if key in self.keys and not refresh:
            return self.keys[key]
        r = loader()
        self.uploads += 1
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
            elif self.wrap:                       # SS2.0 engine: 0.25-s steps, wrap in the steady branch, cross-fade
                src = self.sources[u.sound]
                a = np.zeros((2, sr))
                a[:, :self.n_valid] = window_conv(src, self.rirs[u.rir], u.t0, self.n_valid, u.wrap is not False)
                if u.last_rir >= 0:
                    lw = u.wrap if u.last_wrap is None else u.last_wrap
                    b = np.zeros((2, sr))
                    b[:, :self.n_valid] = window_conv(src, self.rirs[u.last_rir], u.t0, self.n_valid, lw is not False)
                    a = O.crossfade(b, a, sr)
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


class OracleContext:
    """Stands in for ss_amd.context.AudioContext on machines without a GPU: same observe() / add_source() surface, the
    arithmetic done by the oracle into CPU tensors.  `bank(slot)` -> [L, 2] RIR of a bank slot."""

    def __init__(self, sr, bank):
        self.sr, self.bank = sr, bank
        self.sources, self.names, self.lengths = [], {}, []
        self.calls = 0
        self.spectrogram_shape = O.spectrogram_shape(sr)

    def add_source(self, name, clip):
        if name not in self.names:
            self.names[name] = len(self.sources)
            self.sources.append(np.asarray(clip, np.float32))
            self.lengths.append(len(clip))
        return self.names[name]

    def observe(self, sound, t0, rir, spectrogram_out=None, audiogoal_out=None, dis_sound=None, dis_rir=None, **kw):
        self.calls += 1
        sr = self.sr
        for i in range(len(sound)):
            if rir[i] < 0:
                a = np.zeros((2, sr), np.float32)
            else:
                a = O.conv_window_fft(self.sources[sound[i]], self.bank(int(rir[i])), int(t0[i]), sr)
                if dis_rir is not None and dis_rir[i] >= 0:
                    a = a + O.conv_window_fft(self.sources[dis_sound[i]], self.bank(int(dis_rir[i])), 0, sr)
            a = a.astype(np.float32)
            if audiogoal_out is not None:
                audiogoal_out[i] = torch.from_numpy(a)
            if spectrogram_out is not None:
                spectrogram_out[i] = torch.from_numpy(O.compute_spectrogram(a).astype(np.float32))


class OracleColumnEngine(OracleEngine):
    """OracleEngine with the COLUMN surface of ss_amd.renderer.AudioEngine (``store`` = a real RirStore on the CPU,
    ``observe_columns``), so that DeferredResolver's column path - CRC keys, resident-pair arrays, eviction hook, clipped-row
    reloads - runs without a GPU; the arithmetic is the oracle's."""

    def __init__(self, sr, slots=64, **kw):
        super().__init__(sr, **kw)
        from ss_amd.renderer import RirStore, UnitRequest
        self._unit = UnitRequest
        self.store = RirStore(slots, sr, "cpu", truncate_to=None if self.wrap else sr, max_cap=1 << 17)
        self.store.defer_uploads = True                    # as AudioEngine: single rows queue up until the launch
        self.renderer = NS(spectrogram_shape=O.spectrogram_shape(sr), device=torch.device("cpu"), out_len=sr, sr=sr)
        self.column_calls = 0

    def source_id(self, name, clip):
        if self.store.truncate_to is not None and np.shape(clip)[0] != self.sr:
            self.store.truncate_to = None                  # as AudioEngine.source_id: whole RIRs from now on
        return super().source_id(name, clip)

    # ---- the C record path (AudioEngine.observe_requests) with the launch replaced by the oracle: the lookups run in the real
    # library (ss_ctx_requests_units: host only), so DeferredResolver's miss handling - report, load, call again - runs on CPU
    def enable_native_requests(self):
        from ss_amd.context import AudioContext
        self._ctx = AudioContext(self.sr)
        self._req_miss = dict(buf=np.zeros((0,), np.int32))
        self.request_calls = 0
        self.observe_requests = self._observe_requests
        self.context = lambda: self._ctx
        return self

    def source_id_ctx(self, name, clip):
        sid = self.source_id(name, clip)
        if getattr(self, "_ctx", None) is not None:
            assert self._ctx.add_source_len(name, len(clip)) == sid
        return sid

    def _observe_requests(self, recs, n, tables, spectrogram_out=None, audiogoal_out=None):
        self.request_calls += 1
        for name, sid in sorted(self.names.items(), key=lambda kv: kv[1]):      # sounds registered since the last call
            self._ctx.add_source_len(name, len(self.sources[sid]))
        cols, miss = self._ctx.requests_units(recs, n, tables)
        self._req_miss["buf"] = miss
        if miss.shape[0]:
            return int(miss.shape[0])
        cols = {k: v for k, v in cols.items() if not (k.startswith("dis_") and not (cols["dis_rir"] >= 0).any())}
        self.observe_columns(cols, spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out)
        return 0

    def _row(self, slot):
        n = int(self.store.host_len[slot])
        return np.ascontiguousarray(self.store.bank.data[slot, :, :n].numpy().T)

    def observe_columns(self, cols, spectrogram_out=None, audiogoal_out=None):
        self.column_calls += 1
        self.store.flush_uploads()                         # (AudioEngine: _sync_context_bank -> sync_spectra -> flush)
        n = len(cols["sound"])
        units = []
        self.rirs = {}
        for i in range(n):
            if cols["rir"][i] < 0:
                units.append(self._unit(silent=True))
                continue
            u = self._unit(sound=int(cols["sound"][i]), t0=int(cols["t0"][i]), rir=int(cols["rir"][i]))
            self.rirs[u.rir] = self._row(u.rir)
            if "dis_rir" in cols and cols["dis_rir"][i] >= 0:
                u.dis_sound, u.dis_rir = int(cols["dis_sound"][i]), int(cols["dis_rir"][i])
                self.rirs[u.dis_rir] = self._row(u.dis_rir)
            if "wrap" in cols:
                u.wrap = bool(cols["wrap"][i])
            if "last_rir" in cols and cols["last_rir"][i] >= 0:
                u.last_rir, u.last_wrap = int(cols["last_rir"][i]), bool(cols["last_wrap"][i])
                self.rirs[u.last_rir] = self._row(u.last_rir)
            units.append(u)
        self.observe(units, want_audiogoal=audiogoal_out is not None, want_spectrogram=spectrogram_out is not None,
                     spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out)
