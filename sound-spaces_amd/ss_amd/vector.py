"""Vector-step audio for in-process vector envs WITHOUT per-env Python on the step path.

The reference produces a step's audio observations with a Python call chain per env
(``sensor.get_observation`` -> ``sim.get_current_spectrogram_observation`` -> ``_compute_audiogoal``,
soundspaces/tasks/nav.py:102-105, soundspaces/simulator.py:608-701) and then stacks them in ``batch_obs``
(ss_baselines/common/utils.py:126-153).  ``VectorAudioObserver`` (sim_audio.py) already turns that into one launch, but
still walks the envs in Python to read their state.  Here the audio-relevant state of all N simulators lives in numpy
COLUMNS:

* ``VectorSimState``  struct-of-arrays: sound id, ``_audio_index``, ``_episode_step_count``, ``_duration``, receiver /
  source / distractor node, ``_rotation_angle``, scene id — one row per env.  ``bind(sim, row)`` re-homes those
  attributes of a live simulator into its row (data descriptors on a per-class subclass), so the simulator's own
  ``step`` / ``reconfigure`` keep reading and writing ``self._receiver_position_index`` etc. unchanged and every
  write lands in the column; ``gather(sims)`` is the plain copy loop for simulators one would rather not touch.
* ``RirIndex``  (scene, receiver, source) -> first bank slot of the pair's 4 azimuths (stored adjacently,
  ``RirStore(group=4)``), dense int32 tables built at scene load: the lookup of a whole step is one fancy index
  (reference: one ``os.path.join`` + ``wavfile.read`` per env and step, simulator.py:615-618).
* ``FastVectorAudioObserver.observe_into(rollouts)``  ~10 numpy operations on N-vectors -> ONE ctypes call
  (``ss_ctx_observe``: planner, window cache and descriptor upload are C++ inside libss_hip.so) that writes the
  spectrograms of all envs straight into ``rollouts.observations['spectrogram'][step + 1]``.

Semantics are those of the reference's cache-miss path (``HAS_DISTRACTOR_SOUND`` behaviour, simulator.py:679-681):
every env renders its current pose every step; ``_audio_index`` advances exactly where ``_compute_audiogoal`` advances
it (:634-635: multi-second clips only, not when silent)."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

_NONE = np.iinfo(np.int64).min            # column value of an attribute that is None on the simulator

# simulator attribute -> column  (soundspaces/simulator.py:110-117, 303-305)
_INT_ATTRS = {
    "_episode_step_count": "step_count",
    "_duration": "duration",
    "_receiver_position_index": "recv",
    "_source_position_index": "src",
    "_distractor_position_index": "dis_src",
    "_rotation_angle": "rot",
    "_audio_index": "audio_index",
}
_NAME_ATTRS = {"_current_sound": "sound", "_current_distractor_sound": "dis_sound"}


class VectorSimState:
    def __init__(self, n: int):
        self.n = n
        for col in _INT_ATTRS.values():
            setattr(self, col, np.full((n,), _NONE, np.int64))
        self.scene = np.zeros((n,), np.int64)
        self.sound = np.full((n,), -1, np.int64)              # ids resolved lazily from the names (see resolve_sounds)
        self.dis_sound = np.full((n,), -1, np.int64)
        self.names: Dict[str, List[Optional[str]]] = {"sound": [None] * n, "dis_sound": [None] * n}
        self.dirty = np.zeros((n,), bool)                      # a sound name changed: id must be looked up again
        self.sims: List = [None] * n

    # ---- binding ---------------------------------------------------------------------------------------------
    def bind(self, sim, row: int) -> None:
        """Re-home the audio state of `sim` into row `row` (see the module docstring)."""
        cls = type(sim)
        bound = _bound_class(cls)
        values = {a: getattr(sim, a, None) for a in list(_INT_ATTRS) + list(_NAME_ATTRS)}
        for a in values:
            sim.__dict__.pop(a, None)
        object.__setattr__(sim, "_ss_state", self)
        object.__setattr__(sim, "_ss_row", row)
        sim.__class__ = bound
        for a, v in values.items():
            setattr(sim, a, v)
        self.sims[row] = sim

    def gather(self, sims: Sequence) -> None:
        """Copy loop (no binding): read every simulator's attributes into the columns."""
        for i, sim in enumerate(sims):
            for a, col in _INT_ATTRS.items():
                v = getattr(sim, a, None)
                getattr(self, col)[i] = _NONE if v is None else int(v)
            for a, col in _NAME_ATTRS.items():
                v = getattr(sim, a, None)
                if self.names[col][i] != v:
                    self.names[col][i] = v
                    self.dirty[i] = True
            self.sims[i] = sim

    def scatter_audio_index(self, sims: Sequence) -> None:
        """gather() mode only: write the advanced ``_audio_index`` back (bound simulators share the column)."""
        for i, sim in enumerate(sims):
            v = self.audio_index[i]
            if v != _NONE and not isinstance(getattr(type(sim), "_audio_index", None), property):
                sim._audio_index = int(v)

    def resolve_sounds(self, source_id: Callable[[str, np.ndarray], int]) -> None:
        """Names -> sound ids for the rows whose sound changed (episode boundaries only; needs the clip, which the
        simulator loads AFTER it sets ``_current_sound``, simulator.py:356-360 — hence lazily, at observation time)."""
        for i in np.flatnonzero(self.dirty):
            sim = self.sims[i]
            for col in ("sound", "dis_sound"):
                name = self.names[col][i]
                clip = sim._source_sound_dict.get(name) if (name is not None and sim is not None) else None
                getattr(self, col)[i] = source_id(name, clip) if clip is not None else -1
        self.dirty[:] = False


_BOUND: Dict[type, type] = {}


def _bound_class(cls: type) -> type:
    if cls in _BOUND:
        return _BOUND[cls]
    ns = {}

    def int_col(col):
        def get(self):
            v = getattr(self._ss_state, col)[self._ss_row]
            return None if v == _NONE else int(v)

        def put(self, v):
            getattr(self._ss_state, col)[self._ss_row] = _NONE if v is None else int(v)
        return property(get, put)

    def name_col(col):
        def get(self):
            return self._ss_state.names[col][self._ss_row]

        def put(self, v):
            st = self._ss_state
            if st.names[col][self._ss_row] != v:
                st.names[col][self._ss_row] = v
                st.dirty[self._ss_row] = True
        return property(get, put)

    for a, col in _INT_ATTRS.items():
        ns[a] = int_col(col)
    for a, col in _NAME_ATTRS.items():
        ns[a] = name_col(col)
    bound = type("SsBound" + cls.__name__, (cls,), ns)
    _BOUND[cls] = bound
    return bound


class RirIndex:
    """(scene, receiver node, source node) -> first bank slot of the pair's azimuth group, or -1."""

    def __init__(self, azimuths: int = 4):
        self.azimuths = azimuths
        self._tables: List[np.ndarray] = []
        self._names: Dict[str, int] = {}
        self._flat = np.zeros((0,), np.int32)
        self._off = np.zeros((0,), np.int64)
        self._dim = np.zeros((0,), np.int64)
        self._stale = False
        self.version = 0                                       # bumps whenever the flat tables are re-allocated

    def add_scene(self, name: str, n_nodes: int) -> int:
        if name in self._names:
            return self._names[name]
        self._names[name] = len(self._tables)
        self._tables.append(np.full((n_nodes, n_nodes), -1, np.int32))
        self._stale = True
        return self._names[name]

    def scene_id(self, name: str) -> int:
        return self._names[name]

    def ensure_nodes(self, scene: int, n_nodes: int) -> None:
        """Grow the table of `scene` to at least n_nodes x n_nodes (scenes registered before their node count is known:
        the deferred resolver learns the nodes from the requests it sees)."""
        t = self._tables[scene]
        if n_nodes <= t.shape[0]:
            return
        n = max(n_nodes, 2 * t.shape[0])
        g = np.full((n, n), -1, np.int32)
        g[:t.shape[0], :t.shape[0]] = t
        self._tables[scene] = g
        self._stale = True

    def set(self, scene: int, recv, src, base) -> None:
        self._tables[scene][recv, src] = base
        if self._stale:
            return
        try:                                                   # keep the flat copy (and pointers into it) current
            self._flat[self._off[scene]:self._off[scene] + self._dim[scene] ** 2].reshape(self._dim[scene], -1)[recv, src] = base
        except (IndexError, ValueError):
            self._stale = True

    def tables(self):
        """(flat int32, offsets int64, dims int64) of all scenes; stable until ``version`` changes."""
        if self._stale:
            self._rebuild()
        return self._flat, self._off, self._dim

    def _rebuild(self) -> None:
        self._dim = np.array([t.shape[0] for t in self._tables], np.int64)
        self._off = np.concatenate([[0], np.cumsum(self._dim * self._dim)[:-1]]).astype(np.int64)
        self._flat = np.concatenate([t.reshape(-1) for t in self._tables]) if self._tables else np.zeros((0,), np.int32)
        self._stale = False
        self.version += 1

    def lookup(self, scene: np.ndarray, recv: np.ndarray, src: np.ndarray, azimuth: np.ndarray) -> np.ndarray:
        """Vectorised: bank slot per env (base + azimuth // (360 / azimuths)), -1 where the pair is not resident."""
        if self._stale:
            self._rebuild()
        dim = self._dim[scene]
        ok = (recv >= 0) & (recv < dim) & (src >= 0) & (src < dim)
        flat = self._off[scene] + np.where(ok, recv, 0) * dim + np.where(ok, src, 0)
        base = self._flat[flat]
        slot = base + azimuth // (360 // self.azimuths)
        return np.where(ok & (base >= 0), slot, -1)


def load_scene_pairs(store, index: RirIndex, scene: str, scene_rir_dir: str, reader, n_nodes: Optional[int] = None,
                     azimuths=(0, 90, 180, 270), workers: int = 8, limit: Optional[int] = None) -> int:
    """Scene load: every ``<scene_rir_dir>/<azimuth>/<receiver>_<source>.wav`` (soundspaces/README.md:38-42,
    simulator.py:615-616) into the store with the azimuths of a pair in ADJACENT slots (``store.group == len(azimuths)``)
    and the pair's first slot into the index.  Returns the number of pairs loaded."""
    import os
    assert store.group == len(azimuths)
    pairs = set()
    for az in azimuths:
        d = os.path.join(scene_rir_dir, str(az))
        if os.path.isdir(d):
            for name in os.listdir(d):
                if name.endswith(".wav"):
                    a, b = name[:-4].split("_")
                    pairs.add((int(a), int(b)))
    pairs = sorted(pairs)
    if limit is not None:
        pairs = pairs[:limit]
    if not pairs:
        return 0
    n_nodes = n_nodes or 1 + max(max(p) for p in pairs)
    sid = index.add_scene(scene, n_nodes)

    def group_loader(pair):
        def load():
            out = []
            for az in azimuths:
                path = os.path.join(scene_rir_dir, str(az), "{}_{}.wav".format(*pair))
                out.append(reader(path) if os.path.exists(path) else None)
            return out
        return load
    chunk = max(1, store.slots // store.group // 2)
    from .renderer import _native_wav
    native = _native_wav(reader) and hasattr(store, "load_files")
    for lo in range(0, len(pairs), chunk):
        part = pairs[lo:lo + chunk]
        if native:                                              # the library's own reader: no scipy, no host transposes
            files = []
            for pair in part:
                fl = [os.path.join(scene_rir_dir, str(az), "{}_{}.wav".format(*pair)) for az in azimuths]
                files.append([f if os.path.exists(f) else None for f in fl] if store.group > 1 else fl[0])
            bases = store.load_files([(scene, p) for p in part], files, reader=reader, threads=workers)
        else:
            bases = store.slot_many([(scene, p) for p in part], [group_loader(p) for p in part], workers=workers)
        r, s = np.array([p[0] for p in part]), np.array([p[1] for p in part])
        index.set(sid, r, s, np.asarray(bases, np.int32))
    return len(pairs)


class FastVectorAudioObserver:
    """One launch per vector step, state read from columns (module docstring)."""

    def __init__(self, ctx, state: VectorSimState, index: RirIndex, sampling_rate: int, has_distractor: bool = False,
                 miss: Optional[Callable[[int, int, int, int], int]] = None, native: bool = True, pose_cache: bool = False):
        """ctx: ss_amd.context.AudioContext with its RIR bank set; miss(env, recv, src, azimuth) -> slot is called for
        envs whose pair is not in the index (loads it AND enters it into the index; default: raise).
        pose_cache: the REFERENCE-EXACT mode for multi-second sounds.  The reference memoises observations per (source,
        receiver, azimuth) (simulator.py:678-701); the key ignores ``_audio_index``, so an agent that stands still or comes
        back to a pose gets the observation FIRST rendered there again - old clip window, even a silent or pre-silence one -
        and ``_audio_index`` does not advance on the hit (:634-635 run inside the miss only).  With pose_cache=True the
        column observer does the same: a per-env map pose -> row of a device-side pool; hits are device-to-device row
        copies, misses are rendered and stored; an env's map is dropped when its scene or sound changes (:395-397).
        Ignored with a distractor (the reference bypasses its caches then, :679-681).  Default False: every step renders
        the cache-miss path (identical for 1-s sounds, whose window never changes)."""
        self.ctx, self.state, self.index, self.sr = ctx, state, index, int(sampling_rate)
        self.has_distractor = has_distractor
        self.miss = miss
        self.pose_cache = bool(pose_cache) and not has_distractor
        self._pose_maps: List[Dict[int, int]] = [dict() for _ in range(state.n)]
        self._pose_tag = np.full((state.n, 2), -2, np.int64)      # (scene, sound) the env's map belongs to
        self._pool: Dict[str, object] = {}                       # output name -> [capacity, ...] device tensor of cached rows
        self._pool_free: List[int] = []
        self._pool_cap = 0
        self.pose_hits = self.pose_misses = 0
        self.native = native and hasattr(ctx, "observe_sims") and not self.pose_cache
        self._bound, self._bound_version = None, -1
        self._rollouts, self._names = None, []
        self._clip_len = np.zeros((0,), np.int64)

    def _lengths(self) -> np.ndarray:
        if self._clip_len.shape[0] != len(self.ctx.lengths):
            self._clip_len = np.asarray(self.ctx.lengths, np.int64)
        return self._clip_len

    def columns(self, hold: Optional[np.ndarray] = None):
        """State -> the unit columns of this step (and advance ``_audio_index`` like simulator.py:634-635).  hold: envs
        (bool mask) served from the pose cache this step: rendered as silent rows, index not advanced."""
        st, sr = self.state, self.sr
        if st.dirty.any():
            st.resolve_sounds(self.ctx.add_source)
        known = st.sound >= 0
        clip_len = np.where(known, self._lengths()[np.where(known, st.sound, 0)], sr)
        silent = (st.step_count > st.duration) | ~known                                              # :610
        multi = clip_len != sr                                                                       # :629
        t0 = np.where(multi, st.audio_index * sr, 0)
        az = (-st.rot) % 360                                                                         # :573
        rir = self.index.lookup(st.scene, st.recv, st.src, az)
        # pairs that are not resident are resolved (or the step refused) BEFORE anything is advanced, exactly like the
        # native path (ss_ctx_observe_sims: "nothing launched, nothing advanced")
        if self.miss is not None:
            for i in np.flatnonzero((rir < 0) & ~silent):
                rir[i] = self.miss(int(i), int(st.recv[i]), int(st.src[i]), int(az[i]))
        if hold is not None:
            silent = silent | hold                            # no render, no RIR needed for a hit
        cols = dict(sound=np.where(silent, 0, st.sound), t0=np.where(silent, 0, t0), rir=np.where(silent, -1, rir))
        bad = (cols["rir"] < 0) & ~silent
        if self.has_distractor:                                                                      # :649-664
            has_dis = (st.dis_sound >= 0) & ~silent
            dr = self.index.lookup(st.scene, st.recv, st.dis_src, az)
            if self.miss is not None:
                for i in np.flatnonzero((dr < 0) & has_dis):
                    dr[i] = self.miss(int(i), int(st.recv[i]), int(st.dis_src[i]), int(az[i]))
            bad |= (dr < 0) & has_dis
            # no distractor sound -> no distractor term (the native path sets drir = -1 as well)
            cols.update(dis_sound=np.where(has_dis, st.dis_sound, 0), dis_rir=np.where(has_dis, dr, -1))
        if bad.any():
            i = int(np.flatnonzero(bad)[0])
            raise KeyError(f"no RIR loaded for env {i}: (scene {int(st.scene[i])}, receiver {int(st.recv[i])}, source "
                           f"{int(st.src[i])}, azimuth {int(az[i])}) - pass miss= to load pairs on demand")
        adv = multi & ~silent & (clip_len >= sr)          # (clips shorter than 1 s never advance: len // sr == 0)
        if hold is not None:                              # pose-cache hits: the reference does not run _compute_audiogoal
            adv &= ~hold
        if adv.any():
            st.audio_index[adv] = (st.audio_index[adv] + 1) % (clip_len[adv] // sr)               # :635
        return cols

    def observe(self, spectrogram_out=None, audiogoal_out=None) -> None:
        """Native path (default): the per-step host work runs inside libss_hip.so (``ss_ctx_observe_sims``) on pointers
        to the state columns; ``native=False`` keeps the numpy formulation of ``columns()`` (same results, tested)."""
        if self.pose_cache:
            return self._observe_with_pose_cache(spectrogram_out, audiogoal_out)
        if not self.native:
            self.ctx.observe(spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out, **self.columns())
            return
        st = self.state
        if st.dirty.any():
            st.resolve_sounds(self.ctx.add_source)
        for _ in range(2):
            self.index.tables()                                                                      # rebuild if stale
            if self._bound is None or self._bound_version != self.index.version:
                self._bound = self.ctx.bind_sims(st, self.index, self.has_distractor)
                self._bound_version = self.index.version
            missing = self.ctx.observe_sims(self._bound, spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out)
            if missing.shape[0] == 0:
                return
            if self.miss is None:
                raise KeyError(f"RIR pairs of envs {missing.tolist()} are not resident and no miss loader was given")
            az = (-st.rot) % 360
            for i in missing:                               # load the pairs (the loader puts them into the index)
                i = int(i)
                self.miss(i, int(st.recv[i]), int(st.src[i]), int(az[i]))
                if self.has_distractor and st.dis_sound[i] >= 0:
                    self.miss(i, int(st.recv[i]), int(st.dis_src[i]), int(az[i]))
        raise KeyError("RIR pairs still missing after the miss loader ran (it must enter them into the RirIndex)")

    # ---- reference-exact pose cache (see __init__) ---------------------------------------------------------------------
    def _pool_rows(self, outs, need: int) -> List[int]:
        import torch
        while len(self._pool_free) < need:                   # grow the pools (contents kept)
            new_cap = max(256, 2 * self._pool_cap)
            for name, t in outs.items():
                grown = torch.empty((new_cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                if name in self._pool:
                    grown[:self._pool_cap] = self._pool[name]
                self._pool[name] = grown
            self._pool_free += list(range(new_cap - 1, self._pool_cap - 1, -1))
            self._pool_cap = new_cap
        return [self._pool_free.pop() for _ in range(need)]

    def _observe_with_pose_cache(self, spectrogram_out, audiogoal_out) -> None:
        import torch
        st = self.state
        if st.dirty.any():
            st.resolve_sounds(self.ctx.add_source)
        outs = {k: v for k, v in (("spectrogram", spectrogram_out), ("audiogoal", audiogoal_out)) if v is not None}
        if set(self._pool) - set(outs) or (self._pool and set(outs) - set(self._pool)):
            raise ValueError("pose_cache: the observer must be asked for the same outputs every step")
        n = st.n
        az = (-st.rot) % 360
        key = (st.src << 40) | (st.recv << 20) | az           # (source, receiver, azimuth): simulator.py:683
        hit_rows, hit_pool, miss_rows = [], [], []
        for i in range(n):                                    # per-env maps: this mode trades the column path's speed for
            tag = (int(st.scene[i]), int(st.sound[i]))        # the reference's exact cache semantics
            if tag != (int(self._pose_tag[i, 0]), int(self._pose_tag[i, 1])):      # scene / sound changed (:395-397)
                self._pool_free += list(self._pose_maps[i].values())
                self._pose_maps[i].clear()
                self._pose_tag[i] = tag
            j = self._pose_maps[i].get(int(key[i]))
            if j is None:
                miss_rows.append(i)
            else:
                hit_rows.append(i)
                hit_pool.append(j)
        hold = np.zeros((n,), bool)
        hold[hit_rows] = True
        self.pose_hits += len(hit_rows)
        self.pose_misses += len(miss_rows)
        self.ctx.observe(spectrogram_out=spectrogram_out, audiogoal_out=audiogoal_out, **self.columns(hold=hold))
        if (miss_rows or hit_rows) and hasattr(self.ctx, "join"):
            self.ctx.join()                                   # (overlap lanes: the pool copies below READ and WRITE the step's rows)
        dev = next(iter(outs.values())).device
        if miss_rows:                                         # store what was rendered (silent rows included: the
            slots = self._pool_rows(outs, len(miss_rows))     # reference caches its zeros under the pose as well)
            for i, j in zip(miss_rows, slots):
                self._pose_maps[i][int(key[i])] = j
            mi = torch.as_tensor(miss_rows, dtype=torch.long, device=dev)
            sj = torch.as_tensor(slots, dtype=torch.long, device=dev)
            for name, t in outs.items():
                self._pool[name].index_copy_(0, sj, t.index_select(0, mi))
        if hit_rows:
            hi = torch.as_tensor(hit_rows, dtype=torch.long, device=dev)
            hj = torch.as_tensor(hit_pool, dtype=torch.long, device=dev)
            for name, t in outs.items():
                t.index_copy_(0, hi, self._pool[name].index_select(0, hj))

    def observe_into(self, rollouts):
        """Render this vector step straight into the rollout rows the next ``rollouts.insert()`` fills
        (ss_baselines/common/rollout_storage.py:89-92); returns the slots (a ``DeviceObservations``)."""
        if self._rollouts is not rollouts:
            self._rollouts, self._names = rollouts, [s for s in ("spectrogram", "audiogoal") if s in rollouts.observations]
        slots = rollouts.next_observation_slots(self._names)
        self.observe(spectrogram_out=slots.get("spectrogram"), audiogoal_out=slots.get("audiogoal"))
        return slots
