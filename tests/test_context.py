"""The context API of libss_hip.so (include/ss_hip.h: ss_ctx_*).  The planner + window cache are host C++ inside the
library and run without a GPU (ss_ctx_plan), so they are checked here against the Python planner (ss_amd/planning.py,
the executable specification) on random steps; the GPU half (ss_ctx_observe) is in the `-m gpu` tests below."""
import numpy as np
import pytest

from oracle import ss_oracle as O
from ss_amd import planning as P
from ss_amd.context import AudioContext

SR = 16000


def py_plan(lengths, cap, n_valid, wrap_mode, sound, t0, rir, dis_sound=None, dis_rir=None, last_rir=None, wrap=None,
            last_wrap=None):
    """planning.py applied unit by unit -> per unit ((rir, m_min, count, starts...), (term-1 ...)) without slot numbers."""
    nbh_max, nby = max(1, P.ceil_div(cap, P.KB)), max(1, P.ceil_div(n_valid, P.KB))
    out = []
    for i in range(len(sound)):
        if rir[i] < 0:
            out.append(None)
            continue
        L = lengths[sound[i]]
        over = t0[i] + n_valid > L
        w0 = bool(wrap_mode and over and (wrap is None or wrap[i]))
        ws = P.plan_window_set(L, int(t0[i]), nbh_max, nby, w0)
        if ws.count == 0:
            out.append(None)
            continue
        a = (int(rir[i]), ws.m_min, ws.count, ws.starts, w0, int(sound[i]))
        b = None
        if last_rir is not None and last_rir[i] >= 0:
            lw = wrap if last_wrap is None else last_wrap
            w1 = bool(wrap_mode and over and (lw is None or lw[i]))
            ws1 = P.plan_window_set(L, int(t0[i]), nbh_max, nby, w1)
            b = (int(last_rir[i]), ws1.m_min, ws1.count, ws1.starts, w1, int(sound[i]))
        elif dis_rir is not None and dis_rir[i] >= 0:
            ws1 = P.plan_window_set(lengths[dis_sound[i]], 0, nbh_max, nby, False)
            if ws1.count:
                b = (int(dis_rir[i]), ws1.m_min, ws1.count, ws1.starts, False, int(dis_sound[i]))
        out.append((a, b))
    return out


def check_against_python(ctx, lengths, offsets, cap, n_valid, wrap_mode, pool, **cols):
    """pool: dict slot -> (src_offset, src_len, start, wrap) of every window the context has ever planned."""
    desc, flags, wins = ctx.plan(**cols)
    for w in wins:
        pool[int(w[4])] = tuple(int(v) for v in w[:4])
    ref = py_plan(lengths, cap, n_valid, wrap_mode, **cols)
    any_b = False
    for i, r in enumerate(ref):
        d = desc[i]
        if r is None:
            assert d[0] == -1 and d[4] == -1
            continue
        for term, t in enumerate(r):
            row = d[4 * term:4 * term + 4]
            if t is None:
                assert row[0] == -1
                continue
            any_b |= term == 1
            ridx, m_min, count, starts, w, snd = t
            assert (row[0], row[2], row[3]) == (ridx, m_min, count)
            for k in range(count):                       # the slots hold exactly the windows planning.py asks for
                assert pool[int(row[1]) + k] == (offsets[snd], lengths[snd], starts[k], int(w))
    if cols.get("last_rir") is not None and any(r is not None and r[1] is not None for r in ref):
        assert flags == 2
    else:
        assert flags == (0 if any_b else 1)
    return desc, wins


@pytest.mark.parametrize("cap,n_valid,wrap_mode", [(16000, 16000, 0), (40000, 16000, 0), (44100, 44100, 0),
                                                   (20000, 4000, 1), (50000, 4000, 1)])
def test_cxx_planner_equals_python_planner(cap, n_valid, wrap_mode):
    rng = np.random.default_rng(cap + n_valid)
    sr = 44100 if n_valid == 44100 else SR
    lengths = [sr, sr, 3 * sr, 5 * sr, 20 * sr]
    offsets = list(np.cumsum([0] + lengths[:-1]))
    ctx = AudioContext(sr, step_time=None if n_valid == sr else n_valid / sr, wrap=bool(wrap_mode), max_window_sets=32)
    for i, L in enumerate(lengths):
        assert ctx.add_source_len(f"s{i}", L) == i
    ctx.set_rir_cap_for_planning(cap)
    pool = {}
    for step in range(30):
        n = int(rng.integers(1, 70))
        sound = rng.integers(0, 5, n)
        if wrap_mode:
            t0 = np.array([rng.integers(0, lengths[s]) for s in sound])
        else:
            t0 = np.array([0 if lengths[s] == sr else rng.integers(0, lengths[s] // sr + 1) * sr for s in sound])
        rir = rng.integers(-1, 50, n)
        cols = dict(sound=sound, t0=t0, rir=rir)
        if wrap_mode:
            cols["wrap"] = (rng.uniform(size=n) < 0.7).astype(np.uint8)
            if step % 2:
                cols["last_rir"] = rng.integers(-1, 50, n)
                cols["last_wrap"] = (rng.uniform(size=n) < 0.5).astype(np.uint8)
        elif step % 3 == 0:
            cols["dis_sound"] = rng.integers(0, 2, n)
            cols["dis_rir"] = rng.integers(-1, 50, n)
        check_against_python(ctx, lengths, offsets, cap, n_valid, wrap_mode, pool, **cols)
    st = ctx.stats()
    assert st["steps"] == 30 and st["hits"] > 0 and st["misses"] > 0
    assert st["slots_per_key"] == max(1, P.ceil_div(cap, P.KB)) + max(1, P.ceil_div(n_valid, P.KB)) - 1


def test_cache_is_bounded_lru_and_grows_only_when_one_step_needs_it():
    """the Python renderer's window cache only grew (SS2.0 draws a new t0 per env and step -> HBM leak).  The
    context's cache recycles least-recently-used keys; keys of recent steps are protected; a step that alone needs
    more keys than the cache holds grows it."""
    ctx = AudioContext(SR, step_time=0.25, wrap=True, max_window_sets=16)
    ctx.add_source_len("s", 3 * SR)
    ctx.set_rir_cap_for_planning(SR)
    z = lambda n: np.zeros(n, np.int32)
    for step in range(200):                               # 4 envs, a new t0 each step: 800 distinct keys in total
        t0 = (np.arange(4) * 7919 + step * 4000) % (3 * SR)
        ctx.plan(z(4), t0, z(4))
    st = ctx.stats()
    assert st["capacity"] <= 64 and st["resident"] <= st["capacity"] and st["evictions"] > 600
    ctx.plan(z(4), (np.arange(4) * 7919 + 199 * 4000) % (3 * SR), z(4))          # the last step again: all hits
    assert ctx.stats()["misses"] == st["misses"]
    big = AudioContext(SR, max_window_sets=4)
    for i in range(40):
        big.add_source_len(f"s{i}", SR)
    big.set_rir_cap_for_planning(SR)
    desc, flags, wins = big.plan(np.arange(40), z(40), z(40))                     # 40 keys in ONE step, capacity 4
    assert len(wins) == 40 and len(set(desc[:, 1])) == 40 and big.stats()["grows"] >= 4


def test_planner_rejects_bad_input():
    from ss_amd._lib import SsHipError
    ctx = AudioContext(SR)
    ctx.add_source_len("s", SR)
    ctx.set_rir_cap_for_planning(SR)
    with pytest.raises(SsHipError):
        ctx.plan(np.array([3]), np.array([0]), np.array([0]))                     # unknown sound id
    with pytest.raises(SsHipError):                                              # distractor AND previous RIR on one unit
        ctx.plan(np.array([0]), np.array([0]), np.array([0]), dis_sound=np.array([0]), dis_rir=np.array([1]),
                 last_rir=np.array([1]))


def test_refused_step_leaves_no_keys_behind():
    """plan_units() used to insert cache entries for units 0..i-1 before refusing the step at unit i; the next
    plan then reported those keys as HITS although no spectra had been computed for them.  Planning is transactional
    now: a refused step does not touch the cache."""
    from ss_amd._lib import SsHipError
    ctx = AudioContext(SR)
    for i in range(3):
        ctx.add_source_len(f"s{i}", SR)
    ctx.set_rir_cap_for_planning(SR)
    z = lambda *v: np.array(v, np.int32)
    with pytest.raises(SsHipError):
        ctx.plan(z(0, 3), z(0, 0), z(0, 0))                                       # unit 0 fine, unit 1: unknown sound
    with pytest.raises(SsHipError):                                              # unit 0 fine, unit 1: dis + last
        ctx.plan(z(1, 0), z(0, 0), z(0, 0), dis_sound=z(0, 0), dis_rir=z(-1, 1), last_rir=z(-1, 1))
    with pytest.raises(SsHipError):                                              # a cross-fade and a distractor in one launch
        ctx.plan(z(2, 0), z(0, 0), z(0, 0), dis_sound=z(0, 0), dis_rir=z(-1, 1), last_rir=z(1, -1))
    st = ctx.stats()
    assert st["resident"] == 0 and st["misses"] == 0 and st["hits"] == 0 and st["steps"] == 0
    desc, flags, wins = ctx.plan(z(0), z(0), z(0))
    assert len(wins) == 1 and ctx.stats()["misses"] == 1 and ctx.stats()["hits"] == 0      # a miss, not a poisoned hit


# ---- GPU half ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_failed_or_plan_only_steps_do_not_poison_the_window_cache():
    """(1) a refused ss_ctx_observe, then the same keys must render correctly; (2) ss_ctx_plan on a context that
    also observes: the planned-only keys have no spectra - the next observe starts from an empty cache; (3) a bank swap
    drops the spectral form of the old bank (ss_ctx_set_rir_bank resets it)."""
    import torch
    from ss_amd._lib import SsHipError
    from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest
    dev = "cuda:0"
    rng = np.random.default_rng(18)
    src = O.synth_sources(rng, SR, k=3)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=4)]
    bank = RirBank.from_arrays(rirs, dev)
    r = BatchedAudioRenderer(SR, device=dev)
    ctx = AudioContext(SR)
    for i, s_ in enumerate(src):
        r.add_source(f"s{i}", s_)
        ctx.add_source(f"s{i}", s_)
    r.set_rir_bank(bank)
    ctx.set_rir_bank(bank.data, bank.lengths)
    sg = torch.full((2, 65, 26, 2), float("nan"), device=dev)
    with pytest.raises(SsHipError):
        ctx.observe([0, 7], [0, 0], [0, 1], spectrogram_out=sg)                  # unit 1: unknown sound -> nothing launched
    want = r.render(r.plan([UnitRequest(0, 0, 0), UnitRequest(1, 0, 1)]))[1]
    ctx.observe([0, 1], [0, 0], [0, 1], spectrogram_out=sg)
    torch.cuda.synchronize()
    assert torch.equal(sg, want)
    ctx.plan([2, 1], [0, 0], [2, 3])                                             # planner only: key (2, 0) gets no spectrum
    want2 = r.render(r.plan([UnitRequest(2, 0, 2), UnitRequest(0, 0, 3)]))[1]
    sg.fill_(float("nan"))
    ctx.observe([2, 0], [0, 0], [2, 3], spectrogram_out=sg)
    torch.cuda.synchronize()
    assert torch.equal(sg, want2)
    # spectral form, then a different bank WITHOUT new spectra: must render the new bank (time-domain kernels)
    bank.build_spectra()
    ctx.set_rir_spectra(bank.spectra)
    ctx.observe([2, 0], [0, 0], [2, 3], spectrogram_out=sg)
    torch.cuda.synchronize()
    assert float((sg - want2).abs().max()) <= 2e-6 * float(want2.abs().max())
    bank2 = RirBank.from_arrays(rirs[::-1], dev)
    r.set_rir_bank(bank2)
    ctx.set_rir_bank(bank2.data, bank2.lengths)
    want3 = r.render(r.plan([UnitRequest(2, 0, 2), UnitRequest(0, 0, 3)]))[1]
    ctx.observe([2, 0], [0, 0], [2, 3], spectrogram_out=sg)
    torch.cuda.synchronize()
    assert torch.equal(sg, want3)


@pytest.mark.gpu
def test_overlap_mode_orders_a_step_behind_pending_work_of_the_callers_stream():
    """The overlap mode's input fence (event record on the caller's stream + wait on the lane) is skipped when the caller's
    stream has nothing pending (hipStreamQuery) - and ONLY then: RIR rows rewritten by a copy that is still queued behind
    milliseconds of other work on the caller's stream when ss_ctx_observe is called must be the rows the step renders."""
    import torch
    from ss_amd.renderer import RirBank
    dev = "cuda:0"
    rng = np.random.default_rng(41)
    src = O.synth_sources(rng, SR, k=2)
    rows_a = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=8)]
    rows_b = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=8)]
    bank = RirBank.from_arrays(rows_a, dev)
    data_a = bank.data.clone()
    data_b = RirBank.from_arrays(rows_b, dev).data.clone()
    assert data_a.shape == data_b.shape
    ref, ctx = AudioContext(SR), AudioContext(SR)
    for c in (ref, ctx):
        for i, s_ in enumerate(src):
            c.add_source(f"s{i}", s_)
        c.set_rir_bank(bank.data, bank.lengths)
    ctx.set_overlap(4)
    sound, t0, rir = rng.integers(0, 2, 8), np.zeros(8, np.int64), np.arange(8)
    want = {}
    for name, d in (("a", data_a), ("b", data_b)):
        bank.data.copy_(d)
        torch.cuda.synchronize()
        want[name] = torch.empty((8, 65, 26, 2), device=dev)
        ref.observe(sound, t0, rir, spectrogram_out=want[name])
        torch.cuda.synchronize()
    assert not torch.equal(want["a"], want["b"])
    big = torch.randn((4096, 4096), device=dev)
    got = torch.empty((8, 65, 26, 2), device=dev)
    for trial, name in enumerate("abab"):
        if trial >= 2:
            torch.cuda.synchronize()                            # idle stream at the call: the fence is skipped
        else:
            acc = big
            for _ in range(6):                                  # ~ms of queued work in front of the row copy
                acc = (acc @ big) * 1e-3
        bank.data.copy_(data_a if name == "a" else data_b, non_blocking=True)
        if trial >= 2:
            torch.cuda.synchronize()
        ctx.observe(sound, t0, rir, spectrogram_out=got)
        ctx.join()
        torch.cuda.synchronize()
        assert torch.equal(got, want[name]), trial


@pytest.mark.gpu
@pytest.mark.parametrize("n_lanes", [2, 3, 4])
def test_overlap_mode_two_lanes_equal_single_stream(n_lanes):
    """ss_ctx_set_overlap(n), n = 2 .. 4: 90 steps alternate between the internal streams with nothing but ss_ctx_join at the end -
    cache misses whose spectra the OTHER lane's next step hits, evictions under a small cache, steps of different sizes
    (descriptors in place and uploaded), a refused step in the middle - every row equal to the single-stream context's."""
    import torch
    from ss_amd._lib import SsHipError
    from ss_amd.renderer import RirBank
    dev = "cuda:0"
    rng = np.random.default_rng(28)
    src = [O.synth_sources(rng, SR, k=1, seconds=s_)[0] for s_ in (1, 1, 3, 6, 9)]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=16)]
    bank = RirBank.from_arrays(rirs, dev)
    ctxs = [AudioContext(SR, max_window_sets=4), AudioContext(SR, max_window_sets=4)]
    for c in ctxs:
        for i, s_ in enumerate(src):
            c.add_source(f"s{i}", s_)
        c.set_rir_bank(bank.data, bank.lengths)
    ctxs[1].set_overlap(n_lanes)
    with pytest.raises(SsHipError):
        ctxs[0].set_overlap(5)                                                   # (at most four lanes)
    steps = []
    for k in range(90):
        n = int(rng.choice([3, 40, 300]))
        sound = rng.integers(0, 5, n)
        t0 = np.array([0 if len(src[s_]) == SR else int(rng.integers(0, len(src[s_]) // SR)) * SR for s_ in sound])
        steps.append((sound, t0, rng.integers(-1, 16, n)))
    outs = [[torch.full((len(s_[0]), 65, 26, 2), float("nan"), device=dev) for s_ in steps] for _ in ctxs]
    side = torch.cuda.Stream(device=dev)
    for which, c in enumerate(ctxs):
        with torch.cuda.stream(side if which else torch.cuda.current_stream()):
            for k, (sound, t0, rir) in enumerate(steps):
                if k == 33:
                    with pytest.raises(SsHipError):
                        c.observe([99], [0], [0], spectrogram_out=outs[which][k][:1])     # refused: lanes keep their turn
                c.observe(sound, t0, rir, spectrogram_out=outs[which][k])
            c.join()
    torch.cuda.synchronize()
    for k in range(len(steps)):
        assert torch.equal(outs[1][k], outs[0][k]), k
    st = ctxs[1].stats()
    assert st["misses"] > 10 and st["hits"] > 100
    ctxs[1].set_overlap(1)                                                       # and back: same results again
    k = 5
    o = torch.empty_like(outs[0][k])
    ctxs[1].observe(*steps[k], spectrogram_out=o)
    torch.cuda.synchronize()
    assert torch.equal(o, outs[0][k])


@pytest.mark.gpu
def test_ctx_observe_matches_renderer_and_oracle():
    """ss_ctx_observe (planner + cache + ring in the library) renders the same step as the Python-planned renderer,
    bit for bit, over several steps with cache hits, evictions and a growing batch; spot-checked against the oracle."""
    import torch
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    dev = "cuda:0"
    rng = np.random.default_rng(8)
    src = [O.synth_sources(rng, SR, k=1, seconds=s)[0] for s in (1, 1, 1, 3, 5)]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=12)]
    bank = RirBank.from_arrays(rirs, dev)
    r = BatchedAudioRenderer(SR, device=dev)
    ctx = AudioContext(SR, max_window_sets=8)
    for i, s in enumerate(src):
        r.add_source(f"s{i}", s)
        ctx.add_source(f"s{i}", s)
    r.set_rir_bank(bank)
    ctx.set_rir_bank(bank.data, bank.lengths)
    for step, n in enumerate((5, 33, 128, 7)):
        sound = rng.integers(0, 5, n)
        t0 = np.array([0 if len(src[s]) == SR else rng.integers(0, len(src[s]) // SR) * SR for s in sound])
        rir = rng.integers(-1, 12, n)
        sg = torch.empty((n, 65, 26, 2), device=dev)
        ag = torch.empty((n, 2, SR), device=dev)
        ctx.observe(sound, t0, rir, spectrogram_out=sg, audiogoal_out=ag)
        ag2, sg2 = r.render(r.plan_arrays(sound, t0, rir), want_audiogoal=True)
        assert torch.equal(sg, sg2) and torch.equal(ag, ag2)
        sg3 = torch.empty_like(sg)
        ctx.observe(sound, t0, rir, spectrogram_out=sg3)                          # spectrogram only (fused, no waveform)
        assert float((sg3 - sg).abs().max()) <= 1e-5 * float(sg.abs().max())
        k = int(np.flatnonzero(rir >= 0)[0])
        ref = O.conv_window_fft(src[sound[k]], rirs[rir[k]], int(t0[k]), SR)
        assert O.relerr(ag[k].cpu().numpy(), ref) <= 1e-4
    assert ctx.stats()["hits"] > 0


@pytest.mark.gpu
def test_ctx_ring_many_steps_in_flight_and_stream_switches():
    """The descriptor ring of ss_ctx_observe: 70 steps queued WITHOUT a host sync in between (several ring revolutions;
    small steps read their descriptors in place from pinned memory, the 300-unit steps upload them), the caller hopping
    between two streams at irregular points (a group of slots is closed early on a stream switch); every step's output
    is checked afterwards against the Python-planned renderer."""
    import torch
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    dev = "cuda:0"
    rng = np.random.default_rng(21)
    src = [O.synth_sources(rng, SR, k=1, seconds=s)[0] for s in (1, 1, 3)]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=16)]
    bank = RirBank.from_arrays(rirs, dev)
    r = BatchedAudioRenderer(SR, device=dev)
    ctx = AudioContext(SR, max_window_sets=8)
    for i, s in enumerate(src):
        r.add_source(f"s{i}", s)
        ctx.add_source(f"s{i}", s)
    r.set_rir_bank(bank)
    ctx.set_rir_bank(bank.data, bank.lengths)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    steps, outs, which = [], [], 0
    torch.cuda.synchronize()
    for k in range(70):
        n = 300 if k % 9 == 4 else int(rng.integers(1, 40))
        sound = rng.integers(0, 3, n)
        t0 = np.array([0 if len(src[s]) == SR else rng.integers(0, 3) * SR for s in sound])
        rir = rng.integers(-1, 16, n)
        if k in (2, 3, 11, 29, 30, 31, 50):
            which ^= 1
        sg = torch.empty((n, 65, 26, 2), device=dev)
        with torch.cuda.stream(streams[which]):
            ctx.observe(sound, t0, rir, spectrogram_out=sg)
        steps.append((sound, t0, rir))
        outs.append(sg)
    torch.cuda.synchronize()
    for (sound, t0, rir), sg in zip(steps, outs):
        _, ref = r.render(r.plan_arrays(sound, t0, rir))
        assert torch.equal(sg, ref)


@pytest.mark.gpu
def test_ctx_observe_distractor_crossfade_and_44k():
    import torch
    from ss_amd.renderer import BatchedAudioRenderer, RirBank, UnitRequest
    dev = "cuda:0"
    rng = np.random.default_rng(9)
    # distractors (SS1.0)
    src = list(O.synth_sources(rng, SR, k=3))
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=6)]
    bank = RirBank.from_arrays(rirs, dev)
    ctx = AudioContext(SR)
    for i, s in enumerate(src):
        ctx.add_source(f"s{i}", s)
    ctx.set_rir_bank(bank.data, bank.lengths)
    n = 9
    sound, rir = rng.integers(0, 3, n), rng.integers(0, 6, n)
    ds, dr = rng.integers(0, 3, n), rng.integers(-1, 6, n)
    ag = torch.empty((n, 2, SR), device=dev)
    ctx.observe(sound, np.zeros(n), rir, audiogoal_out=ag, dis_sound=ds, dis_rir=dr)
    for i in range(n):
        ref = O.compute_audiogoal(src[sound[i]], rirs[rir[i]], SR, distractor=src[ds[i]] if dr[i] >= 0 else None,
                                  distractor_rir=rirs[dr[i]] if dr[i] >= 0 else None)
        assert O.relerr(ag[i].cpu().numpy(), ref) <= 1e-4
    # SS2.0 cross-fade
    src3 = O.tile_short_source(src[0], SR)
    rl = [np.ascontiguousarray(O.synth_rir(rng, SR, length=L, n=1)[0].T) for L in (9000, 12000, 20000)]
    bank2 = RirBank.from_arrays(rl, dev)
    c2 = AudioContext(SR, step_time=0.25, wrap=True)
    c2.add_source("s", src3)
    c2.set_rir_bank(bank2.data, bank2.lengths)
    idx = np.array([100, 15000, 30000, 46000, 47000])
    cur, last = np.array([0, 1, 2, 0, 1]), np.array([1, 2, 0, -1, 2])
    Ls = np.array([9000, 12000, 20000])
    ag = torch.empty((5, 2, SR), device=dev)
    sg = torch.empty((5, 65, 26, 2), device=dev)
    c2.observe(np.zeros(5), idx, cur, spectrogram_out=sg, audiogoal_out=ag, last_rir=last,
               wrap=(idx >= Ls[cur]).astype(np.uint8), last_wrap=(idx >= Ls[np.maximum(last, 0)]).astype(np.uint8))
    for i in range(5):
        ref = O.compute_audiogoal_continuous(src3, rl[cur[i]], SR, int(idx[i]), 0.25,
                                             last_rir=rl[last[i]] if last[i] >= 0 else None, use_crossfade=True)
        assert O.relerr(ag[i].cpu().numpy(), ref) <= 1e-4
        assert O.relerr(sg[i].cpu().numpy(), O.compute_spectrogram(ref.astype(np.float32))) <= 1e-4
    # 44.1 kHz, spectrogram only: the hand-over buffer is the context's own
    sr = 44100
    s44 = O.synth_sources(rng, sr, k=1)[0]
    r44 = [np.ascontiguousarray(O.synth_rir(rng, sr, n=1)[0].T)]
    b44 = RirBank.from_arrays(r44, dev)
    c3 = AudioContext(sr)
    c3.add_source("s", s44)
    c3.set_rir_bank(b44.data, b44.lengths)
    sg = torch.empty((2, 65, 69, 2), device=dev)
    c3.observe(np.zeros(2), np.zeros(2), np.array([0, -1]), spectrogram_out=sg)
    a = O.compute_audiogoal(s44, r44[0], sr)
    assert O.relerr(sg[0].cpu().numpy(), O.compute_spectrogram(a)) <= 1e-4 and not sg[1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 2])
def test_observe_features_equals_observe_then_feature_kernels(overlap):
    """ss_ctx_observe_features (BASELINE.json configs[4]: savi with the fused GCC-PHAT + log-mel sensor): the step and its
    extension features on one stream - in overlap mode on the step's own lane, no join between them - equal
    ss_ctx_observe followed by the stand-alone feature kernels; a plain step on the same context afterwards is unaffected."""
    import torch
    from ss_amd import ops, planning as P
    from ss_amd.renderer import RirBank
    dev = "cuda:0"
    rng = np.random.default_rng(31)
    src = list(O.synth_sources(rng, SR, k=3)) + [O.synth_sources(rng, SR, k=1, seconds=3)[0]]
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, SR, n=6)]
    bank = RirBank.from_arrays(rirs, dev)
    ctx = AudioContext(SR)
    for i, s_ in enumerate(src):
        ctx.add_source(f"s{i}", s_)
    ctx.set_rir_bank(bank.data, bank.lengths)
    n = 9
    cols = [dict(sound=rng.integers(0, 4, n), t0=np.zeros(n, np.int64), rir=rng.integers(0, 6, n),
                 dis_sound=rng.integers(0, 3, n), dis_rir=rng.integers(0, 6, n)) for _ in range(4)]
    for c in cols:
        c["t0"] = np.where(c["sound"] == 3, SR * rng.integers(0, 3, n), 0)
        c["rir"][2] = -1                                                        # a silent unit
    ms, mw, _ = P.mel_filterbank_sparse(SR, 64)
    msd, mwd = torch.from_numpy(ms).to(dev), torch.from_numpy(mw).to(dev)
    T = 1 + SR // 160
    want = []
    for c in cols:                                                             # the reference order: observe, then the kernels
        ag, sg = torch.empty((n, 2, SR), device=dev), torch.empty((n, 65, 26, 2), device=dev)
        ctx.observe(spectrogram_out=sg, audiogoal_out=ag, **c)
        want.append((ag, sg, ops.logmel(ag, msd, mwd), ops.gccphat(ag)))
    torch.cuda.synchronize()
    ctx.set_overlap(overlap)
    stream = torch.cuda.current_stream().cuda_stream
    got = []
    for c in cols:
        ag, sg = torch.full((n, 2, SR), 9.0, device=dev), torch.full((n, 65, 26, 2), 9.0, device=dev)
        lm, gc = torch.full((n, 64, T, 2), 9.0, device=dev), torch.full((n, 65, T), 9.0, device=dev)
        f = ctx.features(lm, msd, mwd, 1e-6, gc, 32, 1e-8)
        ctx.observe_prepared_features(ctx.prepare(**c), sg.data_ptr(), ag.data_ptr(), stream, f)
        got.append((ag, sg, lm, gc, f))
    ctx.join()
    torch.cuda.synchronize()
    for (ag, sg, lm, gc, _), (ag0, sg0, lm0, gc0) in zip(got, want):
        assert torch.equal(ag, ag0) and torch.equal(sg, sg0)
        assert float((lm - lm0).abs().max()) <= 5e-5 and float((gc - gc0).abs().max()) <= 5e-6
        assert not ag[2].any() and torch.allclose(lm[2], torch.full_like(lm[2], float(np.log(1e-6))), rtol=1e-6)
    # ... and the ORACLE on what this entry point wrote (the entry cfg[4]'s bench line times): per stage - the oracle's
    # features of the waveform the kernels rendered - at the north star's 1e-4, and END TO END - oracle waveform
    # (simulator.py:629-664 restated) -> oracle feature - with the bound recorded here: the convolution's 1e-6 through log / PHAT
    ag, sg, lm, gc, _ = got[0]
    ag_np, c = ag.cpu().numpy(), cols[0]
    worst = {"logmel": 0.0, "gccphat": 0.0, "logmel_e2e": 0.0, "gccphat_e2e": 0.0, "spectrogram_e2e": 0.0}
    for i in range(n):
        if c["rir"][i] < 0:
            continue
        lm_i, gc_i = lm[i].cpu().numpy(), gc[i].cpu().numpy()
        ref_lm, ref_gc = O.compute_logmel(ag_np[i], SR), O.compute_gcc_phat(ag_np[i])
        worst["logmel"] = max(worst["logmel"], float(np.abs(lm_i - ref_lm).max() / np.abs(ref_lm).max()))
        worst["gccphat"] = max(worst["gccphat"], float(np.abs(gc_i - ref_gc).max()))
        s_i = int(c["sound"][i])
        a = O.compute_audiogoal(src[s_i], rirs[int(c["rir"][i])], SR, audio_index=int(c["t0"][i]) // SR,
                                distractor=src[int(c["dis_sound"][i])], distractor_rir=rirs[int(c["dis_rir"][i])]).astype(np.float32)
        assert O.relerr(ag_np[i], a) <= 1e-4
        e_lm, e_gc = O.compute_logmel(a, SR), O.compute_gcc_phat(a)
        worst["logmel_e2e"] = max(worst["logmel_e2e"], float(np.abs(lm_i - e_lm).max() / np.abs(e_lm).max()))
        worst["gccphat_e2e"] = max(worst["gccphat_e2e"], float(np.abs(gc_i - e_gc).max()))
        worst["spectrogram_e2e"] = max(worst["spectrogram_e2e"], O.relerr(sg[i].cpu().numpy(), O.compute_spectrogram(a)))
    print("ss_ctx_observe_features vs oracle:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert worst["logmel"] <= 1e-4 and worst["gccphat"] <= 1e-4 and worst["spectrogram_e2e"] <= 1e-4
    # end to end as well (measured on the MI355X: log-mel 1.5e-5, GCC-PHAT 5.7e-6, spectrogram 4.9e-7; per stage 1.1e-6 / 5.7e-6)
    assert worst["logmel_e2e"] <= 1e-4 and worst["gccphat_e2e"] <= 1e-4
    ctx.set_overlap(1)
    sg = torch.empty((n, 65, 26, 2), device=dev)
    ctx.observe(spectrogram_out=sg, **cols[0])
    torch.cuda.synchronize()
    assert torch.equal(sg, want[0][1])
    with pytest.raises(Exception):                                             # the features read the waveform: no audiogoal, no call
        ctx.observe_prepared_features(ctx.prepare(**cols[0]), sg.data_ptr(), None, stream, got[0][4])


def test_two_sample_rir_is_read_as_wav_layout():
    """A [2, 2] array is ambiguous between the wav layout [L, 2] and the planar [2, L]; `wavfile.read` (simulator.py:615)
    returns the wav layout, so that is what a 2 x 2 array means (found by scripts/gpu_fuzz.py: a two-tap RIR came out
    transposed)."""
    from ss_amd.renderer import RirBank, _planar
    h = np.array([[1.0, 2.0], [3.0, 4.0]], np.float32)               # sample 0 = (L 1, R 2), sample 1 = (L 3, R 4)
    bank = RirBank.from_arrays([h, np.arange(6, dtype=np.float32).reshape(3, 2), np.arange(6, dtype=np.float32).reshape(2, 3)], "cpu")
    assert bank.data[0, :, :2].tolist() == [[1.0, 3.0], [2.0, 4.0]]
    assert bank.data[1, :, :3].tolist() == [[0.0, 2.0, 4.0], [1.0, 3.0, 5.0]]
    assert bank.data[2, :, :3].tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    assert bank.lengths.tolist() == [2, 3, 3]
    assert _planar(h).tolist() == [[1.0, 3.0], [2.0, 4.0]]
