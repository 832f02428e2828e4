"""The library's native RIR-file reader (ss_wav_read_rirs_f32, csrc/ss_wavio.hpp) and ``RirStore.load_files`` against
scipy.io.wavfile.read, the reader the reference uses on every cache miss (soundspaces/simulator.py:615-624): same samples
for float32 stereo files, the reference's zero-RIR fallbacks for unreadable / empty files, everything unusual handed to
the Python reader.  Host-only code: runs without a GPU (CPU store)."""
import os

import numpy as np
import pytest
from scipy.io import wavfile

from ss_amd import _lib
from ss_amd.renderer import RirStore, load_scene_rirs
from ss_amd.sim_audio import wav_rir_reader


@pytest.fixture(scope="module")
def wavs(tmp_path_factory):
    d = tmp_path_factory.mktemp("rirs")
    rng = np.random.default_rng(0)
    files = {}
    for name, L in [("a", 16000), ("ragged", 12345), ("one", 1), ("empty", 0), ("long", 40000)]:
        p = str(d / f"{name}.wav")
        wavfile.write(p, 16000, rng.standard_normal((L, 2)).astype(np.float32))
        files[name] = p
    files["i16"] = str(d / "i16.wav")
    wavfile.write(files["i16"], 16000, (rng.standard_normal((100, 2)) * 1000).astype(np.int16))
    files["mono"] = str(d / "mono.wav")
    wavfile.write(files["mono"], 16000, rng.standard_normal((100,)).astype(np.float32))
    files["junk"] = str(d / "junk.wav")
    open(files["junk"], "wb").write(b"hello world, this is not a wav file at all")
    files["trunc"] = str(d / "trunc.wav")
    open(files["trunc"], "wb").write(open(files["a"], "rb").read()[:5000])
    # a LIST chunk between fmt and data (ffmpeg writes one), odd-sized: must be skipped with its pad byte
    raw = open(files["ragged"], "rb").read()
    i = raw.index(b"data")
    extra = b"LIST" + (5).to_bytes(4, "little") + b"INFOx" + b"\0"
    body = raw[:i] + extra + raw[i:]
    body = body[:4] + (len(body) - 8).to_bytes(4, "little") + body[8:]
    files["list"] = str(d / "list.wav")
    open(files["list"], "wb").write(body)
    # the data chunk cut short BEHIND the frames a clipped read keeps (keep = 16000 of 40000): still not a file scipy reads
    files["long_trunc"] = str(d / "long_trunc.wav")
    open(files["long_trunc"], "wb").write(open(files["long"], "rb").read()[:200000])
    # a tiny file: header and samples inside the reader's first 4-KiB read
    files["tiny"] = str(d / "tiny.wav")
    wavfile.write(files["tiny"], 16000, rng.standard_normal((300, 2)).astype(np.float32))
    files["missing"] = str(d / "missing.wav")
    return files


def test_native_reader_equals_scipy_and_reports_what_it_does_not_read(wavs):
    names = list(wavs)
    paths = [wavs[n] for n in names]
    cap = 16384
    for planar in (False, True):
        dst = np.full((len(paths), 2, cap) if planar else (len(paths), cap, 2), 7.0, np.float32)
        kept, frames, status = _lib.wav_read_rirs(paths, dst, cap, keep=16000, planar=planar, threads=3)
        st = dict(zip(names, status))
        assert [st[n] for n in ("a", "ragged", "one", "list", "tiny")] == [_lib.WAV_OK] * 5
        assert st["long_trunc"] == _lib.WAV_UNSUPPORTED
        assert st["empty"] == _lib.WAV_EMPTY and st["missing"] == _lib.WAV_MISSING
        assert st["i16"] == st["mono"] == st["junk"] == st["trunc"] == _lib.WAV_UNSUPPORTED
        assert st["long"] == _lib.WAV_OK and dict(zip(names, kept))["long"] == 16000 and dict(zip(names, frames))["long"] == 40000
        for i, n in enumerate(names):
            row = dst[i].T if planar else dst[i]
            if status[i] == _lib.WAV_OK:
                _, ref = wavfile.read(paths[i])
                assert np.array_equal(row[:kept[i]], ref[:kept[i]]) and not row[kept[i]:].any()
            else:
                assert not row.any() and kept[i] == 0
    # a row too short for what is to be kept: reported, nothing read
    dst = np.zeros((1, 1000, 2), np.float32)
    kept, frames, status = _lib.wav_read_rirs([wavs["a"]], dst, 1000, keep=-1)
    assert status[0] == _lib.WAV_TOO_LONG and frames[0] == 16000 and not dst.any()


@pytest.mark.parametrize("truncate_to", [16000, None])
def test_store_load_files_equals_the_python_reader_path(wavs, truncate_to):
    names = ["a", "ragged", "one", "empty", "long", "i16", "junk", "list", "a"]
    paths = [wavs[n] for n in names]
    s1 = RirStore(16, 16000, "cpu", truncate_to=truncate_to)
    s2 = RirStore(16, 16000, "cpu", truncate_to=truncate_to)
    got = s1.load_files(paths, paths)
    ref = s2.slot_many(paths, [(lambda p=p: wav_rir_reader(p)) for p in paths], workers=1)
    assert got[0] == got[-1] and len(set(got)) == len(names) - 1
    assert s1.cap == s2.cap and (truncate_to is not None or s1.cap >= 40000)
    for a, b in zip(got, ref):
        assert np.array_equal(s1.bank.data[a].numpy(), s2.bank.data[b].numpy())
        assert s1.host_len[a] == s2.host_len[b] and int(s1.bank.lengths[a]) == int(s2.bank.lengths[b])
        assert s1._clipped[a] == s2._clipped[b]
    # integer PCM keeps scipy's values (the reference would convolve them as they are), junk is the zero RIR
    i16 = got[names.index("i16")]
    assert s1.host_len[i16] == 100 and abs(s1.bank.data[i16]).max() > 10
    assert s1.host_len[got[names.index("junk")]] == 0 and s1.host_len[got[names.index("empty")]] == 0
    # hits the second time
    h = s1.hits
    assert s1.load_files(paths, paths) == got and s1.hits == h + len(paths)
    with pytest.raises(FileNotFoundError):
        s1.load_files([wavs["missing"]], [wavs["missing"]])
    assert wavs["missing"] not in s1._slot_of
    sl = s1.load_files([wavs["missing"]], [wavs["missing"]], missing_ok=True)[0]
    assert s1.host_len[sl] == 0 and not s1.bank.data[sl].any()


def test_store_load_files_azimuth_groups_and_eviction(wavs):
    s1 = RirStore(8, 16000, "cpu", truncate_to=16000, group=4)
    files = [[wavs["a"], None, wavs["ragged"], wavs["one"]], [wavs["ragged"], wavs["a"], wavs["a"], wavs["empty"]]]
    base = s1.load_files(["p0", "p1"], files)
    for b, fl in zip(base, files):
        for g, f in enumerate(fl):
            ref = wav_rir_reader(f) if f else None
            n = 0 if ref is None else min(16000, ref.shape[0])
            assert s1.host_len[b + g] == n
            if n:
                assert np.array_equal(s1.bank.data[b + g].numpy()[:, :n], ref[:n].T)
            assert not s1.bank.data[b + g].numpy()[:, n:].any()
    evicted = []
    s1.on_evict = lambda key, slot: evicted.append(key)
    b2 = s1.load_files(["p2"], [[wavs["a"]] * 4])[0]
    assert evicted == ["p0"] and b2 == base[0]                  # LRU: p0 is the oldest entry
    with pytest.raises(ValueError):
        s1.load_files(["x", "y", "z"], [[wavs["a"]] * 4] * 3)


def test_load_scene_rirs_takes_the_native_reader_for_the_stock_reader(wavs, tmp_path):
    rng = np.random.default_rng(1)
    for az in (0, 90):
        os.makedirs(tmp_path / str(az))
        for r in range(3):
            wavfile.write(str(tmp_path / str(az) / f"{r}_0.wav"), 16000, rng.standard_normal((500 + r, 2)).astype(np.float32))
    s1 = RirStore(8, 16000, "cpu", truncate_to=16000)
    n = load_scene_rirs(s1, str(tmp_path), wav_rir_reader, azimuths=(0, 90))
    assert n == 6 and s1.misses == 6
    for az in (0, 90):
        for r in range(3):
            p = os.path.join(str(tmp_path), str(az), f"{r}_0.wav")
            sl = s1.slot(p, lambda: pytest.fail("resident"))
            assert np.array_equal(s1.bank.data[sl].numpy()[:, :500 + r], wav_rir_reader(p).T)


def test_bucketed_store_load_files_routes_by_probed_length(wavs):
    """BucketedRirStore.load_files: frame counts probed by the library's reader (header only), every file into the smallest
    bucket that holds it, same rows / lengths as the Python-reader path (slot_many)."""
    from ss_amd.renderer import BucketedRirStore
    names = ["a", "ragged", "long", "empty", "i16", "junk", "tiny", "a"]
    paths = [wavs[n] for n in names]
    s1 = BucketedRirStore([8, 4], [16000, 65536], "cpu", truncate_to=None)
    s2 = BucketedRirStore([8, 4], [16000, 65536], "cpu", truncate_to=None)
    got = s1.load_files(paths, paths)
    ref = s2.slot_many(paths, [(lambda p=p: wav_rir_reader(p)) for p in paths], workers=1)
    assert got[0] == got[-1] and ref[0] == ref[-1]
    assert s1.bank.bucket_of(got[names.index("long")]) == 1 and s1.bank.bucket_of(got[names.index("a")]) == 0
    for g, r_ in zip(got, ref):                                 # (slot numbers may differ: the order of loading does)
        b = s1.bank.bucket_of(g)
        assert b == s2.bank.bucket_of(r_)
        assert np.array_equal(s1.stores[b].bank.data[g - s1.first[b]].numpy(), s2.stores[b].bank.data[r_ - s2.first[b]].numpy())
        assert s1.host_len[g] == s2.host_len[r_]
    assert s1.load_files(paths, paths) == got                  # hits


@pytest.mark.gpu
@pytest.mark.parametrize("from_host", [True, False])
def test_bank_scatter_rows_equals_a_numpy_scatter(from_host):
    """ss_bank_scatter_rows_f32 (k_scatter_rows): staged wav-layout rows -> planar bank rows + the length table, bit for
    bit, staged block / slots / lengths in pinned host memory (the kernel reads them over the host link) or on the device;
    odd capacities and strides (scalar path), zero-length rows, bank_len = NULL."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for cap, pad in ((16000, 0), (9001, 0), (9000, 6), (513, 1)):
        n, R = 7, 12
        stride = 2 * cap + pad
        staged = torch.zeros((n, stride), dtype=torch.float32, pin_memory=True)
        lens = np.array([cap, 0, 1, cap - 1, cap // 2, 2, 3][:n], np.int32)
        snp = staged.numpy()
        for i, L in enumerate(lens):
            snp[i, :2 * L] = rng.standard_normal(2 * L).astype(np.float32)
        slots = rng.permutation(R)[:n].astype(np.int32)
        t_slots = torch.from_numpy(slots.copy()).pin_memory()
        t_lens = torch.from_numpy(lens.copy()).pin_memory()
        bank = torch.full((R, 2, cap), 7.0, device=dev)
        bank_len = torch.full((R,), -5, dtype=torch.int32, device=dev)
        src, sl_t, ln_t = (staged, t_slots, t_lens) if from_host else (staged.to(dev), t_slots.to(dev), t_lens.to(dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
        lib = _lib.load()
        _lib.check(lib.ss_bank_scatter_rows_f32(src.data_ptr(), stride, sl_t.data_ptr(), ln_t.data_ptr(), n, bank.data_ptr(),
                                                bank.stride(0), bank.stride(1), cap, bank_len.data_ptr(), stream), "scatter")
        torch.cuda.synchronize()
        want = np.full((R, 2, cap), 7.0, np.float32)
        want_len = np.full((R,), -5, np.int32)
        for i, (sl, L) in enumerate(zip(slots, lens)):
            want[sl] = 0.0
            want[sl, :, :L] = snp[i, :2 * L].reshape(L, 2).T
            want_len[sl] = L
        assert np.array_equal(bank.cpu().numpy(), want) and np.array_equal(bank_len.cpu().numpy(), want_len)
        _lib.check(lib.ss_bank_scatter_rows_f32(src.data_ptr(), stride, sl_t.data_ptr(), ln_t.data_ptr(), n, bank.data_ptr(),
                                                bank.stride(0), bank.stride(1), cap, None, stream), "scatter")
        torch.cuda.synchronize()
        assert np.array_equal(bank.cpu().numpy(), want)
        assert lib.ss_bank_scatter_rows_f32(src.data_ptr(), 2 * cap - 2, sl_t.data_ptr(), ln_t.data_ptr(), n, bank.data_ptr(),
                                            bank.stride(0), bank.stride(1), cap, None, stream) == -1
        # pageable host memory would be a GPU memory fault inside the kernel: refused, nothing launched
        pageable = np.zeros((n, stride), np.float32)
        assert lib.ss_bank_scatter_rows_f32(pageable.ctypes.data, stride, sl_t.data_ptr(), ln_t.data_ptr(), n, bank.data_ptr(),
                                            bank.stride(0), bank.stride(1), cap, None, stream) == -1
        assert lib.ss_bank_scatter_rows_f32(src.data_ptr(), stride, slots.ctypes.data, ln_t.data_ptr(), n, bank.data_ptr(),
                                            bank.stride(0), bank.stride(1), cap, None, stream) == -1
        torch.cuda.synchronize()
        assert np.array_equal(bank.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("from_host", [True, False])
def test_gpu_store_loads_files_and_live_rows_like_the_host_store(wavs, from_host):
    """RirStore.load_files / upload_rows on a GPU store (one launch of the library's scatter per block) leave the bank the
    host store's torch scatter leaves: rows, lengths, clipping flags - also after a second block reuses the staging memory."""
    import torch
    names = ["a", "ragged", "one", "empty", "long", "i16", "junk", "list", "tiny"]
    paths = [wavs[n] for n in names]
    for truncate_to in (16000, None):
        g = RirStore(16, 16000, "cuda:0", truncate_to=truncate_to)
        g.scatter_from_host = from_host
        h = RirStore(16, 16000, "cpu", truncate_to=truncate_to)
        sg, sh = g.load_files(paths, paths), h.load_files(paths, paths)
        sg2, sh2 = g.load_files(paths[::-1][:4] + ["x"], paths[::-1][:4] + [wavs["a"]]), h.load_files(paths[::-1][:4] + ["x"], paths[::-1][:4] + [wavs["a"]])
        torch.cuda.synchronize()
        assert g.cap == h.cap
        for a, b in zip(sg + sg2, sh + sh2):
            assert np.array_equal(g.bank.data[a].cpu().numpy(), h.bank.data[b].numpy())
            assert int(g.bank.lengths[a]) == int(h.bank.lengths[b]) == g.host_len[a] == h.host_len[b]
            assert g._clipped[a] == h._clipped[b]
    rng = np.random.default_rng(9)
    g, h = RirStore(8, 4000, "cuda:0"), RirStore(8, 4000, "cpu")
    g.scatter_from_host = from_host
    for rep in range(3):
        rows = [rng.standard_normal((int(L), 2)).astype(np.float32) for L in rng.integers(1, 9000, 5)]
        slots = rng.permutation(8)[:5].tolist()
        g.upload_rows(slots, rows)
        h.upload_rows(slots, rows)
        torch.cuda.synchronize()
        assert g.cap == h.cap and np.array_equal(g.bank.data.cpu().numpy(), h.bank.data.numpy())
        assert np.array_equal(g.bank.lengths.cpu().numpy(), h.bank.lengths.numpy()) and np.array_equal(g.host_len, h.host_len)


def test_upload_rows_takes_read_only_rows_and_routes_odd_layouts_row_by_row():
    """``RirStore.upload_rows`` reads the rows' addresses through a ctypes view of their buffers (a quarter of the cost of
    ``__array_interface__`` at 128 rows per step); read-only arrays (``np.frombuffer`` of a pipe's bytes) have no such view
    and take the slow way; a step with a row in another layout (float64, planar, a strided view) goes row by row through
    ``_upload``.  Same bank either way."""
    rng = np.random.default_rng(3)
    base = [rng.standard_normal((int(L), 2)).astype(np.float32) for L in (100, 1, 777, 1500)]
    ro = [np.frombuffer(r.tobytes(), np.float32).reshape(r.shape) for r in base]
    assert not ro[0].flags.writeable
    odd = [base[0].astype(np.float64), np.ascontiguousarray(base[1].T), base[2], rng.standard_normal((1500, 4)).astype(np.float32)[:, :2]]
    want = [base[0], base[1], base[2], odd[3]]
    for rows, ref in ((base, base), (ro, base), (odd, want)):
        a, b = RirStore(8, 2000, "cpu"), RirStore(8, 2000, "cpu")
        a.upload_rows([5, 0, 3, 6], rows)
        for sl, r in zip([5, 0, 3, 6], ref):
            b._upload(sl, r)
        b.flush_uploads()
        assert np.array_equal(a.bank.data.numpy(), b.bank.data.numpy())
        assert np.array_equal(a.bank.lengths.numpy(), b.bank.lengths.numpy()) and np.array_equal(a.host_len, b.host_len)


def test_a_file_that_claims_more_frames_than_it_holds_never_grows_the_bank(wavs, tmp_path):
    """ whole-RIR mode: a corrupt header (3000 frames on disk, 200000 claimed) must reach the Python reader as
    'unsupported' - not make every slot of the store 200000 frames long before anyone has looked at the file."""
    raw = bytearray(open(wavs["a"], "rb").read())
    i = raw.index(b"data")
    raw[i + 4:i + 8] = (200000 * 8).to_bytes(4, "little")
    p = str(tmp_path / "liar.wav")
    open(p, "wb").write(bytes(raw[:i + 8 + 3000 * 8]))
    dst = np.zeros((1, 1000, 2), np.float32)
    kept, frames, status = _lib.wav_read_rirs([p], dst, 1000, keep=-1)
    assert status[0] == _lib.WAV_UNSUPPORTED and kept[0] == 0 and not dst.any()
    st = RirStore(8, 4000, "cpu", truncate_to=None, max_cap=1 << 20)
    st.load_files([("liar",)], [p])                     # the scipy fallback decides (a lenient read or the zero RIR) ...
    assert st.cap < 200000                              # ... and the rows never grow to the claimed length


def test_the_reader_pool_survives_fork(wavs):
    """ the persistent thread pool belongs to the process that built it: a fork()ed child (multiprocessing 'fork',
    as bench.py's CPU baseline uses) gets its own instead of waiting for threads that do not exist there."""
    import multiprocessing as mp
    paths = [wavs["a"], wavs["ragged"], wavs["tiny"], wavs["a"]]
    dst = np.zeros((len(paths), 16384, 2), np.float32)
    _lib.wav_read_rirs(paths, dst, 16384, keep=16000, threads=4)          # the parent's pool has workers now
    q = mp.get_context("fork").Queue()

    def child():
        d = np.zeros((len(paths), 16384, 2), np.float32)
        kept, _, status = _lib.wav_read_rirs(paths, d, 16384, keep=16000, threads=4)
        q.put((kept.tolist(), status.tolist(), float(np.abs(d - dst).max())))
    pr = mp.get_context("fork").Process(target=child)
    pr.start()
    pr.join(30)
    assert not pr.is_alive(), "the forked child hung in the reader pool"
    kept, status, diff = q.get(timeout=5)
    assert status == [_lib.WAV_OK] * 4 and diff == 0.0 and kept[0] == 16000


@pytest.mark.gpu
def test_pose_misses_are_served_inside_the_step_s_call(tmp_path):
    """Round 6 (ss_ctx_observe_requests_load): a deferred-mode step that meets poses whose RIR file is not resident is ONE C call -
    file names built, files read by the library's reader into the store's pinned block, one scatter launch into free entries, the
    resident-pair arrays extended in place, the step launched - and the store's / resolver's own tables follow from the report.
    Against the three-call path (report -> load_files -> call again) and the oracle: same observations; what the fast path does
    not cover (an int16 file: scipy's semantics) falls back and still agrees; a store smaller than the walk evicts inside the call."""
    import pickle
    import types
    import torch
    from oracle import ss_oracle as O
    from ss_amd.deferred import DeferredResolver, attach_deferred
    from ss_amd.renderer import AudioEngine
    NS = types.SimpleNamespace
    sr, n_nodes, n_env = 16000, 6, 8
    rng = np.random.default_rng(11)
    root = tmp_path / "rirs"
    rirs = {}
    for az in (0, 90):
        (root / str(az)).mkdir(parents=True)
        for r in range(n_nodes):
            for s_ in range(n_nodes):
                L = int(rng.integers(2000, 16001))
                h = O.synth_rir(np.random.default_rng(100 * az + 10 * r + s_), sr, length=L, n=1)[0]
                p = str(root / str(az) / f"{r}_{s_}.wav")
                if (az, r, s_) == (90, 5, 5):                    # one file scipy reads but the library's reader hands back
                    wavfile.write(p, sr, (h.T * 20000).astype(np.int16))
                    rirs[p] = (h.T * 20000).astype(np.int16).astype(np.float32)
                else:
                    wavfile.write(p, sr, np.ascontiguousarray(h.T))
                    rirs[p] = np.ascontiguousarray(h.T)
    clip = O.synth_sources(np.random.default_rng(5), sr, k=1)[0]

    class Sim:
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = str(root)
        _source_sound_dict = {"s.wav": clip}
        _current_sound, _audio_index, _episode_step_count, _duration = "s.wav", 0, 0, 500
        _receiver_position_index = _source_position_index = 0
        azimuth_angle = 0
        current_source_sound = property(lambda self: clip)
        _audio_length = 1

    def run(in_call, slots):
        sims = [Sim() for _ in range(n_env)]
        for i, sm in enumerate(sims):
            attach_deferred(sm, env_rank=i)
        res = DeferredResolver(AudioEngine(sr, device="cuda:0", rir_slots=slots), fast=True, prefetch_azimuths=False)
        res.native_miss_path = in_call
        outs = []
        walk = np.random.default_rng(3)
        for k in range(6):
            for sm in sims:
                sm._receiver_position_index, sm._source_position_index = int(walk.integers(0, n_nodes)), int(walk.integers(0, n_nodes))
                sm.azimuth_angle = int(walk.choice([0, 90]))
                sm._episode_step_count += 1
            if k == 4:
                sims[0]._receiver_position_index = sims[0]._source_position_index = 5
                sims[0].azimuth_angle = 90                       # the int16 file
            reqs = [pickle.loads(pickle.dumps(sm.get_current_spectrogram_observation(None))) for sm in sims]
            out = res.resolve(reqs, want_audiogoal=True)
            torch.cuda.synchronize()
            outs.append((out["audiogoal"].cpu().numpy(), out["spectrogram"].cpu().numpy(),
                         [os.path.join(str(root), str(sm.azimuth_angle), f"{sm._receiver_position_index}_{sm._source_position_index}.wav") for sm in sims]))
        return res, outs

    res_a, a = run(True, 256)
    res_b, b = run(False, 256)
    assert res_a.library_loaded >= 20 and res_b.library_loaded == 0
    store = res_a.engine.store
    assert len(store._slot_of) == store.misses and len(store._free) == store.slots - len(store._slot_of)
    for (key, slot) in store._slot_of.items():                   # the store's books agree with what the library wrote
        assert key[0] == "ix" and store._key_at[slot] == key and store._used[slot]
        k = key[1]
        path = os.path.join(res_a._table_dirs[k >> 40], f"{(k >> 20) & 0xFFFFF}_{k & 0xFFFFF}.wav")
        assert int(store.host_len[slot]) == min(rirs[path].shape[0], 16000)
        assert int(res_a._pair_slots[np.searchsorted(res_a._pair_keys, k)]) == slot
    assert np.all(np.diff(res_a._pair_keys) > 0)
    for (ag1, sg1, paths), (ag2, sg2, _) in zip(a, b):
        np.testing.assert_array_equal(ag1, ag2)
        np.testing.assert_array_equal(sg1, sg2)
        for i, pth in enumerate(paths):
            ref = O.compute_audiogoal(clip, rirs[pth], sr)
            assert O.relerr(ag1[i], ref) < 1e-4 and O.relerr(sg1[i], O.compute_spectrogram(ref.astype(np.float32))) < 1e-4
    res_c, c = run(True, 10)                                     # 10 entries for up to 8 new poses per step: the library evicts
    st_c = res_c.engine.store                                    # (least recently used entries that this step does not resolve to)
    assert st_c.misses > 10 and res_c.library_loaded > 10 and len(st_c._slot_of) <= 10
    assert sorted(st_c._slot_of.values()) == sorted(np.flatnonzero(st_c._used).tolist())
    assert sorted(int(v) for v in res_c._pair_slots) == sorted(st_c._slot_of[k_] for k_ in st_c._slot_of if k_[0] == "ix")
    for (ag1, sg1, _), (ag3, sg3, _) in zip(a, c):
        np.testing.assert_array_equal(ag1, ag3)
    res_d, d = run(False, 10)                                    # ... exactly like the store's own eviction path
    for (ag1, _, _), (ag4, _, _) in zip(a, d):
        np.testing.assert_array_equal(ag1, ag4)


@pytest.mark.gpu
def test_eager_calls_read_new_poses_with_the_library_s_reader(wavs, tmp_path):
    """``attach()`` with the stock reader: a pose whose file is not resident goes through ``AudioEngine.rir_file_slot`` ->
    ``RirStore.load_files`` (the library's reader + one scatter launch) instead of scipy + a row upload; reference semantics file by
    file (simulator.py:615-624): float32 files as they are, an int16 file through scipy, an empty / a junk file = the zero RIR."""
    import shutil
    import types
    from oracle import ss_oracle as O
    from ss_amd import sensors, sim_audio
    from ss_amd.renderer import AudioEngine
    NS = types.SimpleNamespace
    sr = 16000
    root = tmp_path / "scene"
    names = {(0, 1, 2): "a", (0, 3, 4): "ragged", (90, 1, 2): "i16", (90, 3, 4): "empty", (180, 1, 2): "junk", (180, 3, 4): "tiny"}
    for (az, r, s_), n in names.items():
        os.makedirs(root / str(az), exist_ok=True)
        shutil.copy(wavs[n], root / str(az) / f"{r}_{s_}.wav")
    clip = O.synth_sources(np.random.default_rng(2), sr, k=1)[0]

    class Sim:
        config = NS(AUDIO=NS(RIR_SAMPLING_RATE=sr, HAS_DISTRACTOR_SOUND=False), USE_RENDERED_OBSERVATIONS=True)
        binaural_rir_dir = str(root)
        _source_sound_dict = {"s.wav": clip}
        _current_sound, _audio_index, _episode_step_count, _duration = "s.wav", 0, 0, 500
        current_source_sound = property(lambda self: clip)
        _audio_length = 1
    sim = Sim()
    eng = AudioEngine(sr, device="cuda:0", rir_slots=16)
    sim_audio.attach(sim, eng, lazy_audiogoal=False)
    ag_s, sg_s = sensors.AudioGoalSensor(sim=sim, config=NS()), sensors.SpectrogramSensor(sim=sim, config=NS())
    for (az, r, s_), n in names.items():
        sim.azimuth_angle, sim._receiver_position_index, sim._source_position_index = az, r, s_
        sg = sg_s.get_observation(observations=None, episode=None)
        ag = ag_s.get_observation(observations=None, episode=None)
        h = wav_rir_reader(wavs[n])
        if h is None or not np.size(h):
            assert not ag.any() and not sg.any(), n
            continue
        ref = O.compute_audiogoal(clip, np.asarray(h, np.float32), sr)
        assert O.relerr(ag, ref) < 1e-4 and O.relerr(sg, O.compute_spectrogram(ref.astype(np.float32))) < 1e-4, n
    assert eng.store.misses == len(names) and len(eng.store._slot_of) == len(names)
    sim.azimuth_angle, sim._receiver_position_index, sim._source_position_index = 0, 1, 2      # a pose seen before: a hit
    sim._spectrogram_cache.clear(); sim._audiogoal_cache.clear()
    sg_s.get_observation(observations=None, episode=None)
    assert eng.store.misses == len(names)


@pytest.mark.gpu
def test_in_call_loader_orders_its_scatter_behind_steps_in_flight(tmp_path):
    """Overlap mode (ss_ctx_set_overlap) + a store of three entries: thirty large steps that read entry A are queued on the lanes,
    then a new pose is loaded by the library's own loader (ss_ctx_load_rir_files through ``AudioEngine.rir_file_slot``), which
    evicts the least recently used entry - A - while those steps are still in flight.  The scatter goes on the caller's stream:
    it has to wait for the lanes (write after read), or the queued steps render the NEW pose's RIR."""
    import torch
    from oracle import ss_oracle as O
    from ss_amd.renderer import AudioEngine
    from ss_amd.sim_audio import wav_rir_reader
    sr, n_units, n_steps = 16000, 512, 30
    rng = np.random.default_rng(21)
    paths = []
    for k in range(4):
        p = str(tmp_path / f"{k}_0.wav")
        wavfile.write(p, sr, np.ascontiguousarray(O.synth_rir(rng, sr, length=16000, n=1)[0].T))
        paths.append(p)
    clip = O.synth_sources(rng, sr, k=1)[0]
    eng = AudioEngine(sr, device="cuda:0", rir_slots=3)
    sid = eng.source_id("s", clip)
    slots = []
    for p in paths[:3]:
        eng.begin_batch()
        slots.append(eng.rir_file_slot(p, wav_rir_reader))
    ctx = eng.context()
    cols = dict(sound=np.full(n_units, sid), t0=np.zeros(n_units, np.int64), rir=np.full(n_units, slots[0]))
    ref = torch.empty((n_units, 65, 26, 2), device="cuda:0")
    eng.observe_columns(cols, spectrogram_out=ref)
    torch.cuda.synchronize()
    ctx.set_overlap(2)
    out = torch.zeros((n_steps, n_units, 65, 26, 2), device="cuda:0")
    torch.cuda.synchronize()
    for k in range(n_steps):
        eng.observe_columns(cols, spectrogram_out=out[k])
    eng.begin_batch()
    new = eng.rir_file_slot(paths[3], wav_rir_reader)               # evicts A (= slots[0]) inside ONE C call, steps still queued
    assert new == slots[0] and eng.store.misses == 4
    ctx.join()
    torch.cuda.synchronize()
    for k in range(n_steps):
        assert torch.equal(out[k], ref), k
    sg = torch.empty((1, 65, 26, 2), device="cuda:0")               # ... and the new pose renders from the rewritten entry
    eng.observe_columns(dict(sound=np.array([sid]), t0=np.zeros(1, np.int64), rir=np.array([new])), spectrogram_out=sg)
    ctx.join()
    torch.cuda.synchronize()
    a = O.compute_audiogoal(clip, wav_rir_reader(paths[3]), sr)
    assert O.relerr(sg[0].cpu().numpy(), O.compute_spectrogram(a.astype(np.float32))) < 1e-4


def test_a_long_file_that_only_scipy_reads_grows_the_rows(tmp_path):
    """Whole-RIR mode (truncate_to=None): a float32 file longer than the rows grows them (WAV_TOO_LONG -> _ensure_cap); an int16
    file - read by the Python reader, scipy's semantics - has to do the same (scripts/gpu_fuzz_plugin.py: it raised instead)."""
    rng = np.random.default_rng(4)
    long16 = str(tmp_path / "long16.wav")
    wavfile.write(long16, 16000, (rng.standard_normal((23000, 2)) * 3000).astype(np.int16))
    short = str(tmp_path / "short.wav")
    wavfile.write(short, 16000, rng.standard_normal((500, 2)).astype(np.float32))
    s1 = RirStore(8, 16000, "cpu", truncate_to=None)
    a, b = s1.load_files([short, long16], [short, long16])
    assert s1.cap >= 23000 and s1.host_len[b] == 23000 and s1.host_len[a] == 500
    ref = wav_rir_reader(long16)
    assert np.array_equal(s1.bank.data[b, :, :23000].numpy(), ref.T) and not s1.bank.data[b, :, 23000:].any()
    assert np.array_equal(s1.bank.data[a, :, :500].numpy(), wav_rir_reader(short).T)


@pytest.mark.gpu
def test_store_writes_go_behind_steps_in_flight_in_overlap_mode():
    """The same hazard through the PYTHON store (`eng.rir_slot(key, loader)`: an upload on the current stream into the least
    recently used entry): once the engine's context runs on overlap lanes, every device write of the engine's store joins the
    lanes first (`RirStore.before_device_write`)."""
    import torch
    from oracle import ss_oracle as O
    from ss_amd.renderer import AudioEngine
    sr, n_units, n_steps = 16000, 512, 30
    rng = np.random.default_rng(22)
    rirs = [np.ascontiguousarray(O.synth_rir(rng, sr, length=16000, n=1)[0].T) for _ in range(4)]
    clip = O.synth_sources(rng, sr, k=1)[0]
    eng = AudioEngine(sr, device="cuda:0", rir_slots=3)
    sid = eng.source_id("s", clip)
    slots = []
    for k in range(3):
        eng.begin_batch()
        slots.append(eng.rir_slot(("pose", k), lambda k=k: rirs[k]))
    ctx = eng.context()
    cols = dict(sound=np.full(n_units, sid), t0=np.zeros(n_units, np.int64), rir=np.full(n_units, slots[0]))
    ref = torch.empty((n_units, 65, 26, 2), device="cuda:0")
    eng.observe_columns(cols, spectrogram_out=ref)
    torch.cuda.synchronize()
    ctx.set_overlap(2)
    out = torch.zeros((n_steps, n_units, 65, 26, 2), device="cuda:0")
    torch.cuda.synchronize()
    for k in range(n_steps):
        eng.observe_columns(cols, spectrogram_out=out[k])
    eng.begin_batch()
    new = eng.rir_slot(("pose", 3), lambda: rirs[3])                # evicts pose 0's entry while the steps are still queued
    assert new == slots[0]
    sg = torch.empty((1, 65, 26, 2), device="cuda:0")
    eng.observe_columns(dict(sound=np.array([sid]), t0=np.zeros(1, np.int64), rir=np.array([new])), spectrogram_out=sg)
    ctx.join()
    torch.cuda.synchronize()
    for k in range(n_steps):
        assert torch.equal(out[k], ref), k
    a = O.compute_audiogoal(clip, rirs[3], sr)
    assert O.relerr(sg[0].cpu().numpy(), O.compute_spectrogram(a.astype(np.float32))) < 1e-4
