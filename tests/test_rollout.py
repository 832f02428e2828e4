"""§8(f) rank 1: ss_amd.rollout (batch_obs + RolloutStorage) against vectors produced by RUNNING the reference's
classes (tests/golden/make_golden_rollout.py), plus the zero-copy slot path and, on the GPU, kernels writing
straight into the rollout."""
import os
import numpy as np
import pytest
import torch

import rollout_scenario as S
from ss_amd import rollout as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rollout_vectors.npz")


def gold(name):
    z = np.load(GOLD)
    pre = name + "/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


@pytest.mark.parametrize("sc", S.SCENARIOS, ids=lambda s: s["name"])
@pytest.mark.parametrize("zero_copy", [False, True])
def test_replay_equals_reference_run(sc, zero_copy):
    want = gold(sc["name"])
    got = S.replay(sc, R.RolloutStorage, R.batch_obs, device=torch.device("cpu"), zero_copy=zero_copy)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)       # copies and identical fp32 expressions


def test_insert_skips_the_copy_for_slot_tensors_and_copies_everything_else():
    sc = S.SCENARIOS[0]
    rs = R.RolloutStorage(sc["num_steps"], sc["num_envs"], S.obs_space(sc["shapes"]), sc["action"], 4, 1)
    slots = rs.next_observation_slots(["spectrogram"])
    assert isinstance(slots, R.DeviceObservations)
    assert slots["spectrogram"].data_ptr() == rs.observations["spectrogram"][1].data_ptr()
    assert slots["spectrogram"].is_contiguous()
    slots["spectrogram"].fill_(3.0)
    other = torch.full((sc["num_envs"], 2, 16), 5.0)
    N = sc["num_envs"]
    z = lambda *s: torch.zeros(*s)
    rs.insert({"spectrogram": slots["spectrogram"], "audiogoal": other}, z(1, N, 4), z(N, 1).long(), z(N, 1), z(N, 1),
              z(N, 1), torch.ones(N, 1))
    assert rs.step == 1
    assert float(rs.observations["spectrogram"][1].min()) == 3.0 and float(rs.observations["audiogoal"][1].min()) == 5.0
    # a tensor of the right shape but other storage is still copied
    rs.insert({"spectrogram": torch.full_like(slots["spectrogram"], 7.0)}, z(1, N, 4), z(N, 1).long(), z(N, 1),
              z(N, 1), z(N, 1), torch.ones(N, 1))
    assert float(rs.observations["spectrogram"][2].max()) == 7.0


def test_batch_obs_passthrough_and_mixed_containers():
    dev_obs = R.DeviceObservations(spectrogram=torch.ones(3, 5, 3, 2, dtype=torch.float64), skipme=torch.zeros(3))
    b = R.batch_obs(dev_obs, device=None, skip_list=["skipme"])
    assert set(b) == {"spectrogram"} and b["spectrogram"].dtype == torch.float32
    obs = [{"a": np.arange(4, dtype=np.int64), "b": [1.0, 2.0]}, {"a": torch.arange(4), "b": [3.0, 4.0]}]
    b = R.batch_obs(obs)
    assert b["a"].dtype == torch.float32 and b["a"].shape == (2, 4) and b["b"].tolist() == [[1.0, 2.0], [3.0, 4.0]]


def test_generator_rejects_more_batches_than_envs():
    sc = S.SCENARIOS[0]
    rs = R.RolloutStorage(2, 2, S.obs_space({"x": (1,)}), sc["action"], 1)
    with pytest.raises(AssertionError):
        next(rs.recurrent_generator(torch.zeros(2, 2, 1), 3))


@pytest.mark.gpu
def test_kernels_write_straight_into_the_rollout():
    from oracle import ss_oracle as O
    from ss_amd.renderer import BatchedAudioRenderer, RirBank
    dev = torch.device("cuda:0")
    sr, N, T = 16000, 6, 3
    rng = np.random.default_rng(0)
    r = BatchedAudioRenderer(sr, device=dev)
    clips = O.synth_sources(rng, sr, k=3)
    for i, c in enumerate(clips):
        r.add_source(str(i), c)
    rirs = [np.ascontiguousarray(h.T) for h in O.synth_rir(rng, sr, n=8)]            # wav layout [L, 2]
    r.set_rir_bank(RirBank.from_arrays(rirs, dev))
    space = S.obs_space({"spectrogram": r.spectrogram_shape, "audiogoal": (2, sr)})
    rs = R.RolloutStorage(T, N, space, S.ActionSpace(), 4, 1, device=dev)
    z = lambda *s, **k: torch.zeros(*s, device=dev, **k)
    for step in range(T):
        snd, rir = rng.integers(0, 3, N), rng.integers(0, 8, N)
        slots = rs.next_observation_slots()
        plan = r.plan_arrays(snd, np.zeros(N, np.int64), rir)
        r.render(plan, spectrogram_out=slots["spectrogram"], audiogoal_out=slots["audiogoal"])
        before = rs.step
        rs.insert(slots, z(1, N, 4), z(N, 1, dtype=torch.long), z(N, 1), z(N, 1), z(N, 1), torch.ones(N, 1, device=dev))
        got_s = rs.observations["spectrogram"][before + 1].cpu().numpy()
        got_a = rs.observations["audiogoal"][before + 1].cpu().numpy()
        # oracle per env
        for e in range(N):
            want_a = O.compute_audiogoal(clips[snd[e]], rirs[rir[e]], sr)
            want_s = O.compute_spectrogram(want_a)
            assert O.relerr(got_a[e], want_a) <= 1e-4 and O.relerr(got_s[e], want_s) <= 1e-4
