"""TEST INFRASTRUCTURE: one seeded PPO-rollout scenario, replayed through the reference RolloutStorage/batch_obs
(tests/golden/make_golden_rollout.py -> rollout_vectors.npz) and through ss_amd.rollout (tests/test_rollout.py)."""
import types
import numpy as np, torch


class ActionSpace:                      # the reference branches on the CLASS NAME (rollout_storage.py:49-57)
    pass


class Box:
    def __init__(self, shape):
        self.shape = tuple(shape)


def obs_space(shapes):
    return types.SimpleNamespace(spaces={k: Box(v) for k, v in shapes.items()})


SCENARIOS = [
    dict(name="discrete", seed=11, num_steps=5, num_envs=4, hidden=6, layers=2, action=ActionSpace(),
         shapes={"spectrogram": (5, 3, 2), "audiogoal": (2, 16), "pointgoal": (2,)}, inserts=7, mini_batches=2,
         gamma=0.99, tau=0.95),
    dict(name="continuous", seed=12, num_steps=4, num_envs=6, hidden=3, layers=1, action=Box((3,)),
         shapes={"spectrogram": (9, 5, 2)}, inserts=4, mini_batches=3, gamma=0.9, tau=0.8),
]


def step_inputs(sc, rng, k):
    """per-env observation dicts (what the vector env returns) + the policy outputs of step k"""
    N = sc["num_envs"]
    obs = []
    for e in range(N):
        d = {}
        for name, shp in sc["shapes"].items():
            a = rng.standard_normal(shp).astype(np.float32 if name != "pointgoal" else np.float64)
            d[name] = a if (e + k) % 2 == 0 else torch.from_numpy(a)          # both containers batch_obs accepts
        d["skipped"] = [1.0, 2.0]
        obs.append(d)
    discrete = isinstance(sc["action"], ActionSpace)
    act = (torch.from_numpy(rng.integers(0, 4, (N, 1))) if discrete
           else torch.from_numpy(rng.standard_normal((N,) + sc["action"].shape).astype(np.float32)))
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    return obs, dict(recurrent_hidden_states=f(sc["layers"], N, sc["hidden"]), actions=act,
                     action_log_probs=f(N, 1), value_preds=f(N, 1), rewards=f(N, 1),
                     masks=torch.from_numpy((rng.random((N, 1)) > 0.2).astype(np.float32)))


def replay(sc, storage_cls, batch_obs, device, reference=False, zero_copy=False):
    rng = np.random.default_rng(sc["seed"])
    rs = storage_cls(sc["num_steps"], sc["num_envs"], obs_space(sc["shapes"]), sc["action"], sc["hidden"], sc["layers"])
    if device is not None:
        rs.to(device)
    dev = device
    obs0, _ = step_inputs(sc, rng, -1)
    b0 = batch_obs(obs0, device=dev, skip_list=["skipped"])
    for s in rs.observations:
        rs.observations[s][0].copy_(b0[s])
    res = {}
    for k in range(sc["inserts"]):
        obs, pol = step_inputs(sc, rng, k)
        if zero_copy:
            # the renderer's outputs land in the storage slot itself; insert() must recognise the alias
            slots = rs.next_observation_slots()
            b = batch_obs(obs, device=dev, skip_list=["skipped"])
            for s in slots:
                slots[s].copy_(b[s])
            b = slots
        else:
            b = batch_obs(obs, device=dev, skip_list=["skipped"])
        if k == 0:
            for s in b:
                res[f"batch0/{s}"] = b[s].detach().cpu().numpy().copy()
        pol = {n: (v.to(dev) if dev is not None else v) for n, v in pol.items()}
        rs.insert(b, pol["recurrent_hidden_states"], pol["actions"], pol["action_log_probs"], pol["value_preds"],
                  pol["rewards"], pol["masks"])
        if (k + 1) % sc["num_steps"] == 0 and k + 1 < sc["inserts"]:
            rs.after_update()
    nv = torch.from_numpy(rng.standard_normal((sc["num_envs"], 1)).astype(np.float32))
    nv = nv.to(dev) if dev is not None else nv
    for use_gae in (True, False):
        rs.compute_returns(nv, use_gae, sc["gamma"], sc["tau"])
        res[f"returns_gae{int(use_gae)}"] = rs.returns.detach().cpu().numpy().copy()
    for n in ("recurrent_hidden_states", "rewards", "value_preds", "action_log_probs", "actions", "prev_actions", "masks"):
        res[n] = getattr(rs, n).detach().cpu().numpy().copy()
    for s in rs.observations:
        res[f"obs/{s}"] = rs.observations[s].detach().cpu().numpy().copy()
    res["step"] = np.asarray(rs.step)
    adv = rs.returns[:-1] - rs.value_preds[:-1]
    torch.manual_seed(sc["seed"])
    names = ["hidden", "actions", "prev_actions", "value_preds", "returns", "masks", "old_log_probs", "adv"]
    for bi, batch in enumerate(rs.recurrent_generator(adv, sc["mini_batches"])):
        for s, v in batch[0].items():
            res[f"gen{bi}/obs/{s}"] = v.detach().cpu().numpy().copy()
        for n, v in zip(names, batch[1:]):
            res[f"gen{bi}/{n}"] = v.detach().cpu().numpy().copy()
    rs.after_update()
    for s in rs.observations:
        res[f"after/obs0/{s}"] = rs.observations[s][0].detach().cpu().numpy().copy()
    res["after/masks0"] = rs.masks[0].detach().cpu().numpy().copy()
    return res
