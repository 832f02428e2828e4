"""Device-resident observation batching and rollout storage (the step right after the
audio path).

Mirrors, with the same names / signatures / results:
  * `to_tensor`, `batch_obs`            <- ss_baselines/common/utils.py:117-153
  * `RolloutStorage`                    <- ss_baselines/common/rollout_storage.py:14-243
    (constructor, `to`, `insert`, `after_update`, `compute_returns`, `recurrent_generator`, `_flatten_helper`)
The reference builds a batch with a Python loop of `torch.from_numpy(..).float()` per env and sensor, `torch.stack`,
`.to(device)`, then `insert()` copies it into `observations[sensor][step + 1]`.  Here

  * the storage is allocated on its device from the start (`device=`), and `next_observation_slots()` hands out the
    very rows `observations[sensor][step + 1]` that the next `insert()` would fill, so the HIP kernels
    (`BatchedAudioRenderer.render(..., spectrogram_out=slot)`, `VectorAudioObserver`) write the spectrogram /
    audiogoal of all envs straight into the rollout: no per-env tensors, no stack, no H2D copy, no copy in insert()
    (`insert` recognises a tensor that already IS the slot and skips it);
  * `batch_obs` stacks once per sensor on the host (one H2D per sensor instead of per env) and leaves sensors that are
    already batched device tensors (`DeviceObservations`) alone;
  * `recurrent_generator` gathers the envs of a mini-batch with one `index_select` per tensor instead of a Python loop
    over envs.
Numerics are identical to the reference (tests/test_rollout.py replays tests/golden/rollout_vectors.npz, produced by
running the reference classes)."""
from collections import defaultdict
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch


def to_tensor(v):
    """ss_baselines/common/utils.py:117-123"""
    if torch.is_tensor(v):
        return v
    if isinstance(v, np.ndarray):
        return torch.from_numpy(v)
    return torch.tensor(v, dtype=torch.float)


class DeviceObservations(dict):
    """sensor uuid -> tensor already batched over envs ([num_envs, ...]) and resident on the training device, as
    produced by `VectorAudioObserver` / the renderer.  `batch_obs` passes these through untouched."""


def batch_obs(observations, device: Optional[torch.device] = None, skip_list: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """List of per-env observation dicts -> dict of [num_envs, ...] float32 tensors on `device`
    (ss_baselines/common/utils.py:126-153).  A `DeviceObservations` (or a list whose sensors are all of them
    pre-batched) is already in that form; only dtype / device are normalised."""
    if isinstance(observations, DeviceObservations):
        return {k: v.to(device=device, dtype=torch.float) for k, v in observations.items() if k not in skip_list}
    per_sensor: Dict[str, List] = defaultdict(list)
    for obs in observations:
        for sensor, v in obs.items():
            if sensor in skip_list:
                continue
            per_sensor[sensor].append(v)
    batch = {}
    for sensor, vals in per_sensor.items():
        if all(isinstance(v, np.ndarray) for v in vals) and len({(v.shape, v.dtype) for v in vals}) == 1:
            # one host stack + one transfer (the cast happens on the device when there is one)
            t = torch.from_numpy(np.stack(vals, axis=0))
        else:
            t = torch.stack([to_tensor(v).float() for v in vals], dim=0)
        batch[sensor] = t.to(device=device, dtype=torch.float)
    return batch


class RolloutStorage:
    """Rollout buffers of the PPO trainers, `[num_steps (+1), num_envs, ...]` (rollout_storage.py:14-62)."""

    def __init__(self, num_steps, num_envs, observation_space, action_space, recurrent_hidden_state_size,
                 num_recurrent_layers=1, device=None):
        z = lambda *shape, **kw: torch.zeros(*shape, device=device, **kw)
        self._row_views = {}
        self.observations = {
            sensor: z(num_steps + 1, num_envs, *space.shape) for sensor, space in observation_space.spaces.items()
        }
        self.recurrent_hidden_states = z(num_steps + 1, num_recurrent_layers, num_envs, recurrent_hidden_state_size)
        self.rewards = z(num_steps, num_envs, 1)
        self.value_preds = z(num_steps + 1, num_envs, 1)
        self.returns = z(num_steps + 1, num_envs, 1)
        self.action_log_probs = z(num_steps, num_envs, 1)
        discrete = action_space.__class__.__name__ == "ActionSpace"          # the reference's test, by class name
        action_shape = 1 if discrete else action_space.shape[0]
        adt = torch.long if discrete else torch.float
        self.actions = z(num_steps, num_envs, action_shape, dtype=adt)
        self.prev_actions = z(num_steps + 1, num_envs, action_shape, dtype=adt)
        self.masks = torch.ones(num_steps + 1, num_envs, 1, device=device)
        self.num_steps = num_steps
        self.step = 0

    _FIELDS = ("recurrent_hidden_states", "rewards", "value_preds", "returns", "action_log_probs", "actions",
               "prev_actions", "masks")

    def to(self, device):
        for sensor in self.observations:
            self.observations[sensor] = self.observations[sensor].to(device)
        for name in self._FIELDS:
            setattr(self, name, getattr(self, name).to(device))

    # ---- zero-copy hand-off from the audio kernels ------------------------------------------------------------
    def observation_slot(self, sensor: str, step: Optional[int] = None) -> torch.Tensor:
        """The contiguous `[num_envs, ...]` row that the next `insert()` fills for `sensor`
        (`observations[sensor][self.step + 1]`, rollout_storage.py:89-92), or row `step` if given."""
        t = self.observations[sensor]
        rows = self._row_views.get(sensor)
        if rows is None or rows[0] is not t:                 # views of the rows, made once per storage tensor (a tensor
            rows = self._row_views[sensor] = (t, list(t.unbind(0)))      # index costs microseconds on the per-step path)
        return rows[1][self.step + 1 if step is None else step]

    def next_observation_slots(self, sensors: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
        return DeviceObservations({s: self.observation_slot(s) for s in (sensors or self.observations)})

    @staticmethod
    def _is_same_storage(a: torch.Tensor, b: torch.Tensor) -> bool:
        return (a.device == b.device and a.dtype == b.dtype and a.shape == b.shape and a.stride() == b.stride()
                and a.data_ptr() == b.data_ptr())

    def insert(self, observations, recurrent_hidden_states, actions, action_log_probs, value_preds, rewards, masks):
        """rollout_storage.py:78-102; sensors whose tensor already is the slot are not copied."""
        nxt = self.step + 1
        for sensor, v in observations.items():
            dst = self.observations[sensor][nxt]
            if not (torch.is_tensor(v) and self._is_same_storage(v, dst)):
                dst.copy_(v)
        self.recurrent_hidden_states[nxt].copy_(recurrent_hidden_states)
        self.actions[self.step].copy_(actions)
        self.prev_actions[nxt].copy_(actions)
        self.action_log_probs[self.step].copy_(action_log_probs)
        self.value_preds[self.step].copy_(value_preds)
        self.rewards[self.step].copy_(rewards)
        self.masks[nxt].copy_(masks)
        self.step = nxt % self.num_steps

    def after_update(self):
        """last row becomes row 0 of the next rollout (rollout_storage.py:104-110)"""
        for t in list(self.observations.values()) + [self.recurrent_hidden_states, self.masks, self.prev_actions]:
            t[0].copy_(t[-1])

    def compute_returns(self, next_value, use_gae, gamma, tau):
        """discounted returns / GAE(gamma, tau), backwards over the steps (rollout_storage.py:112-132)"""
        T = self.rewards.size(0)
        if use_gae:
            self.value_preds[-1] = next_value
            gae = 0
            for t in range(T - 1, -1, -1):
                not_done = self.masks[t + 1]
                delta = self.rewards[t] + gamma * self.value_preds[t + 1] * not_done - self.value_preds[t]
                gae = delta + gamma * tau * not_done * gae
                self.returns[t] = gae + self.value_preds[t]
        else:
            self.returns[-1] = next_value
            for t in range(T - 1, -1, -1):
                self.returns[t] = self.returns[t + 1] * gamma * self.masks[t + 1] + self.rewards[t]

    def recurrent_generator(self, advantages, num_mini_batch):
        """Mini-batches of whole env trajectories, `num_envs // num_mini_batch` envs each, in a random env order
        (rollout_storage.py:134-229): yields (observations, hidden states of step 0, actions, prev_actions,
        value_preds, returns, masks, old log-probs, advantages), time-major tensors flattened to (T*N, ...)."""
        num_envs = self.rewards.size(1)
        assert num_envs >= num_mini_batch, (
            "Trainer requires the number of processes ({}) to be greater than or equal to the number of "
            "trainer mini batches ({}).".format(num_envs, num_mini_batch))
        per_batch = num_envs // num_mini_batch
        perm = torch.randperm(num_envs)                     # host RNG, as in the reference (same seed -> same order)
        T = self.num_steps
        for start in range(0, num_envs, per_batch):
            # the reference indexes perm[start + offset] for offset < per_batch (IndexError on a ragged tail)
            if start + per_batch > num_envs:
                raise IndexError("index {} is out of bounds for dimension 0 with size {}".format(num_envs, num_envs))
            ind = perm[start:start + per_batch].to(self.rewards.device)
            take = lambda x: self._flatten_helper(T, per_batch, x.index_select(1, ind))
            yield (
                {s: take(o[:-1]) for s, o in self.observations.items()},
                self.recurrent_hidden_states[0].index_select(1, ind),
                take(self.actions), take(self.prev_actions[:-1]), take(self.value_preds[:-1]),
                take(self.returns[:-1]), take(self.masks[:-1]), take(self.action_log_probs), take(advantages),
            )

    @staticmethod
    def _flatten_helper(t: int, n: int, tensor: torch.Tensor) -> torch.Tensor:
        """(t, n, ...) -> (t*n, ...) (rollout_storage.py:231-243)"""
        return tensor.reshape(t * n, *tensor.size()[2:])
