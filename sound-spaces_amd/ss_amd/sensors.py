"""Habitat task sensors of the audio path — drop-ins for soundspaces/tasks/nav.py:37-105.

Same registry names (``AudioGoalSensor``, ``SpectrogramSensor``), same uuids (``audiogoal`` / ``spectrogram``),
same ``SensorTypes.PATH``, same ``spaces.Box`` (float32, shape (2, sr) / (65, T4, 2) channel-last), same constructor
and ``get_observation`` signatures, same delegation to ``sim.get_current_*_observation``; the arithmetic behind
runs on the MI355X (ss_amd.sim_audio.attach installs it on the simulator)."""
from __future__ import annotations

from typing import Any

import numpy as np

from . import planning
from .habitat_compat import Sensor, SensorTypes, registry, spaces


@registry.register_sensor
class AudioGoalSensor(Sensor):
    def __init__(self, *args: Any, sim, config, **kwargs: Any):
        self._sim = sim
        super().__init__(config=config)

    def _get_uuid(self, *args: Any, **kwargs: Any):
        return "audiogoal"

    def _get_sensor_type(self, *args: Any, **kwargs: Any):
        return SensorTypes.PATH

    def _get_observation_space(self, *args: Any, **kwargs: Any):
        sensor_shape = (2, self._sim.config.AUDIO.RIR_SAMPLING_RATE)
        return spaces.Box(low=np.finfo(np.float32).min, high=np.finfo(np.float32).max, shape=sensor_shape,
                          dtype=np.float32)

    def get_observation(self, *args: Any, observations, episode, **kwargs: Any):
        return self._sim.get_current_audiogoal_observation()


@registry.register_sensor
class SpectrogramSensor(Sensor):
    cls_uuid: str = "spectrogram"

    def __init__(self, *args: Any, sim, config, **kwargs: Any):
        self._sim = sim
        super().__init__(config=config)

    def _get_uuid(self, *args: Any, **kwargs: Any):
        return "spectrogram"

    def _get_sensor_type(self, *args: Any, **kwargs: Any):
        return SensorTypes.PATH

    def _get_observation_space(self, *args: Any, **kwargs: Any):
        # the reference runs compute_spectrogram(np.ones((2, sr))) just to read the shape (nav.py:76-84);
        # the shape is closed-form, so no GPU work (and no GPU) is needed to build the observation space
        shape = planning.spectrogram_shape(self._sim.config.AUDIO.RIR_SAMPLING_RATE)
        return spaces.Box(low=np.finfo(np.float32).min, high=np.finfo(np.float32).max, shape=shape, dtype=np.float32)

    @staticmethod
    def compute_spectrogram(audio_data, pad_mode: str = "reflect"):
        """[2, T] waveform -> [65, ceil((1+T//160)/4), 2] log1p(mean-pooled |STFT|) via the HIP kernel
        (ss_spectrogram_f32).  Accepts a numpy array (returns numpy) or a CUDA tensor (returns a CUDA tensor)."""
        import torch
        from . import ops
        if isinstance(audio_data, torch.Tensor):
            return ops.spectrogram(audio_data.to(torch.float32).contiguous()[None], pad_mode)[0]
        x = torch.from_numpy(np.ascontiguousarray(audio_data, dtype=np.float32))[None].to("cuda")
        return ops.spectrogram(x, pad_mode)[0].cpu().numpy()

    def get_observation(self, *args: Any, observations, episode, **kwargs: Any):
        return self._sim.get_current_spectrogram_observation(self.compute_spectrogram)


# lets HipSimAudio recognise "the spectrogram of this package" and use the fused kernel's output for it
SpectrogramSensor.compute_spectrogram._ss_hip_fused = True
