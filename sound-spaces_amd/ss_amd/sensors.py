"""Habitat task sensors of the audio path — drop-ins for soundspaces/tasks/nav.py:37-105.

Same registry names (``AudioGoalSensor``, ``SpectrogramSensor``, av_wan's ``Intensity``), same uuids (``audiogoal`` /
``spectrogram`` / ``intensity``),
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


@registry.register_sensor(name="Intensity")
class Intensity(Sensor):
    """av_wan's intensity placeholder sensor (ss_baselines/av_wan/avwan_sensors.py:60-100; selected by
    ``TASK.INTENSITY.TYPE = "Intensity"``, av_wan/config/default.py:191-194): onset = first sample above 0.1 * max over
    both ears (the earlier of the two ears), value = mean square of the 150 samples from the onset.  Same registry name,
    uuid, sensor type and Box as the reference; the reduction runs in ``k_intensity`` (``ss_intensity_f32``)."""

    def __init__(self, sim, config, *args: Any, **kwargs: Any):
        self._sim = sim
        super().__init__(config=config)

    def _get_uuid(self, *args: Any, **kwargs: Any):
        return "intensity"

    def _get_sensor_type(self, *args: Any, **kwargs: Any):
        return SensorTypes.COLOR                                                   # avwan_sensors.py:72 (sic)

    def _get_observation_space(self, *args: Any, **kwargs: Any):
        return spaces.Box(low=0, high=1, shape=(1,), dtype=bool)                   # avwan_sensors.py:75-80 (sic)

    @staticmethod
    def compute_intensity(audiogoal, num_frame: int = 150):
        """[2, T] (numpy -> python float) or [N, 2, T] CUDA tensor (-> CUDA tensor [N])."""
        import torch
        from . import ops
        if isinstance(audiogoal, torch.Tensor):
            return ops.intensity(audiogoal.to(torch.float32).contiguous(), num_frame)
        x = torch.from_numpy(np.ascontiguousarray(audiogoal, dtype=np.float32))[None].to("cuda")
        return float(ops.intensity(x, num_frame)[0])

    def get_observation(self, *args: Any, observations, episode, **kwargs: Any):
        audiogoal = self._sim.get_current_audiogoal_observation()
        return [self.compute_intensity(audiogoal, 150)]                             # avwan_sensors.py:93-100: [rms]


# lets HipSimAudio recognise "the spectrogram of this package" and use the fused kernel's output for it
SpectrogramSensor.compute_spectrogram._ss_hip_fused = True
