"""Binding to the Habitat plugin API (habitat.core.registry / habitat.core.simulator / gym.spaces).

When habitat-lab is installed the real classes are used, so the sensors of ss_amd.sensors register into the same
``registry`` the reference's do (soundspaces/tasks/nav.py:37,63) and are selected by the unchanged YAML
(``TASK.SPECTROGRAM_SENSOR.TYPE: SpectrogramSensor``, ss_baselines/av_nav/config/default.py:100-106).
When it is not installed (this build container, the GPU box) minimal stand-ins with the same surface are used so
the package still imports and the plugin contract can be exercised by the tests."""
from __future__ import annotations

import enum

import numpy as np

try:  # pragma: no cover - exercised only where habitat is installed
    from habitat.core.registry import registry
    from habitat.core.simulator import Sensor, SensorTypes
    from gym import spaces
    HAVE_HABITAT = True
except Exception:  # ImportError and friends
    HAVE_HABITAT = False

    class SensorTypes(enum.Enum):
        NULL = 0
        COLOR = 1
        DEPTH = 2
        PATH = 8

    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    class spaces:  # noqa: N801 - mirrors ``gym.spaces``
        Box = _Box

    class Sensor:
        """habitat.core.simulator.Sensor: uuid / sensor_type / observation_space resolved at construction."""

        def __init__(self, *args, **kwargs):
            self.config = kwargs["config"] if "config" in kwargs else None
            self.uuid = self._get_uuid(*args, **kwargs)
            self.sensor_type = self._get_sensor_type(*args, **kwargs)
            self.observation_space = self._get_observation_space(*args, **kwargs)

        def _get_uuid(self, *args, **kwargs):
            raise NotImplementedError

        def _get_sensor_type(self, *args, **kwargs):
            raise NotImplementedError

        def _get_observation_space(self, *args, **kwargs):
            raise NotImplementedError

        def get_observation(self, *args, **kwargs):
            raise NotImplementedError

    class _Registry:
        def __init__(self):
            self.mapping = {"sensor": {}}

        def register_sensor(self, to_register=None, *, name=None):
            def wrap(cls):
                self.mapping["sensor"][name or cls.__name__] = cls
                return cls
            return wrap(to_register) if to_register is not None else wrap

        def get_sensor(self, name):
            return self.mapping["sensor"].get(name)

    registry = _Registry()
