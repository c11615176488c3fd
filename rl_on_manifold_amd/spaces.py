"""Minimal MushroomRL-compatible space / MDP-info objects (duck types).

The reference builds `mushroom_rl.utils.spaces.Box` and `mushroom_rl.core.MDPInfo`
(/root/reference/atacom/atacom.py:50-51, circle_base.py:22-25); agents only touch
`.low / .high / .shape`, `.gamma`, `.horizon`, `.observation_space`, `.action_space`, `.copy()`.
If MushroomRL is importable its own classes are used so `isinstance` checks in user code keep working.
"""
import copy

import numpy as np

try:  # pragma: no cover - MushroomRL is not in this image
    from mushroom_rl.utils.spaces import Box            # type: ignore
    from mushroom_rl.core import MDPInfo                # type: ignore
except Exception:  # noqa: BLE001
    class Box:
        def __init__(self, low, high, shape=None):
            if shape is None:
                self._low = np.array(low, dtype=np.float64)
                self._high = np.array(high, dtype=np.float64)
            else:
                self._low = np.full(shape, low, dtype=np.float64)
                self._high = np.full(shape, high, dtype=np.float64)

        @property
        def low(self):
            return self._low

        @property
        def high(self):
            return self._high

        @property
        def shape(self):
            return self._low.shape

    class MDPInfo:
        def __init__(self, observation_space, action_space, gamma, horizon):
            self.observation_space = observation_space
            self.action_space = action_space
            self.gamma = gamma
            self.horizon = horizon

        @property
        def size(self):
            return self.observation_space.shape + self.action_space.shape

        @property
        def shape(self):
            return self.observation_space.shape + self.action_space.shape

        def copy(self):
            return copy.deepcopy(self)
