"""Minimal stand-ins for the dm_env specs the reference agent is constructed with.

`DQNAgent.__init__` only reads `observation_spec.shape` -> (n_games, obs_len) and
`action_spec.num_values` (hanabi_agents/rlax_dqn/rlax_rainbow.py:249-274); any object with those
attributes (e.g. dm_env.specs.Array / DiscreteArray) works. These two are provided because dm_env is
not installed in this image.
"""
import numpy as np


class ObservationSpec:
    def __init__(self, shape, dtype=np.int8):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    def generate_value(self):
        return np.zeros(self.shape, self.dtype)


class ActionSpec:
    def __init__(self, num_values):
        self.num_values = int(num_values)
        self.shape = ()
        self.dtype = np.dtype(np.int32)
