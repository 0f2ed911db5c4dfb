"""MI355X-native `hanabi_agents.rlax_dqn`: same public names as the reference package
(hanabi_agents/rlax_dqn/__init__.py:2-3) plus the replay / network building blocks."""
from . import bitpack
from .experience_buffer import ExperienceBuffer
from .noisy_mlp import NoisyLinear, NoisyMLP
from .params import RlaxRainbowParams
from .rlax_rainbow import DQNAgent, DQNLearning, DQNPolicy
from .specs import ActionSpec, ObservationSpec
from .transition import Transition

__all__ = ["DQNAgent", "RlaxRainbowParams", "DQNPolicy", "DQNLearning", "ExperienceBuffer", "PriorityBuffer",
           "NoisyLinear", "NoisyMLP", "Transition", "ObservationSpec", "ActionSpec", "smoke_agent_step"]


def __getattr__(name):
    if name == "PriorityBuffer":  # imports hanabi_hip lazily (HIP-only component)
        from .priority_buffer import PriorityBuffer

        return PriorityBuffer
    raise AttributeError(name)


def smoke_agent_step():
    """One explore -> add_experience -> update round of a small agent on cuda:0 (used by __graft_entry__.smoke)."""
    import numpy as np
    import torch

    n, obs_len, n_act = 64, 658, 20
    params = RlaxRainbowParams(train_batch_size=32, experience_buffer_size=1024)
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda")
    rng = np.random.default_rng(0)
    obs = rng.integers(0, 2, (n, obs_len)).astype(np.int8)
    legal = np.ones((n, n_act), np.int8)
    agent.add_experience_first((None, (obs, legal)), np.zeros(n, np.int64))
    actions = agent.explore((None, (obs, legal)))
    assert actions.shape == (n,) and ((actions >= 0) & (actions < n_act)).all()
    obs2 = rng.integers(0, 2, (n, obs_len)).astype(np.int8)
    agent.add_experience((None, (obs2, legal)), actions, rng.random(n), np.ones(n, np.int64))
    agent.update()
    assert torch.isfinite(agent.last_loss).item()
