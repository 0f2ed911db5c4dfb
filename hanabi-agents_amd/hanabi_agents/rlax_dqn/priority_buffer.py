"""Prioritized replay on the GPU-resident sum tree.

Mirror of the live part of the reference class (hanabi_agents/rlax_dqn/priority_buffer.py:13-52):
  * `add_transitions` first sets the leaves about to be overwritten to `max_priority`
    (priority_buffer.py:29-32) — here one `hb_tree_fill_range` launch on a device scalar instead
    of two Python lists of N elements copied through pybind;
  * `sample_batch(B)` draws stratified keys linspace(1/B, 1, B) - U[0, 1/B), descends the tree
    and returns (indices, probabilities (leaf + 1e-10) / total, Transition) (priority_buffer.py:36-46);
  * `update_priorities(indices, |td|)` stores (p + 1e-10) ** alpha and tracks max / min priority
    (priority_buffer.py:48-52; alpha = 0.6, max_priority starts at alpha).
The tree, the running max/min and all indices stay on the device (`*_dev` methods); the
reference-shaped methods return host values.
"""
import numpy as np
import torch

from .experience_buffer import ExperienceBuffer, _dev


class PriorityBuffer(ExperienceBuffer):
    def __init__(self, observation_len: int, action_len: int, reward_len: int, capacity: int, alpha: float = 0.6,
                 device=None, seed=0, packed=False):
        super().__init__(observation_len, action_len, reward_len, capacity, device=device, seed=seed, packed=packed)
        from hanabi_hip import SumTree  # HIP-only: raises without a GPU / the compiled library

        self.sum_tree = SumTree(capacity, device=self.device)
        self.alpha = alpha
        self._max_priority = torch.full((1,), alpha, dtype=torch.float32, device=self.device)
        self._min_priority = torch.full((1,), alpha, dtype=torch.float32, device=self.device)

    # reference attributes (host copies on demand)
    @property
    def max_priority(self):
        return float(self._max_priority.cpu()[0])

    @property
    def min_priority(self):
        return float(self._min_priority.cpu()[0])

    def add_transitions(self, observation_tm1, action_tm1, reward_t, observation_t, legal_moves_t, terminal_t):
        batch_size = len(observation_tm1)
        if batch_size:
            self.sum_tree.fill_range_dev(self.oldest_entry, batch_size, self._max_priority)
        super().add_transitions(observation_tm1, action_tm1, reward_t, observation_t, legal_moves_t, terminal_t)

    def state_dict(self, include_data=True):
        sd = super().state_dict(include_data)
        sd["max_priority"], sd["min_priority"] = self._max_priority.cpu(), self._min_priority.cpu()
        if include_data:
            sd["tree_nodes"] = self.sum_tree.nodes().cpu()  # every node: inner sums keep their exact fp32 values
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        if sd["has_data"]:
            self.sum_tree.import_nodes(sd["tree_nodes"])
            self._max_priority.copy_(sd["max_priority"])
            self._min_priority.copy_(sd["min_priority"])
        else:
            self.sum_tree.import_nodes(torch.zeros(2 * self.sum_tree.capacity))
            self._max_priority.fill_(self.alpha)
            self._min_priority.fill_(self.alpha)

    # ---- device path ------------------------------------------------------------------------------
    def sample_batch_dev(self, batch_size, uniforms=None):
        """(indices int64 [B], probabilities float64 [B], Transition of device tensors).
        uniforms: optional float64 [B] in [0, 1/B) (parity tests); otherwise drawn on the device."""
        if uniforms is None:
            # default device generator: its Philox offset is graph-safe, so the draw can live inside a captured update
            uniforms = torch.rand(batch_size, dtype=torch.float64, device=self.device) / batch_size
        else:
            uniforms = _dev(uniforms, self.device, torch.float64)
        indices, prios = self.sum_tree.per_sample_dev(uniforms)
        return indices, prios, self.gather_dev(indices)

    def update_priorities_dev(self, indices: torch.Tensor, priorities: torch.Tensor):
        self.sum_tree.per_update_dev(indices, priorities.to(torch.float32).contiguous(), self.alpha,
                                     self._max_priority, self._min_priority)

    # ---- reference-shaped path ----------------------------------------------------------------------
    def sample_batch(self, batch_size, uniforms=None):
        indices, prios, _ = self.sample_batch_dev(batch_size, uniforms)
        idx = indices.cpu().numpy()
        return [int(i) for i in idx], prios.cpu().numpy(), self[idx]

    def update_priorities(self, indices, priorities):
        idx = _dev(np.asarray(indices, dtype=np.int64), self.device, torch.int64)
        self.update_priorities_dev(idx, _dev(priorities, self.device, torch.float32))
