"""Replay ring, resident in HBM.

Same constructor, attributes and methods as the reference's numpy ring
(hanabi_agents/rlax_dqn/experience_buffer.py:5-97): six preallocated arrays, `oldest_entry` /
`size` / `capacity`, `get_update_indices`, `add_transitions` with wrap-around, `buf[indices]`
-> `Transition`, uniform `sample`. The ring-pointer arithmetic is restated line by line (C-12);
the storage is torch tensors on the agent's device, so inserting the N transitions of one env
step and gathering a batch never cross PCIe (the reference's largest avoidable cost, SURVEY
§3.4). Rewards are kept in float32 on the device (the reference stores float64, C-13) and are
handed back as float64 by the numpy-facing accessors.

Row order: `add_transitions` keeps the order of the rows it is given, exactly like the
reference's slice assignments, so ring contents are comparable entry by entry.
"""
import numpy as np
import torch

from . import bitpack
from .transition import Transition


def _dev(x, device, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(x)).to(device=device, dtype=dtype)


class ExperienceBuffer:
    """ExperienceBuffer stores transitions for training (device-resident)."""

    def __init__(self, observation_len: int, action_len: int, reward_len: int, capacity: int, device=None, seed=0,
                 packed=False):
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        d = self.device
        # packed=True: both observation rings hold bit-packed rows (bitpack.py: ceil(obs_len / 32) int32 words, 84 instead of
        # 658 bytes for 2-player full Hanabi); everything handed OUT (`buf[indices]`, gather_dev) is the reference's int8 form
        self.packed = bool(packed)
        self.observation_len = int(observation_len)
        row, odt = (bitpack.words_for(observation_len), torch.int32) if self.packed else (observation_len, torch.int8)
        self._obs_tm1_buf = torch.zeros((capacity, row), dtype=odt, device=d)
        self._act_tm1_buf = torch.zeros((capacity, 1), dtype=torch.int8, device=d)
        self._obs_t_buf = torch.zeros((capacity, row), dtype=odt, device=d)
        self._lms_t_buf = torch.zeros((capacity, action_len), dtype=torch.int8, device=d)
        self._rew_t_buf = torch.zeros((capacity, reward_len), dtype=torch.float32, device=d)
        self._terminal_t_buf = torch.zeros((capacity, 1), dtype=torch.bool, device=d)
        self.oldest_entry = 0
        self.capacity = capacity
        self.size = 0
        self._gen = torch.Generator(device=d).manual_seed(seed)
        self._size_t = torch.zeros((), dtype=torch.float64, device=d)
        self._size_wp = torch.zeros(2, dtype=torch.int64, device=d)   # {size, write pointer} for the n-step gather
        self.track_wp = False         # set by the agent when n_step > 1
        self._synced = None           # (size, oldest_entry) last published by sync_size()
        self.rows_per_insert = None   # constant batch size of all inserts so far (None: nothing yet, -1: it varied)

    # ---- ring arithmetic (experience_buffer.py:21-24,46-81) ---------------------------------------
    def get_update_indices(self, batch_size):
        if self.oldest_entry + batch_size <= self.capacity:
            return list(range(self.oldest_entry, self.oldest_entry + batch_size))
        return list(range(self.oldest_entry, self.capacity)) + list(
            range(0, batch_size - self.capacity + self.oldest_entry))

    def _advance(self, batch_size):
        """Returns the list of (buffer slice, batch slice) pairs and moves oldest_entry/size."""
        if self.rows_per_insert is None:
            self.rows_per_insert = batch_size
        elif self.rows_per_insert != batch_size:
            self.rows_per_insert = -1
        start = self.oldest_entry
        if start + batch_size <= self.capacity:
            parts = [(slice(start, start + batch_size), slice(0, batch_size))]
            if start + batch_size == self.capacity:
                self.size = self.capacity
            self.oldest_entry = (start + batch_size) % self.capacity
            self.size = max(self.size, self.oldest_entry)
        else:
            tail = start + batch_size - self.capacity
            # the reference writes rows [:batch-tail] at the end and the LAST `tail` rows at the front
            parts = [(slice(start, self.capacity), slice(0, batch_size - tail)),
                     (slice(0, tail), slice(batch_size - tail, batch_size))]
            self.oldest_entry = tail
            self.size = self.capacity
        return parts

    def add_transitions(self, observation_tm1, action_tm1, reward_t, observation_t, legal_moves_t, terminal_t):
        """Append a batch (numpy arrays or torch tensors on any device); shapes as in the reference."""
        d = self.device
        cols = (
            (self._obs_tm1_buf, self.obs_rows(observation_tm1)),
            (self._act_tm1_buf, _dev(action_tm1, d, torch.int8).reshape(-1, 1)),
            (self._rew_t_buf, _dev(reward_t, d, torch.float32).reshape(-1, self._rew_t_buf.shape[1])),
            (self._obs_t_buf, self.obs_rows(observation_t)),
            (self._lms_t_buf, _dev(legal_moves_t, d, torch.int8)),
            (self._terminal_t_buf, _dev(terminal_t, d, torch.bool).reshape(-1, 1)),
        )
        batch_size = cols[0][1].shape[0]
        if batch_size == 0:
            return
        # (like the reference, a batch may exceed the free tail and even the capacity by less than one lap:
        #  tests/rlax_dqn/test_experience_buffer.py:99-142 adds 8 rows to a ring of 7)
        for dst, src in self._advance(batch_size):
            for buf, val in cols:
                buf[dst] = val[src]

    def obs_rows(self, obs):
        """Observations in this ring's storage form: int8 [n, obs_len], or packed int32 [n, words]. Accepts either form."""
        if isinstance(obs, torch.Tensor) and bitpack.is_packed(obs, self.observation_len) and self.observation_len != obs.shape[1]:
            obs = obs.to(self.device)
            return obs if self.packed else bitpack.unpack(obs, self.observation_len)
        obs = _dev(obs, self.device, torch.int8)
        return bitpack.pack(obs) if self.packed else obs

    def _obs_out(self, rows):
        return bitpack.unpack(rows, self.observation_len) if self.packed else rows

    # ---- checkpointing (the reference saves network parameters only) ---------------------
    _DATA = ("_obs_tm1_buf", "_act_tm1_buf", "_obs_t_buf", "_lms_t_buf", "_rew_t_buf", "_terminal_t_buf")

    def state_dict(self, include_data=True):
        """Ring pointers, sampler RNG and (optionally) the `size` rows written so far, as host tensors."""
        sd = dict(capacity=self.capacity, oldest_entry=self.oldest_entry, size=self.size, packed=self.packed,
                  rows_per_insert=self.rows_per_insert, gen=self._gen.get_state().cpu(), has_data=bool(include_data))
        if include_data:
            sd["data"] = {name: getattr(self, name)[:self.size].cpu() for name in self._DATA}
        return sd

    def load_state_dict(self, sd):
        """In place (captured graphs keep pointing at the same buffers)."""
        if sd["capacity"] != self.capacity:
            raise ValueError(f"checkpoint ring capacity {sd['capacity']} != {self.capacity}")
        if bool(sd.get("packed", False)) != self.packed:
            raise ValueError("checkpoint ring and this ring differ in observation storage (packed_obs)")
        if sd["has_data"]:
            for name in self._DATA:
                rows = sd["data"][name]
                getattr(self, name)[:rows.shape[0]].copy_(rows)
            self.oldest_entry, self.size = int(sd["oldest_entry"]), int(sd["size"])
        else:  # weights-only resume: start with an empty ring
            self.oldest_entry, self.size = 0, 0
        self.rows_per_insert = sd["rows_per_insert"] if sd["has_data"] else None
        self._gen.set_state(sd["gen"].cpu())
        self.sync_size()

    # ---- access ---------------------------------------------------------------------------------
    def gather_dev(self, indices: torch.Tensor) -> Transition:
        """Batch as device tensors (the learner's path)."""
        return Transition(
            self._obs_out(self._obs_tm1_buf.index_select(0, indices)), self._act_tm1_buf.index_select(0, indices),
            self._rew_t_buf.index_select(0, indices), self._obs_out(self._obs_t_buf.index_select(0, indices)),
            self._lms_t_buf.index_select(0, indices), self._terminal_t_buf.index_select(0, indices))

    def __getitem__(self, indices) -> Transition:
        """Numpy view for reference-shaped callers: int8 / float64 / bool arrays (experience_buffer.py:83-87)."""
        idx = _dev(np.asarray(indices, dtype=np.int64).reshape(-1), self.device, torch.int64)
        t = self.gather_dev(idx)
        return Transition(t.observation_tm1.cpu().numpy(), t.action_tm1.cpu().numpy(),
                          t.reward_t.cpu().numpy().astype(np.float64), t.observation_t.cpu().numpy(),
                          t.legal_moves_t.cpu().numpy(), t.terminal_t.cpu().numpy())

    def sample_indices_dev(self, batch_size: int) -> torch.Tensor:
        """Uniform WITH replacement over [0, size) (experience_buffer.py:96; C-11), own generator."""
        if self.size == 0:
            raise ValueError("cannot sample from an empty buffer")
        if self.device.type == "cuda":
            # graph-safe form: default generator + a device-side copy of `size` (refreshed by sync_size() outside
            # any captured region), so a captured update keeps sampling from the ring as it fills
            u = torch.rand(batch_size, device=self.device, dtype=torch.float64)
            return torch.minimum((u * self._size_t).long(), (self._size_t - 1).long())
        return torch.randint(0, self.size, (batch_size,), device=self.device, generator=self._gen)

    def sync_size(self):
        """Publish the host-side `size` / write pointer to the device scalars read inside captured graphs."""
        if self._synced == (self.size, self.oldest_entry):
            return  # nothing moved since the last call (a full ring between inserts): no launches
        if self._synced is None or self._synced[0] != self.size:
            self._size_t.fill_(float(self.size))
        if self.track_wp:  # only the n-step gather reads these (fill kernels, no host->device copies)
            self._size_wp[0:1].fill_(self.size)
            self._size_wp[1:2].fill_(self.oldest_entry)
        self._synced = (self.size, self.oldest_entry)

    def gather_nstep_dev(self, indices: torch.Tensor, n_step: int, gamma: float):
        """n-step transitions assembled at sample time (PyTorch reference of hb_replay_gather's chain walk).

        Valid when every insert appended `rows_per_insert` rows (lock-step self-play): the same seat's next
        transition of the same game sits rows_per_insert slots further on. Follows up to n_step-1 successors,
        stopping at an episode end or at the write pointer. Returns (Transition, discount [B] = gamma^m)."""
        if n_step > 1 and (self.rows_per_insert is None or self.rows_per_insert < 1):
            raise ValueError("n_step > 1 needs inserts of a constant row count")
        cap, n_ins = self.capacity, self.rows_per_insert or 1
        j = indices.clone()
        rew = self._rew_t_buf[j, 0].clone()
        disc = torch.full_like(rew, gamma)
        alive = torch.ones_like(j, dtype=torch.bool)
        # Ring bounds come from the DEVICE scalars {size, write pointer}, like hb_replay_gather: inside a captured
        # update graph host ints would be frozen at capture time and later replays would walk chains past the real
        # write pointer. sync_size() (called by the agent before every update, outside the graph) refreshes them.
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if not capturing:
            if not self.track_wp:
                self.track_wp, self._synced = True, None
            self.sync_size()
        elif not self.track_wp:
            raise RuntimeError("gather_nstep_dev inside a captured graph needs track_wp (set before the capture)")
        size, wp = self._size_wp[0], self._size_wp[1]
        ahead = torch.where(size >= cap, (wp - 1 - indices) % cap, size - 1 - indices)
        for m in range(1, n_step):
            alive = alive & ~self._terminal_t_buf[j, 0] & (m * n_ins <= ahead)
            nxt = (j + n_ins) % cap
            j = torch.where(alive, nxt, j)
            rew = torch.where(alive, rew + disc * self._rew_t_buf[j, 0], rew)
            disc = torch.where(alive, disc * gamma, disc)
        t = Transition(self._obs_out(self._obs_tm1_buf.index_select(0, indices)), self._act_tm1_buf.index_select(0, indices),
                       rew[:, None], self._obs_out(self._obs_t_buf.index_select(0, j)), self._lms_t_buf.index_select(0, j),
                       self._terminal_t_buf.index_select(0, j))
        return t, disc

    def sample_dev(self, batch_size: int) -> Transition:
        return self.gather_dev(self.sample_indices_dev(batch_size))

    def sample(self, batch_size: int) -> Transition:
        return self[self.sample_indices_dev(batch_size).cpu().numpy()]
