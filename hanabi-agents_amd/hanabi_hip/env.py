"""HanabiEnv — N lock-stepped Hanabi games resident on one MI355X.

Host-side mirror of the parallel environment the reference's agents are driven with (the external
`hanabi_learning_environment` parallel env): `reset()`, `step(actions)` and
batched `(observation [N, obs_len], legal_moves [N, n_actions])` int8 tensors, which is exactly
what `DQNAgent.explore/add_experience` consume as `observations[1]`
(hanabi_agents/rlax_dqn/rlax_rainbow.py:284-308). All buffers are torch CUDA tensors owned by
this object and rewritten in place by every step (zero host traffic).

`packed=True`: the encoder's native output — `obs_bits [N, obs_words] int32`, observation element i
= bit i & 31 of word i >> 5 — is what every step writes (84 instead of 658 bytes per game for
2-player full Hanabi); `obs` is then expanded from it on access (`hb_obs_unpack`), bit for bit what
the unpacked mode writes. The DQN agent, the replay ring and the actor / learner kernels consume
the packed rows directly (`RlaxRainbowParams(packed_obs=True)`).
"""
import ctypes as C

import torch

from . import _capi as K


class HanabiEnv:
    def __init__(self, game="Hanabi-Full", players=2, n_games=1, seed=1234, first_game_id=0, auto_reset=True,
                 lockstep=True, lenient_reward=False, device=None, config=None, decks=None, start_player=0,
                 games_per_wave=None, packed=False):
        if not torch.cuda.is_available():
            raise K.HbError("HanabiEnv needs an MI355X: torch.cuda.is_available() is False and there is no CPU path")
        self.L = K.lib()
        flags = ((K.FLAG_AUTO_RESET if auto_reset else 0) | (K.FLAG_RESET_START_NEXT if lockstep else 0) |
                 (K.FLAG_LENIENT_REWARD if lenient_reward else 0))
        self.cfg = config if config is not None else K.make_config(game, players, flags)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.n = int(n_games)
        self.players = self.cfg.players
        self.num_actions = self.L.hb_num_actions(C.byref(self.cfg))
        self.obs_len = self.L.hb_obs_len(C.byref(self.cfg))
        self.deck_size = self.L.hb_deck_size(C.byref(self.cfg))
        self.state_words = self.L.hb_state_words(C.byref(self.cfg))
        self.obs_words = self.L.hb_obs_words(C.byref(self.cfg))
        self.packed = bool(packed)
        self.first_game_id = int(first_game_id)
        self.seed = int(seed)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            K.check(self.L.hb_env_create(C.byref(self.cfg), self.n, seed, first_game_id, C.byref(h)))
        self.h = h
        dev = self.device
        self._obs = torch.zeros((self.n, self.obs_len), dtype=torch.int8, device=dev)
        self.obs_bits = torch.zeros((self.n, self.obs_words), dtype=torch.int32, device=dev) if self.packed else None
        self._obs_stale = False    # packed mode: `_obs` lags `obs_bits` until someone asks for it
        self.legal = torch.zeros((self.n, self.num_actions), dtype=torch.int8, device=dev)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.terminal = torch.zeros(self.n, dtype=torch.int8, device=dev)
        self.agent_reward = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.agent_step_type = torch.zeros(self.n, dtype=torch.int8, device=dev)
        self.score = torch.zeros(self.n, dtype=torch.int8, device=dev)
        self._decks = None
        self._sel_call = None   # step_select: cached argument addresses
        if games_per_wave is not None:
            self.set_games_per_wave(games_per_wave)
        if decks is not None:
            self.set_decks(decks)
        self.reset(start_player=start_player)

    @property
    def obs(self):
        """[N, obs_len] int8 0/1, the reference's layout. Packed mode: expanded from `obs_bits` when first read after a step."""
        if self._obs_stale:
            K.check(self.L.hb_obs_unpack(K.dptr(self.obs_bits), K.dptr(self._obs), self.n, self.obs_len, K.current_stream()))
            self._obs_stale = False
        return self._obs

    @property
    def net_obs(self):
        """What the agents are fed: the packed rows in packed mode, else `obs`."""
        return self.obs_bits if self.packed else self._obs

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.L.hb_env_destroy(h)
            self.h = None

    # -- configuration -----------------------------------------------------------------------
    def set_games_per_wave(self, g):
        K.check(self.L.hb_env_set_games_per_wave(self.h, int(g)))

    def set_profile_events(self, start=None, stop=None):
        """torch.cuda.Event(enable_timing=True) pair that every following env launch records its own
        begin/end into (dispatch timestamps, like rocprofv3's kernel trace); None, None disables."""
        if start is None:
            K.check(self.L.hb_env_set_profile_events(self.h, None, None))
            return
        if not (start.cuda_event and stop.cuda_event):
            raise K.HbError("record() the events once before passing them (torch creates the HIP event lazily)")
        K.check(self.L.hb_env_set_profile_events(self.h, C.c_void_p(start.cuda_event), C.c_void_p(stop.cuda_event)))

    def set_decks(self, decks):
        """Explicit decks [N, deck_size] uint8 (first card dealt first), or None for Philox shuffles."""
        if decks is None:
            self._decks = None
            K.check(self.L.hb_env_set_decks(self.h, None))
            return
        d = torch.as_tensor(decks, dtype=torch.uint8).reshape(self.n, self.deck_size).contiguous().to(self.device)
        self._decks = d  # keep the borrowed buffer alive
        K.check(self.L.hb_env_set_decks(self.h, K.dptr(d)))

    # -- stepping ----------------------------------------------------------------------------
    def reset(self, mask=None, start_player=0):
        """(Re)deal all games, or those with mask != 0, and refresh obs/legal for the seat to act."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
            assert m.shape == (self.n,)
        K.check(self.L.hb_env_reset(self.h, K.dptr(m), int(start_player), K.current_stream()))
        return self.observe()

    def observe(self):
        if self.packed:
            K.check(self.L.hb_env_observe_packed(self.h, K.dptr(self.obs_bits), None, K.dptr(self.legal), K.dptr(self.agent_reward),
                                                 K.dptr(self.agent_step_type), K.current_stream()))
            self._obs_stale = True
            return self.obs_bits, self.legal
        K.check(self.L.hb_env_observe(self.h, K.dptr(self._obs), K.dptr(self.legal), K.dptr(self.agent_reward),
                                      K.dptr(self.agent_step_type), K.current_stream()))
        return self._obs, self.legal

    def step(self, actions):
        """actions: int32 CUDA tensor [N] of move uids. Returns (obs, legal, reward, terminal) views."""
        a = actions
        if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == torch.int32 and a.is_contiguous()):
            a = torch.as_tensor(a).to(device=self.device, dtype=torch.int32).contiguous()
        assert a.shape == (self.n,)
        if self.packed:
            K.check(self.L.hb_env_step_packed(self.h, K.dptr(a), K.dptr(self.obs_bits), None, K.dptr(self.legal), K.dptr(self.reward),
                                              K.dptr(self.terminal), K.dptr(self.agent_reward), K.dptr(self.agent_step_type),
                                              K.dptr(self.score), K.current_stream()))
            self._obs_stale = True
            return self.obs_bits, self.legal, self.reward, self.terminal
        K.check(self.L.hb_env_step(self.h, K.dptr(a), K.dptr(self._obs), K.dptr(self.legal), K.dptr(self.reward),
                                   K.dptr(self.terminal), K.dptr(self.agent_reward), K.dptr(self.agent_step_type),
                                   K.dptr(self.score), K.current_stream()))
        return self._obs, self.legal, self.reward, self.terminal

    def step_select(self, q, epsilon, seed, draw, first_game_id=0, actions_out=None):
        """step() with the acting agent's epsilon-greedy selection fused into the env kernel (hb_env_step_select_packed):
        game g plays the move hb_policy_select would pick from q[g] (fp32 [N, A]) and this env's current legal mask, with the
        same Philox draws. Returns (actions, obs_bits, legal, reward, terminal); packed envs only."""
        if not self.packed:
            raise ValueError("step_select needs a packed env (HanabiEnv(packed=True))")
        if actions_out is None:
            actions_out = torch.empty(self.n, dtype=torch.int32, device=self.device)
        c = self._sel_call
        if c is None or c[0] != (q.data_ptr(), actions_out.data_ptr(), q.numel(), actions_out.numel()):
            assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.shape == (self.n, self.num_actions)
            assert actions_out.dtype == torch.int32 and actions_out.is_contiguous() and actions_out.shape == (self.n,)
            # (the env's own output buffers never move: their addresses are converted once)
            c = self._sel_call = ((q.data_ptr(), actions_out.data_ptr(), q.numel(), actions_out.numel()), self.legal.data_ptr(),
                                  (self.obs_bits.data_ptr(), None, self.legal.data_ptr(), self.reward.data_ptr(),
                                   self.terminal.data_ptr(), self.agent_reward.data_ptr(), self.agent_step_type.data_ptr(),
                                   self.score.data_ptr()))
        K.check(self.L.hb_env_step_select_packed(self.h, c[0][0], c[1], float(epsilon), int(seed), int(draw), int(first_game_id),
                                                 c[0][1], *c[2], K.current_stream()))
        self._obs_stale = True
        return actions_out, self.obs_bits, self.legal, self.reward, self.terminal

    def random_legal_actions(self, seed, draw, out=None):
        """Uniform-random legal move per game (bench / tests policy)."""
        if out is None:
            out = torch.empty(self.n, dtype=torch.int32, device=self.device)
        K.check(self.L.hb_random_legal_actions(K.dptr(self.legal), self.n, self.num_actions, seed, draw,
                                               self.first_game_id, K.dptr(out), K.current_stream()))
        return out

    # -- introspection -------------------------------------------------------------------------
    def illegal_count(self):
        v = C.c_int64()
        K.check(self.L.hb_env_illegal_count(self.h, C.byref(v)))
        return v.value

    def stats(self):
        """(episodes finished, sum of their final scores) since creation — accumulated inside the step kernel."""
        ep, sc = C.c_int64(), C.c_int64()
        K.check(self.L.hb_env_stats(self.h, C.byref(ep), C.byref(sc)))
        return ep.value, sc.value

    def export_state(self):
        rows = torch.empty((self.n, self.state_words), dtype=torch.int32, device=self.device)
        K.check(self.L.hb_env_export_state(self.h, K.dptr(rows), K.current_stream()))
        return rows

    def import_state(self, rows):
        r = torch.as_tensor(rows).to(device=self.device, dtype=torch.int32).contiguous()
        assert r.shape == (self.n, self.state_words)
        K.check(self.L.hb_env_import_state(self.h, K.dptr(r), K.current_stream()))

    def current_player(self):
        """Seat to act in every game (equal across games in lock-step mode)."""
        return (self.export_state()[:, 0] >> 13) & 7
