"""DQNAgent — the drop-in agent API of the reference, MI355X-native underneath.

Method names, argument meaning and bookkeeping follow `hanabi_agents.rlax_dqn.DQNAgent`
(hanabi_agents/rlax_dqn/rlax_rainbow.py:220-365):

    DQNAgent(observation_spec, action_spec, params=RlaxRainbowParams())
    explore(observations) / exploit(observations)       -> int actions [N]
    add_experience_first(observations, step_types)      FIRST rows seed last_obs
    add_experience(observations, actions, rewards, step_types)
    update()                                            one gradient step
    save_weights(path, fname_part) / restore_weights(online_file, trg_file)
    requires_vectorized_observation()                   -> True

`observations` is the session's tuple whose element [1] is (obs [N, obs_len], legal [N, A])
(rlax_rainbow.py:278,285,293,298). numpy inputs behave exactly like the reference (copied to the
device on entry, numpy actions returned). torch CUDA tensors are taken zero-copy and actions come
back as a CUDA int32 tensor — the path the self-play driver uses so that an env step, the policy,
the replay insert and the learner never leave the GPU.

What runs where: the network GEMMs, softmax/expectation, C51 projection, Adam — PyTorch-ROCm
(rocBLAS/hipBLASLt MFMA GEMMs); prioritized sampling / priority update — the HIP sum tree of
hanabi_hip; replay ring — device tensors. Quirks of the reference that are kept or made
switchable are listed in DESIGN.md §7.
"""
import os
from typing import Tuple

import numpy as np
import torch

from . import bitpack
from . import learning as L
from .experience_buffer import ExperienceBuffer
from .noisy_mlp import NoisyMLP, PlainMLP
from .params import RlaxRainbowParams
from .priority_buffer import PriorityBuffer

_DTYPES = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def _shape_of(spec):
    return tuple(spec.shape)


class DQNPolicy:
    """Greedy and legal-epsilon-greedy action selection (rlax_rainbow.py:30-150)."""

    @staticmethod
    def q_values(network, atoms, obs, legal, distributional=True):
        """q [N, A] with illegal moves at -inf (rlax_rainbow.py:113-119,141-147)."""
        out = network(obs)
        if distributional:
            n_actions, n_atoms = atoms.shape
            q = L.expected_q(out.view(-1, n_actions, n_atoms), atoms)
        else:
            q = out
        return torch.where(legal.bool(), q, torch.full_like(q, float("-inf")))

    @staticmethod
    def sample(q, legal, epsilon, u_explore, u_pick):
        """Legal epsilon-greedy sample (rlax_rainbow.py:34-71).

        The reference mixes (1-eps) * uniform-over-argmax-ties + eps * uniform-over-legal and samples the
        mixture through a cumulative sum with a round-off guard (C-14). The same distribution is drawn
        here without floating-point cumsums: with probability eps pick the k-th legal move, otherwise
        the k-th arg-max tie, k uniform. u_explore, u_pick: uniforms in [0,1) of shape [N].
        """
        legal_b = legal.bool()
        greedy = (q == q.max(dim=-1, keepdim=True).values) & legal_b
        pool = torch.where((u_explore < epsilon)[:, None], legal_b, greedy)
        count = pool.sum(dim=-1)
        k = torch.clamp((u_pick * count).long(), max=torch.clamp(count - 1, min=0))
        hit = (torch.cumsum(pool.int(), dim=-1) == (k + 1)[:, None]) & pool
        return torch.argmax(hit.int(), dim=-1).to(torch.int32)

    @staticmethod
    def policy(network, atoms, epsilon, obs, legal, u_explore, u_pick, distributional=True):
        q = DQNPolicy.q_values(network, atoms, obs, legal, distributional)
        return q, DQNPolicy.sample(q, legal, epsilon, u_explore, u_pick)

    @staticmethod
    def eval_policy(network, atoms, obs, legal, u_pick, distributional=True):
        """rlax.greedy().sample: uniform among arg-max ties (rlax_rainbow.py:125-150)."""
        q = DQNPolicy.q_values(network, atoms, obs, legal, distributional)
        return DQNPolicy.sample(q, legal, 0.0, torch.ones_like(u_pick), u_pick)


class DQNLearning:
    @staticmethod
    def loss(online, target, atoms, transitions, discount, prios, beta_is, mask_terminal=False, distributional=True):
        """mean(td * w_IS) and the per-sample |td| that become the new priorities (rlax_rainbow.py:187-200)."""
        obs_tm1 = transitions.observation_tm1
        obs_t = transitions.observation_t
        a_tm1 = transitions.action_tm1[:, 0].long()
        r_t = transitions.reward_t[:, 0].to(torch.float32)
        term = transitions.terminal_t[:, 0]
        w = L.is_weights(prios, beta_is)
        if distributional:
            a, k = atoms.shape
            logits_tm1 = online(obs_tm1).view(-1, a, k)
            with torch.no_grad():
                logits_t = target(obs_t).view(-1, a, k)
                logits_sel = online(obs_t).view(-1, a, k)
            td = L.categorical_double_q_td(logits_tm1, a_tm1, r_t, discount, atoms, logits_t, logits_sel,
                                           term if mask_terminal else None)
            return torch.mean(td * w), torch.abs(td).detach()
        # scalar double-DQN of the older agent (rlax_dqn.py:170-205): IS-weighted l2 with clipped gradient
        q_tm1 = online(obs_tm1)
        with torch.no_grad():
            q_t = target(obs_t)
            q_sel = online(obs_t)
        td = L.double_q_td(q_tm1, a_tm1, r_t, discount, q_t, q_sel, term)
        return L.clip_gradient(torch.mean(w * 0.5 * td * td)), torch.abs(td).detach()


class DQNAgent:
    def __init__(self, observation_spec, action_spec, params: RlaxRainbowParams = RlaxRainbowParams(), device=None,
                 process_group=None, use_graphs=True, use_fused_learner=True):
        if not callable(params.epsilon):
            eps = params.epsilon
            params = params._replace(epsilon=lambda ts: eps)
        if not callable(params.beta_is):
            beta = params.beta_is
            params = params._replace(beta_is=lambda ts: beta)
        self.params = params
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        # torch.distributed group for the data-parallel gradient all-reduce (None: the default group when one is initialised;
        # False: never — a purely local agent inside a distributed job, e.g. a baseline timed on one rank)
        self.process_group = process_group
        n_games, obs_len = _shape_of(observation_spec)
        self.n_actions = int(action_spec.num_values)
        self.obs_len = obs_len
        cd = _DTYPES[params.compute_dtype]
        self.distributional = bool(params.distributional)

        # Q-network: online and target copies (the reference aliases them until the first update; here the
        # target is a separate module holding equal values, which is observationally identical, C-10)
        def build():
            if self.distributional:
                return NoisyMLP(obs_len, tuple(params.layers) + (self.n_actions * params.n_atoms,), seed=params.seed,
                                compute_dtype=cd)
            return PlainMLP(obs_len, tuple(params.layers) + (self.n_actions,), seed=params.seed, compute_dtype=cd)

        self.online = build().to(self.device)
        self.target = build().to(self.device)
        self.target.load_state_dict(self.online.state_dict())
        for p in self.target.parameters():
            p.requires_grad_(False)
        self.atoms = torch.linspace(-params.atom_vmax, params.atom_vmax, params.n_atoms, device=self.device).repeat(
            self.n_actions, 1)  # [A, K] (rlax_rainbow.py:253-254)
        # optix.adam(lr, eps=3.125e-5) has torch.optim.Adam's form, eps outside the sqrt
        on_gpu = self.device.type == "cuda"
        # GPU paths that do not go through the FusedLearner (vanilla DQN, deeper nets): torch's single-launch fused Adam (same
        # arithmetic as the foreach form, ~10 launches fewer per update: config 2 0.60 -> 0.52 ms per step); HB_ADAM_FUSED=0: foreach
        fused_adam = on_gpu and os.environ.get("HB_ADAM_FUSED", "1") != "0"
        self.optimizer = torch.optim.Adam(self.online.parameters(), lr=params.learning_rate, betas=(0.9, 0.999),
                                          eps=3.125e-5, capturable=on_gpu, foreach=(True if on_gpu and not fused_adam else None),
                                          fused=True if fused_adam else None)
        # one flat fp32 gradient buffer; every parameter's .grad is a view into it (single-bucket all-reduce,
        # no flatten/unflatten copies)
        plist = list(self.online.parameters())
        self._flat_grad = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=self.device)
        off = 0
        for p in plist:
            p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._beta = torch.zeros((), dtype=torch.float32, device=self.device)
        self._beta_host = None
        self.use_graphs = use_graphs
        self._graph1 = self._graph2 = None
        self.train_step = 0
        buf = PriorityBuffer if params.use_priority else ExperienceBuffer
        # packed_obs: last_obs and the observation rings hold bit-packed rows (bitpack.py); either form is accepted on input
        self.packed = bool(params.packed_obs)
        self.experience = buf(obs_len, self.n_actions, 1, params.experience_buffer_size, device=self.device,
                              seed=params.seed, packed=self.packed)
        self.experience.track_wp = params.n_step > 1
        # the reference keeps last_obs as float64 [N, obs_len] (172 MB at 32k games, C-13); int8 (or packed bits) here
        self.last_obs = (torch.zeros((n_games, bitpack.words_for(obs_len)), dtype=torch.int32, device=self.device) if self.packed
                         else torch.zeros((n_games, obs_len), dtype=torch.int8, device=self.device))
        self.requires_vectorized_observation = lambda: True
        self._gen = torch.Generator(device=self.device).manual_seed(params.seed + 1)
        self._last_loss = None
        # fused HIP actor tail / replay insert (hanabi_hip.ops) on the GPU; plain torch ops elsewhere
        self._fused = self.device.type == "cuda" and self.distributional
        # scalar-Q net (BASELINE config 2) on the GPU with bf16 GEMM inputs and one hidden layer: the hidden layer runs on the same
        # hand-written MFMA kernel as the C51 actor, the tiny [N, H] x [H, A] output layer on the library GEMM, the selection on
        # hb_policy_select
        self._plain_fast = (self.device.type == "cuda" and not self.distributional and params.compute_dtype == "bfloat16"
                            and len(params.layers) == 1 and params.layers[0] % 256 == 0)
        self._plain_actor = None
        self._eff_cache = None      # effective (merged) weights of the online net in the GEMM dtype
        self._trg_cache = None      # same for the target net (refreshed in place at every target sync)
        self._x_act = None          # persistent (padded) first-GEMM operand of the actor
        self._pending = None          # (all-reduce handle, eager part-2 arguments) between update_begin / update_finish
        self._disc = params.discount  # scalar gamma, or the [B] tensor gamma^m of the current n-step batch
        self._fl = None             # FusedLearner (GPU, C51, one hidden layer), built at the first update
        self._fv = None             # FusedVanillaLearner (GPU, scalar double-DQN, uniform replay), built at the first update
        self.use_fused_learner = use_fused_learner
        self.use_mfma_actor = True   # csrc/actor.hip when the FusedLearner can feed it; False = cast + library GEMMs + hb_policy_act
        self._draws = 0             # Philox draw counter of the fused sampler
        self.first_game_id = 0      # global id of game 0 (rank * n_games when sharded), keys the sampler's RNG
        self._is_max_local = self._is_max_global = None   # params.global_is_max: per-rank / all-rank IS normalisers
        # params.actor_lag = 1 (asynchronous actor): see update_begin() and hanabi_hip.selfplay for the ordering rules
        self.actor_lag = int(params.actor_lag)
        if self.actor_lag and not (self._fused and use_fused_learner and params.use_priority and len(params.layers) == 1):
            raise ValueError("actor_lag=1 needs the HIP fused learner (GPU, C51 head, one hidden layer) with prioritized replay")
        self.force_collective = False   # see _collective()
        self._last_coll = False
        self._ar_avg = None             # gradient all-reduce averages inside the collective (RCCL) or sums (gloo: divided after)
        self._support0 = None       # atoms[0], contiguous (the support every action shares)
        self._dense_call = None     # add_experience_dense: (input addresses, rows, launcher, fixed arguments, stream getter)
        # split update (always with actor_lag; set_split_update() for synchronous agents driven with a learner stream): the
        # sum tree is written only on the stream the updates run on, sampling + gather is its own launch followed by
        # `gathered_ev`, and the priority write-back runs after `weights_ev` — so the acting stream may insert as soon as the
        # rings have been read and act as soon as Adam has written the weights, instead of waiting for the whole update
        self.split_update = bool(self.actor_lag)
        self.two_graphs = False   # set_two_graphs()
        self._pending_fills = []    # (start, rows) of inserts whose sum-tree leaves the NEXT update_begin() sets
        self.gathered_ev = None     # recorded when an update has finished reading the replay rings
        self.weights_ev = None      # recorded when an update's optimizer step is done (before its priority write-back)

    @property
    def last_loss(self):
        """mean(td * w_IS) of the most recent update (rlax_rainbow.py:196)."""
        if self._fl is not None and self._last_loss is None:
            return self._fl.loss()
        if self._fv is not None and self._last_loss is None:
            return self._fv.loss()
        return self._last_loss

    @last_loss.setter
    def last_loss(self, value):
        self._last_loss = value

    # ---- helpers ------------------------------------------------------------------------------------
    def _unpack(self, observations) -> Tuple[torch.Tensor, torch.Tensor, bool]:
        obs, legal = observations[1]
        on_device = isinstance(obs, torch.Tensor)   # tensors in -> tensor actions out; numpy in -> numpy out (the reference)
        as_dev = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(self.device)
        return as_dev(obs), as_dev(legal), on_device

    def _vec(self, x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype)
        return torch.as_tensor(np.asarray(x)).to(device=self.device, dtype=dtype)

    def _uniform(self, n):
        return torch.rand(n, device=self.device, generator=self._gen)

    def _net_input(self, obs):
        return self._obs_int8(obs).to(torch.float32)

    def _obs_int8(self, obs):
        """The reference's [N, obs_len] 0/1 layout of an observation batch given in either form."""
        return bitpack.unpack(obs, self.obs_len) if bitpack.is_packed(obs, self.obs_len) else obs

    def _obs_store(self, obs):
        """The form last_obs and the rings keep: packed int32 rows (params.packed_obs) or int8."""
        return self.experience.obs_rows(obs)

    def _effective_weights(self):
        """[(W [in,out], bias [out])] of the online net in the GEMM dtype; recomputed only after the weights
        or the noise changed (the reference's noise is frozen, App. C-2, so acting re-uses them all the time)."""
        if self._fused_learner() is not None:
            return self._fl.eff  # padded GEMM operands, kept current by hb_noisy_adam after every update
        if self._eff_cache is None:
            cd = _DTYPES[self.params.compute_dtype]
            self._eff_cache = [tuple(t.to(cd).contiguous() for t in layer.effective()) for layer in self.online.layers]
        return self._eff_cache

    def _act_plain(self, obs, legal, epsilon):
        """Vanilla double-DQN head (rlax_dqn.py:26-33 MLP): q = relu(obs @ W1 + b1) @ W2 + b2, illegal moves masked, epsilon-greedy."""
        from hanabi_hip import _capi as K
        from hanabi_hip.ops import ActorMFMA

        hidden = self.params.layers[0]
        kp = (self.obs_len + 63) // 64 * 64
        pa = self._plain_actor
        if pa is None:
            pa = self._plain_actor = ActorMFMA(self.obs_len, hidden, self.n_actions, 2, kp, self.device)   # (only its hidden half is used)
            pa.stale = True
            pa.graph = pa.graph_seen = None
        fv = self._fv
        if (fv is not None and fv.cd == torch.bfloat16 and self._graphs_enabled() and obs.is_contiguous()
                and os.environ.get("HB_PLAIN_ACT_GRAPH", "1") != "0"):
            # Config 2 is host-bound (4 096 games: ~0.17 ms of Python per step for ~0.1 ms of GPU work): the weight refresh (one
            # transposer launch + two strided copies from the learner's persistent bf16 operands), the hidden-layer kernel, the
            # cast and the tiny output GEMM are replayed as ONE HIP graph once the same observation buffer has been seen three
            # times (lock-step self-play passes the env's persistent rows); only the selection, whose draw counter and epsilon
            # change per call, stays a launch of its own. Same kernels, same operands: identical q values and moves.
            key = (obs.data_ptr(), obs.shape[0], obs.dtype, id(fv), fv.w1cat.data_ptr())
            pg = getattr(pa, "graph", None)
            if pg is not None and pg[0] == key:
                pg[1].replay()
                pa.stale, self._eff_cache = False, True
                return self._plain_select(pg[2], legal, epsilon, obs.shape[0])
            seen = getattr(pa, "graph_seen", None)
            pa.graph_seen = (key, seen[1] + 1) if seen is not None and seen[0] == key else (key, 1)
            if pa.graph_seen[1] >= 3 and getattr(pa, "fv_jobs", None) is not None and pa.h is not None and pa.h.shape[0] == obs.shape[0]:
                n = obs.shape[0]
                packed = obs.dtype == torch.int32
                fn = K.lib().hb_actor_hidden_packed if packed else K.lib().hb_actor_hidden
                qbuf = torch.empty(n, self.n_actions, dtype=torch.float32, device=self.device)

                def forward():
                    st = K.current_stream()
                    K.check(K.lib().hb_actor_pack_weights(pa.fv_jobs, 1, st))
                    pa.w2f.copy_(fv.w2st[0][:, 0:2 * self.n_actions:2])
                    pa.b2f.copy_(fv.b2st[0][0:2 * self.n_actions:2])
                    K.check(fn(K.dptr(obs), n, self.obs_len, K.dptr(pa.w1t), kp, K.dptr(pa.b1), hidden, K.dptr(pa.h), st))
                    torch.addmm(pa.b2f, pa.h.float(), pa.w2f, out=qbuf)

                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    forward()
                pa.graph = (key, g, qbuf)
                g.replay()
                pa.stale, self._eff_cache = False, True
                return self._plain_select(qbuf, legal, epsilon, n)
        if (pa.stale or self._eff_cache is None) and fv is not None and fv.cd == torch.bfloat16:
            # the learner keeps bf16 copies of the online weights current: transpose W1 from there (one launch, cached job
            # table) and refresh the fp32 image of the bf16 output layer (two strided copies) — no allocation, no casts
            if getattr(pa, "fv_jobs", None) is None:
                pa.graph = pa.graph_seen = None   # (a captured forward holds the old buffers)
                pa.w2f = torch.empty(hidden, self.n_actions, dtype=torch.float32, device=self.device)
                pa.b2f = torch.empty(self.n_actions, dtype=torch.float32, device=self.device)
                jobs = (K.HbPackJob * 1)()
                j = jobs[0]
                j.w, j.bias, j.wt, j.bias_out = fv.w1cat.data_ptr(), fv.b1cat.data_ptr(), pa.w1t.data_ptr(), pa.b1.data_ptr()
                j.k_rows, j.n_cols, j.w_ld, j.group_cols, j.k_pad = self.obs_len, hidden, fv.w1cat.stride(0), 0, kp
                pa.fv_jobs = jobs
            K.check(K.lib().hb_actor_pack_weights(pa.fv_jobs, 1, K.current_stream()))
            pa.w2f.copy_(fv.w2st[0][:, 0:2 * self.n_actions:2])
            pa.b2f.copy_(fv.b2st[0][0:2 * self.n_actions:2])
            pa.stale = False
            self._eff_cache = True
        if pa.stale or self._eff_cache is None:
            pa.fv_jobs = None   # (this path re-creates w2f / b2f)
            pa.graph = pa.graph_seen = None
            w1, w2 = self.online.weights[0].detach(), self.online.weights[1].detach()
            b1, b2 = self.online.biases[0].detach(), self.online.biases[1].detach()
            w1p = torch.zeros(kp, hidden, dtype=torch.bfloat16, device=self.device)
            w1p[:self.obs_len] = w1.to(torch.bfloat16)
            jobs = (K.HbPackJob * 1)()
            j = jobs[0]
            b1h = b1.to(torch.bfloat16).contiguous()
            j.w, j.bias, j.wt, j.bias_out = w1p.data_ptr(), b1h.data_ptr(), pa.w1t.data_ptr(), pa.b1.data_ptr()
            j.k_rows, j.n_cols, j.w_ld, j.group_cols, j.k_pad = self.obs_len, hidden, hidden, 0, kp
            K.check(K.lib().hb_actor_pack_weights(jobs, 1, K.current_stream()))
            pa.w2f, pa.b2f = w2.to(torch.bfloat16).float().contiguous(), b2.float().contiguous()
            pa.stale = False
            self._eff_cache = True
        n = obs.shape[0]
        if pa.h is None or pa.h.shape[0] != n:
            pa.h = torch.empty(n, hidden, dtype=torch.bfloat16, device=self.device)
        packed = obs.dtype == torch.int32
        L, s = K.lib(), K.current_stream()
        fn = L.hb_actor_hidden_packed if packed else L.hb_actor_hidden
        K.check(fn(K.dptr(obs.contiguous()), n, self.obs_len, K.dptr(pa.w1t), kp, K.dptr(pa.b1), hidden, K.dptr(pa.h), s))
        q = torch.addmm(pa.b2f, pa.h.float(), pa.w2f)                      # [N, A] fp32: a tiny GEMM
        return self._plain_select(q, legal, epsilon, n)

    def _plain_select(self, q, legal, epsilon, n):
        from hanabi_hip import _capi as K

        L, s = K.lib(), K.current_stream()
        actions = torch.empty(n, dtype=torch.int32, device=self.device)
        self._draws += 1
        K.check(L.hb_policy_select(K.dptr(q), K.dptr(legal.to(torch.int8).contiguous()), n, self.n_actions, float(epsilon),
                                   self.params.seed + 0x9E3779B9, self._draws, self.first_game_id, K.dptr(actions), s))
        self._last_q = q
        return actions

    def _wait_for_weights(self):
        """split update of a synchronous agent: the policy call waits (on the calling stream) for the last optimizer step's
        event — not for the priority write-back that follows it on the learner stream."""
        if self.split_update and not self.actor_lag and self.weights_ev is not None:
            self.weights_ev.wait()

    def _act_fused(self, obs, legal, epsilon):
        """Actor on the GPU: int8 obs -> GEMM dtype, two bias-fused MFMA GEMMs, then ONE kernel for
        softmax-expectation + legal mask + epsilon-greedy sample (hb_policy_act)."""
        from hanabi_hip import ops

        if self.params.resample_noise:
            self.online.resample()
            self._eff_cache = None
        self._wait_for_weights()
        eff = self._effective_weights()
        fl = self._fused_learner() if self.actor_lag else self._fl
        if fl is not None and fl.actor is not None and self.use_mfma_actor and fl.actor.accepts(obs):
            # hand-written MFMA path: int8 or bit-packed observations in, actions out; no bf16 copy of the observations and no
            # logits in HBM
            wset = fl.acting_set()
            self._draws += 1
            if self._support0 is None:
                self._support0 = self.atoms[0].contiguous()
            return fl.actor.act(obs.contiguous(), legal.to(torch.int8).contiguous(), self._support0, epsilon,
                                self.params.seed + 0x9E3779B9, self._draws, self.first_game_id, s=wset)
        if self.actor_lag:
            raise RuntimeError("actor_lag=1: the MFMA actor takes int8 or bit-packed observations")
        obs = self._obs_int8(obs)
        cd = eff[0][0].dtype
        kp = eff[0][0].shape[0]                     # first-layer K, possibly padded (FusedLearner keeps padded operands)
        if obs.dtype == torch.int8 and cd != torch.float32:
            if self._x_act is None or self._x_act.shape != (obs.shape[0], kp) or self._x_act.dtype != cd:
                self._x_act = torch.zeros(obs.shape[0], kp, dtype=cd, device=self.device)   # pad columns stay zero
            x = ops.obs_cast(obs.contiguous(), cd, out=self._x_act)
        else:
            x = obs.to(cd)
            if kp != x.shape[1]:
                x = torch.nn.functional.pad(x, (0, kp - x.shape[1]))
        for i, (w, b) in enumerate(eff):  # bias (+ ReLU) ride in the GEMM epilogue
            x = torch._addmm_activation(b, x, w, use_gelu=False) if i < len(eff) - 1 else torch.addmm(b, x, w)
        self._draws += 1
        return ops.policy_act(x, legal.to(torch.int8).contiguous(), self.atoms[0].contiguous(), epsilon,
                              self.params.seed + 0x9E3779B9, self._draws, self.first_game_id)

    def q_for_step(self, observations, explore=True):
        """explore() / exploit() split in two for drivers that fuse the epsilon-greedy selection into the env step
        (HanabiEnv.step_select): runs the network and returns (q [N, A] fp32, epsilon, seed, draw, first_game_id) — exactly what
        hb_policy_select would have been given — or None when this agent has no MFMA actor (then call explore())."""
        if not self._fused or self.params.resample_noise or not self.use_mfma_actor:
            return None
        obs = observations[1][0]
        if not isinstance(obs, torch.Tensor) or obs.device != self.device:
            obs = self._unpack(observations)[0]
        fl = self._fused_learner() if self.actor_lag else self._fl
        if fl is None or fl.actor is None or not fl.actor.accepts(obs):
            return None
        self._wait_for_weights()
        wset = fl.acting_set()
        self._draws += 1
        if self._support0 is None:
            self._support0 = self.atoms[0].contiguous()
        q = fl.actor.q_values(obs, self._support0, s=wset)   # (launches only: nothing here is recorded by autograd)
        eps = float(self.params.epsilon(self.train_step)) if explore else 0.0
        return q, eps, self.params.seed + 0x9E3779B9, self._draws, self.first_game_id

    def act_for_step(self, observations, explore=True, actions_out=None):
        """explore() / exploit() for lock-step drivers whose observations already are device tensors: when the policy call runs on
        the one-kernel actor (hb_actor_fused_act: forward, C51 expectation and epsilon-greedy selection in one launch) this returns
        the chosen moves (int32 [N], written into `actions_out` when given); otherwise None (then use q_for_step / explore)."""
        if not self._fused or self.params.resample_noise or not self.use_mfma_actor:
            return None
        obs, legal = observations[1]
        if not isinstance(obs, torch.Tensor) or obs.device != self.device or not isinstance(legal, torch.Tensor):
            return None
        fl = self._fused_learner() if self.actor_lag else self._fl
        if fl is None or fl.actor is None or not fl.actor.takes_fused(obs) or legal.dtype != torch.int8:
            return None
        self._wait_for_weights()
        wset = fl.acting_set()
        self._draws += 1
        if self._support0 is None:
            self._support0 = self.atoms[0].contiguous()
        eps = float(self.params.epsilon(self.train_step)) if explore else 0.0
        return fl.actor.act(obs, legal, self._support0, eps, self.params.seed + 0x9E3779B9, self._draws, self.first_game_id, s=wset,
                            actions_out=actions_out)

    # ---- acting (rlax_rainbow.py:277-290) ---------------------------------------------------------------
    @torch.no_grad()
    def exploit(self, observations):
        obs, legal, on_device = self._unpack(observations)
        if self._fused or self._plain_fast:
            actions = (self._act_fused if self._fused else self._act_plain)(obs, legal, 0.0)
            return actions if on_device else actions.cpu().numpy()
        if self.params.resample_noise:
            self.online.resample()
        actions = DQNPolicy.eval_policy(self.online, self.atoms, self._net_input(obs), legal, self._uniform(obs.shape[0]),
                                        self.distributional)
        return actions if on_device else actions.cpu().numpy()

    @torch.no_grad()
    def explore(self, observations):
        obs, legal, on_device = self._unpack(observations)
        if self._fused or self._plain_fast:
            actions = (self._act_fused if self._fused else self._act_plain)(obs, legal, float(self.params.epsilon(self.train_step)))
            return actions if on_device else actions.cpu().numpy()
        if self.params.resample_noise:
            self.online.resample()
        n = obs.shape[0]
        _, actions = DQNPolicy.policy(self.online, self.atoms, float(self.params.epsilon(self.train_step)),
                                      self._net_input(obs), legal, self._uniform(n), self._uniform(n), self.distributional)
        return actions if on_device else actions.cpu().numpy()

    # ---- recording (rlax_rainbow.py:292-308) --------------------------------------------------------------
    def add_experience_first(self, observations, step_types):
        obs, _, _ = self._unpack(observations)
        first = self._vec(step_types, torch.int64) == 0
        self.last_obs = torch.where(first[:, None], self._obs_store(obs), self.last_obs)
        self._dense_call = None

    def add_experience(self, observations, actions, rewards, step_types):
        obs, legal, _ = self._unpack(observations)
        st = self._vec(step_types, torch.int64)
        not_first = st != 0
        idx = torch.nonzero(not_first, as_tuple=False)[:, 0]  # keeps row order, like boolean-mask indexing
        obs8 = self._obs_store(obs)
        if idx.numel():
            self.experience.add_transitions(
                self.last_obs.index_select(0, idx),
                self._vec(actions, torch.int64).index_select(0, idx).reshape(-1, 1),
                self._vec(rewards, torch.float32).index_select(0, idx).reshape(-1, 1),
                obs8.index_select(0, idx),
                legal.index_select(0, idx),
                (st.index_select(0, idx) == 2).reshape(-1, 1))
        self.last_obs = torch.where(not_first[:, None], obs8, self.last_obs)
        self._dense_call = None

    def add_experience_dense(self, observations, actions, rewards, step_types):
        """add_experience for callers that guarantee no FIRST rows (lock-step self-play after the first
        round): every row is a transition, so nothing is compacted and no device->host sync happens."""
        c = self._dense_call
        if c is not None:
            # lock-step drivers pass the same device buffers every time: the checked, converted argument list of the previous
            # call is reused when every input still lives at the same address (a Python-side saving of ~10 us per step)
            o, l = observations[1]
            try:
                same = (o.data_ptr(), l.data_ptr(), actions.data_ptr(), rewards.data_ptr(), step_types.data_ptr(),
                        self.last_obs.data_ptr(), self.experience._obs_t_buf.data_ptr()) == c[0] and o.shape[0] == c[1]
            except AttributeError:
                same = False
            if same:
                buf, n = self.experience, c[1]
                start = buf.oldest_entry
                if self.params.use_priority:
                    if self.split_update:
                        self._queue_fill(start, n)
                    else:
                        buf.sum_tree.fill_range_dev(start, n, buf._max_priority)
                c[2](*c[3], start, c[4]())
                buf._advance(n)
                return
        obs, legal, _ = self._unpack(observations)
        if self.device.type == "cuda":
            from hanabi_hip import ops

            buf, n = self.experience, obs.shape[0]
            start = buf.oldest_entry
            if self.params.use_priority:  # new leaves enter at max priority (priority_buffer.py:29-32)
                if self.split_update:
                    # the sum tree is written ONLY on the stream the updates run on. The leaves of these rows are set by the
                    # next update_begin(), i.e. after the priority write-back of the update in flight and before the next
                    # sampling: the same order of tree writes as in the plain sequential form.
                    self._queue_fill(start, n)
                else:
                    buf.sum_tree.fill_range_dev(start, n, buf._max_priority)
            ins = (self.last_obs, self._obs_store(obs).contiguous(), legal.to(torch.int8).contiguous(),
                   self._vec(actions, torch.int32).contiguous(), self._vec(rewards, torch.float32).contiguous(),
                   self._vec(step_types, torch.int8).contiguous())
            ops.replay_insert(*ins, buf, start)
            buf._advance(n)
            o, l = observations[1]
            if (isinstance(o, torch.Tensor) and all(isinstance(t, torch.Tensor) for t in (l, actions, rewards, step_types))
                    and (ins[1].data_ptr(), ins[2].data_ptr(), ins[3].data_ptr(), ins[4].data_ptr(), ins[5].data_ptr())
                    == (o.data_ptr(), l.data_ptr(), actions.data_ptr(), rewards.data_ptr(), step_types.data_ptr())):
                # nothing had to be converted: remember the call (ops.replay_insert has validated these very buffers)
                self._dense_call = ((o.data_ptr(), l.data_ptr(), actions.data_ptr(), rewards.data_ptr(), step_types.data_ptr(),
                                     self.last_obs.data_ptr(), buf._obs_t_buf.data_ptr()), n,
                                    *ops.replay_insert_call(*ins, buf))
            return
        obs8 = self._obs_store(obs)
        st = self._vec(step_types, torch.int64)
        self.experience.add_transitions(self.last_obs, self._vec(actions, torch.int64).reshape(-1, 1),
                                        self._vec(rewards, torch.float32).reshape(-1, 1), obs8, legal,
                                        (st == 2).reshape(-1, 1))
        self.last_obs.copy_(obs8)  # the caller's buffer is rewritten in place by the next env step

    # ---- learning (rlax_rainbow.py:310-339) -----------------------------------------------------------------
    def _sample(self):
        indices, prios = self._sample_indices()
        if self.params.n_step > 1:
            tr, self._disc = self.experience.gather_nstep_dev(indices, self.params.n_step, self.params.discount)
        else:
            tr, self._disc = self.experience.gather_dev(indices), self.params.discount
        return indices, prios, tr

    def update(self):
        """Make one training step."""
        self.update_begin()
        self.update_finish()

    # Two-phase form of update() for data-parallel runs: update_begin() samples, computes the loss gradient into
    # the flat buffer and STARTS the RCCL all-reduce without blocking; update_finish() waits for it and applies
    # Adam / priorities / target sync. A driver that alternates agents (hanabi_hip.selfplay) runs the next seat's
    # replay insert, policy and env step between the two calls, so the ~3.4 MB gradient exchange over xGMI hides
    # behind ~0.2 ms of independent work. With one rank the pair is exactly update() — and, under HIP graphs, the whole
    # update (Adam included) is one graph launched by update_begin(): do not let anything read the weights between the two.
    def set_two_graphs(self, on=True):
        """Capture the update as TWO graphs — everything that only READS the weights (forward, loss, backward), then the optimizer
        step with the weight packs — although no collective sits between them: a driver can then start the first half BEFORE this
        agent's policy call has finished (both only read the weights) and hold just the second half back (SelfPlaySession's
        early update, hb_chain_run). Synchronous split-update agents only; results do not depend on it."""
        on = bool(on)
        if on != self.two_graphs:
            assert self._pending is None
            self.two_graphs = on
            self._graph1 = self._graph2 = None
        return on

    def set_split_update(self, on=True):
        """Turn the split form of update() on (see __init__) for a synchronous agent; returns whether it is in effect. Needs
        the fused learner with prioritized replay; agents with actor_lag always use it. Results do not depend on it."""
        on = bool(on) or bool(self.actor_lag)
        if on and not (self._fused and self.use_fused_learner and self.params.use_priority and len(self.params.layers) == 1
                       and not self.params.resample_noise and self._graphs_enabled()):
            return False
        if on != self.split_update:
            assert self._pending is None and not self._pending_fills
            self.split_update = on
            self._graph1 = self._graph2 = None   # the captured update has the other shape: capture again at the next update
        return on

    def _queue_fill(self, start, n):
        """split update: remember that rows [start, start + n) of the ring were inserted. Contiguous inserts merge into one range
        (a seat that inserts without training — train=False, or a seat outside train_seats — would otherwise grow the list by one
        entry, and the next drain by one launch, per step); once a whole ring's worth is queued the list is the whole ring.
        Until the list is drained (update_begin / sync / checkpoint_state) the sum tree does not know these rows: code that reads
        the tree directly between steps must call sync() first."""
        cap = self.experience.capacity
        pf = self._pending_fills
        if pf and pf[-1][0] + pf[-1][1] == start and pf[-1][1] + n <= cap:
            pf[-1] = (pf[-1][0], pf[-1][1] + n)
        else:
            pf.append((start, n))
        if len(pf) > 1 and sum(x[1] for x in pf) >= cap:
            pf[:] = [(0, cap)]

    def apply_pending_fills(self):
        """split update: set the sum-tree leaves of the rows inserted since the last update (on the CURRENT stream, which must be
        the one the updates run on)."""
        for start, n in self._pending_fills:
            self.experience.sum_tree.fill_range_dev(start, n, self.experience._max_priority)
        self._pending_fills.clear()

    def _pre_gather(self):
        """split update: sampling + replay gather as their own launch in front of the (captured) rest of the update, followed by
        an event: from there on the update no longer reads the rings, and the acting stream may overwrite their oldest rows."""
        fl = self._fused_learner()
        fl.sample_and_gather(self.params.seed + 0x51ED270B + 0x9E3779B1 * self.first_game_id)
        if self.gathered_ev is None:
            from hanabi_hip import _capi as K

            self.gathered_ev = K.Event()
        self.gathered_ev.record()

    def update_begin(self):
        assert self._pending is None, "update_finish() of the previous update has not been called"
        if self.split_update:
            self.apply_pending_fills()
        self.experience.sync_size()
        beta = float(self.params.beta_is(self.train_step))
        if beta != self._beta_host:  # device scalar read inside the captured graph: refreshed only when it changes
            self._beta.fill_(beta)
            self._beta_host = beta
        if self._graphs_enabled():
            if self._graph1 is None:
                self._capture_update_graphs()
            if self.split_update:
                self._pre_gather()
            self._graph1.replay()
            part2_args = None
        else:
            if self.split_update:
                self._pre_gather()
            self.last_loss, indices, new_prios = self._update_part1()
            part2_args = (indices, new_prios)
        self._pending = (self._allreduce_gradients(async_op=True), part2_args)

    def update_finish(self):
        if self._pending is None:
            return
        work, part2_args = self._pending
        self._pending = None
        self._finish_allreduce(work)
        if part2_args is None:
            if self._graph2 is not None:  # (single rank: the second half was captured into the first graph)
                self._graph2.replay()
            part2_args = (self._g_idx, self._g_prios)
        else:
            self._update_part2(*part2_args)
        self._eff_cache = None
        if self._fl is not None:
            self._fl.weights_updated()  # Adam rewrote the effective weights: the actor's copies follow (lazily, or now: actor_lag)
        if self.split_update:
            self._after_optimizer_step(*part2_args)
        if self.train_step % self.params.target_update_period == 0:  # after the step, including step 0 (C-10)
            self._sync_target()
        self.train_step += 1

    def _sync_target(self):
        with torch.no_grad():
            for dst, src in zip(list(self.target.parameters()) + list(self.target.buffers()),
                                list(self.online.parameters()) + list(self.online.buffers())):
                dst.copy_(src)
        self._refresh_target_cache()

    def _refresh_target_cache(self):
        if self._fl is not None:
            self._fl.refresh_target()
        if self._fv is not None:
            self._fv.refresh_target()
        if self._trg_cache is None:
            return
        with torch.no_grad():  # IN PLACE: captured graphs hold these tensors
            cd = _DTYPES[self.params.compute_dtype]
            for (w_c, b_c), layer in zip(self._trg_cache, self.target.layers):
                w, bias = layer.effective()
                w_c.copy_(w.to(cd))
                b_c.copy_(bias.to(cd))

    # The update is split where the (optional) collective sits:
    #   part 1: PER sample -> gather -> 3 forwards -> loss -> backward into ONE flat gradient buffer
    #   [world > 1: one RCCL all-reduce(AVG) of that flat buffer]
    #   part 2: Adam step -> priority update
    # Eagerly that is ~120 kernel launches (host-bound at ~1.7 ms on the MI355X box); on the GPU each part
    # is captured once into a HIP graph and replayed with a single launch.
    def _target_weights(self):
        """Effective weights of the TARGET net in the GEMM dtype; they only change when the target is synced."""
        if self._trg_cache is None:
            cd = _DTYPES[self.params.compute_dtype]
            with torch.no_grad():
                self._trg_cache = [tuple(t.to(cd).contiguous() for t in layer.effective()) for layer in self.target.layers]
        return self._trg_cache

    def _loss_batched(self, tr, prios):
        """Same arithmetic as DQNLearning.loss for the C51 net, arranged for the GPU: the two ONLINE forwards
        (obs_tm1 with gradient, obs_t for action selection) run as one 2B-row pass over effective weights that
        are formed once; the target pass re-uses cached effective weights; observations go int8 -> GEMM dtype
        directly. Roughly 2.5x fewer kernels than three independent module forwards."""
        cd = _DTYPES[self.params.compute_dtype]
        b = tr.observation_tm1.shape[0]
        x = torch.cat([tr.observation_tm1, tr.observation_t], dim=0).to(cd)
        h = x
        layers = self.online.layers
        for i, layer in enumerate(layers):
            w, bias = layer.effective()
            h = torch.addmm(bias.to(cd), h, w.to(cd))
            if i < len(layers) - 1:
                h = torch.relu(h)
        a, k = self.atoms.shape
        logits_on = h.float().view(2 * b, a, k)
        with torch.no_grad():
            ht = x[b:]
            trg = self._target_weights()
            for i, (w, bias) in enumerate(trg):
                ht = torch.addmm(bias, ht, w)
                if i < len(trg) - 1:
                    ht = torch.relu_(ht)
            logits_t = ht.float().view(b, a, k)
        term = tr.terminal_t[:, 0]
        td = L.categorical_double_q_td(logits_on[:b], tr.action_tm1[:, 0].long(), tr.reward_t[:, 0].to(torch.float32),
                                       self._disc, self.atoms, logits_t, logits_on[b:].detach(),
                                       term if self.params.mask_terminal else None)
        w_is = L.is_weights(prios, self._beta)
        return torch.mean(td * w_is), torch.abs(td).detach()

    def _fused_learner(self):
        """hanabi_agents.rlax_dqn.fused_learner.FusedLearner when applicable (GPU, C51 head, one hidden layer)."""
        if self._fl is None and self._fused and self.use_fused_learner and len(self.online.layers) == 2:
            from .fused_learner import FusedLearner

            self._fl = FusedLearner(self)
            if self.params.use_priority:
                # the fused sample + gather launch re-sums the tree's top levels itself: writers can skip that launch
                self.experience.sum_tree.set_lazy_top(True)
        return self._fl

    def _fused_vanilla(self):
        """hanabi_agents.rlax_dqn.fused_vanilla.FusedVanillaLearner when applicable (GPU, scalar head, one hidden layer,
        uniform replay, batch <= 256)."""
        if self._fv is None and self.use_fused_learner:
            from .fused_vanilla import FusedVanillaLearner

            if FusedVanillaLearner.supports(self):
                self._fv = FusedVanillaLearner(self)
        return self._fv

    def _sample_indices(self):
        b = self.params.train_batch_size
        if self.params.use_priority and self._fl is not None:
            # stratified uniforms drawn inside the sampling kernel, keyed by (seed; query, optimizer step): no
            # generator launch, and nothing for HIP-graph replay to re-seed (torch re-fills the generator's seed /
            # offset tensors with two more launches before every replay of a graph that contains torch.rand)
            # (keyed by the shard too: data-parallel ranks share the seed but must not draw the same strata offsets)
            return self.experience.sum_tree.per_sample_philox_dev(self.params.seed + 0x51ED270B + 0x9E3779B1 * self.first_game_id,
                                                                  self._fl.step, b)
        if self.params.use_priority:
            u = torch.rand(b, dtype=torch.float64, device=self.device)  # scaled to [0, 1/B) inside the kernel
            return self.experience.sum_tree.per_sample_dev(u, unit=True)
        return self.experience.sample_indices_dev(b), torch.ones(b, dtype=torch.float64, device=self.device)

    def _update_part1(self):
        fl = self._fused_learner()
        if fl is not None:
            if self.params.resample_noise:
                self.online.resample()
                self.target.resample()
                fl.refresh_effective()
                fl.refresh_target()
            if self.split_update:
                indices, prios, gathered = fl._idx, fl._prob, True   # filled by _pre_gather(), outside the captured part
            elif self.params.use_priority and "_sample_indices" not in self.__dict__:
                # PER: tree descent and replay gather in one launch (same draws as _sample_indices below)
                indices, prios = fl.sample_and_gather(self.params.seed + 0x51ED270B + 0x9E3779B1 * self.first_game_id)
                gathered = True
            else:
                indices, prios = self._sample_indices()
                gathered = False
            self._note_local_is_max(prios)
            td, _ = fl.part1(indices, prios, gathered=gathered)
            return None, indices, td  # the loss value is formed on demand (last_loss) from td and the IS weights
        fv = self._fused_vanilla()
        if fv is not None:
            indices, prios = self._sample_indices()
            td = fv.part1(indices, prios)
            return None, indices, td   # (the loss value is formed on demand: last_loss)
        indices, prios, tr = self._sample()
        self._note_local_is_max(prios)
        if self.params.resample_noise:
            self.online.resample()
            self.target.resample()
            self._trg_cache = None
        if self._fused:
            loss, new_prios = self._loss_batched(tr, prios)
        else:
            tr = tr._replace(observation_tm1=self._net_input(tr.observation_tm1),
                             observation_t=self._net_input(tr.observation_t))
            loss, new_prios = DQNLearning.loss(self.online, self.target, self.atoms, tr, self._disc, prios,
                                               self._beta, self.params.mask_terminal, self.distributional)
        self._flat_grad.zero_()
        loss.backward()  # every p.grad is a view into _flat_grad: gradients accumulate in place
        return loss.detach(), indices, new_prios

    def _after_optimizer_step(self, indices, new_prios):
        """split update: the weights are final here — mark it for the acting stream, THEN write the priorities back (outside
        the captured graph: the acting stream does not wait for this launch)."""
        if self.weights_ev is None:
            from hanabi_hip import _capi as K

            self.weights_ev = K.Event()
        self.weights_ev.record()
        if self.params.use_priority:
            self.experience.update_priorities_dev(indices, new_prios)

    def _update_part2(self, indices, new_prios):
        if self._fl is not None:
            self._fl.part2()
        else:
            self.optimizer.step()
            if self._fv is not None:
                self._fv.refresh_online()   # the GEMM-dtype copies follow the fp32 master weights
        if self.params.use_priority and not self.split_update:
            # (running this on a forked graph branch beside Adam was measured slower: 0.216 vs 0.190 ms per update)
            self.experience.update_priorities_dev(indices, new_prios)

    def _update_eager(self):
        if self.split_update:
            self._pre_gather()
        self.last_loss, indices, new_prios = self._update_part1()
        self._finish_allreduce(self._allreduce_gradients(async_op=False))
        self._update_part2(indices, new_prios)
        if self.split_update and self.params.use_priority:
            self.experience.update_priorities_dev(indices, new_prios)

    def _graphs_enabled(self):
        return self.use_graphs and self.device.type == "cuda"

    # ---- state touched by one update: snapshot / restore around the capture warm-up ---------------------------
    def _learner_state_tensors(self):
        """Every device tensor an update writes (besides scratch): weights, moments, step counters, effective weights,
        PER running max / min. The sum tree is handled separately (its nodes are exported / imported whole)."""
        ts = [p.data for p in self.online.parameters()] + list(self.online.buffers())
        ts += [p.data for p in self.target.parameters()] + list(self.target.buffers())
        fl = self._fl
        if fl is not None:
            seen = set()
            for mv in fl.state.values():
                for t in mv:
                    if id(t) not in seen:
                        seen.add(id(t))
                        ts.append(t)
            ts += [fl.step, fl.w1cat, fl.b1cat, fl.w2st, fl.b2st]
        if self.params.use_priority:
            ts += [self.experience._max_priority, self.experience._min_priority]
        return ts

    def _snapshot_learner_state(self):
        snap = dict(tensors=[t.clone() for t in self._learner_state_tensors()], last_loss=self._last_loss,
                    rng=torch.cuda.get_rng_state(self.device), gen=self._gen.get_state())
        xg = getattr(self.experience, "_gen", None)   # uniform replay draws its batch indices from the buffer's own generator
        if xg is not None:
            snap["xgen"] = xg.get_state()
        if self.params.use_priority:
            snap["tree"] = self.experience.sum_tree.nodes().clone()
        if self._fl is None:
            # torch.optim.Adam creates its state lazily: remember what existed (nothing before the first step)
            snap["adam"] = {id(p): {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                            for p, st in self.optimizer.state.items()}
        return snap

    def _restore_learner_state(self, snap):
        with torch.no_grad():
            for dst, src in zip(self._learner_state_tensors(), snap["tensors"]):
                dst.copy_(src)
            if "tree" in snap:
                self.experience.sum_tree.import_nodes(snap["tree"])
            if "adam" in snap:
                for p, st in self.optimizer.state.items():
                    old = snap["adam"].get(id(p))
                    for k, v in st.items():
                        if isinstance(v, torch.Tensor):  # in place: the captured graph holds these tensors
                            v.zero_() if old is None else v.copy_(old[k])
        torch.cuda.set_rng_state(snap["rng"], self.device)
        self._gen.set_state(snap["gen"])
        if "xgen" in snap:
            self.experience._gen.set_state(snap["xgen"])
        self._last_loss = snap["last_loss"]
        self._eff_cache = None
        if self._fv is not None:
            self._fv.refresh_all()
        if self._fl is not None and self._fl.thin:   # the transposed weight copies follow the restored operands
            self._fl._transpose(0)
            self._fl._transpose(1)
        if self._fl is not None and not self.actor_lag:   # (actor_lag: the warm-up updates never touch the actor's weight sets)
            self._fl.actor_stale = True

    def _capture_update_graphs(self):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        # Warm-up off the capture stream (allocator, GEMM workspaces, lazily created Adam state). The warm-up updates are
        # real updates, so everything they write is put back afterwards: update() stays exactly ONE gradient step, as
        # in the reference (rlax_rainbow.py:310-339), whether or not graphs are in use.
        self._fused_learner()
        with torch.cuda.stream(side):
            snap = self._snapshot_learner_state()
            for _ in range(3):
                self._update_eager()
            self._restore_learner_state(snap)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._graph1, self._graph2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: calls made by OTHER threads while we capture (the RCCL watchdog polling its events in a
        # data-parallel run) must not invalidate the capture
        if not self._collective() and not (self.two_graphs and not self.actor_lag and self.split_update):
            # no collective between the halves: one graph, one launch (graph2 stays empty-handed: see update_finish)
            with torch.cuda.graph(self._graph1, capture_error_mode="thread_local"):
                self.last_loss, self._g_idx, self._g_prios = self._update_part1()
                self._update_part2(self._g_idx, self._g_prios)
            self._graph2 = None
            return
        if self._collective_in_graph():
            # opt-in (HB_DP_GRAPH_COLLECTIVE=1, RCCL only): the all-reduce is captured INTO the graph between the two halves — one
            # launch per update and no torch.distributed call on the host per update (≈50 us). RCCL's kernels are capturable;
            # validated here with a one-rank group only (tests/test_distributed.py), hence not the default.
            import torch.distributed as dist

            flat = self._fl.flat_grad if self._fl is not None else self._flat_grad
            with torch.cuda.graph(self._graph1, capture_error_mode="thread_local"):
                self.last_loss, self._g_idx, self._g_prios = self._update_part1()
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.process_group)
                self._update_part2(self._g_idx, self._g_prios)
            self._graph2 = None
            return
        with torch.cuda.graph(self._graph1, capture_error_mode="thread_local"):
            self.last_loss, self._g_idx, self._g_prios = self._update_part1()
        with torch.cuda.graph(self._graph2, pool=self._graph1.pool(), capture_error_mode="thread_local"):
            self._update_part2(self._g_idx, self._g_prios)

    def _collective_in_graph(self):
        import torch.distributed as dist

        return (os.environ.get("HB_DP_GRAPH_COLLECTIVE") == "1" and self._graphs_enabled() and self._collective()
                and not self.params.global_is_max and dist.get_backend(self.process_group) == "nccl")

    def _dp_world(self):
        import torch.distributed as dist

        if self.process_group is False:   # explicitly local: an agent that must not join the job's collectives
            return 1
        if self.process_group is None and not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.process_group)

    def _collective(self):
        """Does an update go through the all-reduce path (flat gradient bucket, two captured halves around the collective)?
        More than one rank — or `force_collective` with an initialised process group of ONE rank, which runs exactly the
        multi-rank code path (RCCL included) on a one-GPU box (tests/test_distributed.py)."""
        if self._dp_world() > 1:
            return True
        if self.force_collective and self.process_group is not False:
            import torch.distributed as dist

            return dist.is_available() and dist.is_initialized()
        return False

    def _note_local_is_max(self, prios):
        """global_is_max: remember this rank's max_i (1/P_i)^beta (the normaliser its loss used) for the rescale below."""
        if self.params.global_is_max and self._dp_world() > 1:
            if self._is_max_local is None:
                self._is_max_local = torch.ones(1, dtype=torch.float32, device=self.device)
                self._is_max_global = torch.ones(1, dtype=torch.float32, device=self.device)
            self._is_max_local.copy_(((1.0 / prios).to(torch.float32) ** self._beta).max().reshape(1))

    def _allreduce_gradients(self, async_op=False):
        """Data parallelism: ONE all-reduce (sum) of the flat fp32 gradient over RCCL. Returns the
        torch.distributed Work handle when async_op (None with a single rank).

        With params.global_is_max the gradient is first rescaled from the per-rank IS normaliser to the global one:
        w_i / max_rank(w) = (w_i / max_all(w)) * (max_all / max_rank)  =>  g_global = g_rank * max_rank / max_all,
        max_all from one 4-byte all-reduce(MAX) — the reference's single-batch `w /= max(w)` (rlax_rainbow.py:188-189)
        over the global batch, exactly."""
        import torch.distributed as dist

        coll = self._collective()
        if self._fl is not None and self._fl.direct is not None and self._fl.direct != (not coll):
            raise RuntimeError("the process group changed after the FusedLearner was built: its gradient routing "
                               "(direct GEMM outputs vs packed all-reduce bucket) no longer matches the world size")
        if coll and self._graph1 is not None and self._graph2 is None and self._graphs_enabled():
            coll = False   # the collective lives inside the captured graph (_collective_in_graph): nothing to do on the host
        self._last_coll = coll   # (_finish_allreduce of the same update reuses the answer)
        if not coll:
            return None
        flat = self._fl.flat_grad if self._fl is not None else self._flat_grad
        if self.params.global_is_max:
            self._is_max_global.copy_(self._is_max_local)
            dist.all_reduce(self._is_max_global, op=dist.ReduceOp.MAX, group=self.process_group)
            scale = self._is_max_local / self._is_max_global
            flat.mul_(scale)
            if self._fl is not None:
                self._fl.w_is.mul_(scale)   # last_loss = mean(td * w_IS) reports the globally normalised weights
            elif self._last_loss is not None:
                self._last_loss = self._last_loss * scale[0]
        if self._ar_avg is None:   # RCCL averages inside the collective (ncclAvg); gloo has no AVG: sum, then divide
            self._ar_avg = dist.get_backend(self.process_group) == "nccl"
        work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._ar_avg else dist.ReduceOp.SUM, group=self.process_group,
                               async_op=True)
        if not async_op:
            work.wait()
        return work

    def _finish_allreduce(self, work):
        if not self._last_coll:
            return
        if work is not None:
            work.wait()  # orders the current stream after the collective; does not block the host on NCCL/RCCL
        if not self._ar_avg:
            flat = self._fl.flat_grad if self._fl is not None else self._flat_grad
            flat /= self._dp_world()

    # ---- misc --------------------------------------------------------------------------------------------
    def __repr__(self):
        return f"<rlax_dqn.DQNAgent(params={self.params})>"

    def save_weights(self, path, fname_part):
        """Writes rlax_rainbow_<part>_{online,target}.pkl like the reference (rlax_rainbow.py:344-357); the
        payload is a torch state_dict (tensors only), not a pickle of jax arrays."""
        torch.save(self.online.state_dict(), os.path.join(path, "rlax_rainbow_" + fname_part + "_online.pkl"))
        torch.save(self.target.state_dict(), os.path.join(path, "rlax_rainbow_" + fname_part + "_target.pkl"))

    # ---- full checkpoints ------------------------------------------------------------------
    # save_weights above is the reference's format (parameters only). A checkpoint additionally carries the optimizer
    # moments, the replay ring with its sum tree, `last_obs`, the step counters and every RNG state, so that a resumed
    # run continues bit-for-bit (tests/test_checkpoint.py). Everything is a plain dict of tensors / numbers: it
    # loads with torch.load(weights_only=True).
    def checkpoint_state(self, include_replay=True):
        if self._pending_fills:   # split update: leaves of rows inserted since the last update (the caller has joined the streams)
            self.apply_pending_fills()
        sd = dict(format="hanabi-agents_amd/agent/1", params=repr(self.params), train_step=self.train_step,
                  draws=self._draws, first_game_id=self.first_game_id, seed=int(self.params.seed),
                  online={k: v.cpu() for k, v in self.online.state_dict().items()},
                  target={k: v.cpu() for k, v in self.target.state_dict().items()},
                  gen=self._gen.get_state().cpu(), last_obs=self.last_obs.cpu(),
                  experience=self.experience.state_dict(include_replay),
                  # the device's default generator (replay sampling, noise resampling): process-wide, saved with every agent
                  rng=(torch.cuda.get_rng_state(self.device) if self.device.type == "cuda" else torch.get_rng_state()).cpu())
        fl = self._fl
        if fl is not None:
            sd["fused"] = dict(step=fl.step.cpu(),
                               moments={f"{li}.{name}.{k}": t.cpu() for (li, name), mv in fl.state.items()
                                        for k, t in zip("mv", mv)},
                               eff=[t.cpu() for pair in fl.eff for t in pair], trg=[t.cpu() for pair in fl.trg for t in pair])
            if self.actor_lag:   # the set the policy reads is one update behind `eff`: it is state of its own
                fl.pack_actor()
                sd["fused"]["actor_sets"] = [t.cpu() for t in fl.actor.state_tensors()]
                sd["fused"]["n_packed"] = fl.n_packed
        else:
            sd["optimizer"] = self.optimizer.state_dict()
        return sd

    def load_checkpoint_state(self, sd):
        """In place: tensors captured by the update graphs keep their addresses."""
        if sd.get("format") != "hanabi-agents_amd/agent/1":
            raise ValueError("not an agent checkpoint")
        if self._pending is not None:
            raise RuntimeError("update_finish() pending")
        passes = 1
        if "fused" in sd and self._fused_learner() is None:
            raise ValueError("checkpoint was written by the fused learner, which this agent cannot build")
        for _ in range(2):
            self.online.load_state_dict(sd["online"])
            self.target.load_state_dict(sd["target"])
            self._gen.set_state(sd["gen"].cpu())
            self.last_obs.copy_(sd["last_obs"])
            self.experience.load_state_dict(sd["experience"])
            self.train_step, self._draws = int(sd["train_step"]), int(sd["draws"])
            self.first_game_id = int(sd["first_game_id"])
            self.params = self.params._replace(seed=int(sd["seed"]))  # keys the exploration draws (Philox)
            if "fused" in sd:
                fl, f = self._fl, sd["fused"]
                fl.step.copy_(f["step"])
                for (li, name), mv in fl.state.items():
                    for k, t in zip("mv", mv):
                        t.copy_(f["moments"][f"{li}.{name}.{k}"])
                for dst, src in zip([t for pair in fl.eff for t in pair], f["eff"]):
                    dst.copy_(src)
                for dst, src in zip([t for pair in fl.trg for t in pair], f["trg"]):
                    dst.copy_(src)
                fl.actor_stale = True
                if fl.thin:
                    fl._transpose(0)
                    fl._transpose(1)
                if self.actor_lag:
                    if "actor_sets" not in f:
                        raise ValueError("checkpoint was written with actor_lag=0")
                    for dst, src in zip(fl.actor.state_tensors(), f["actor_sets"]):
                        dst.copy_(src)
                    fl.n_packed, fl.packed_ev, fl.actor_stale = int(f["n_packed"]), [None, None], False
                    self._pending_fills = []   # (checkpoint_state drains the list before it writes the tree)
            else:
                self.optimizer.load_state_dict(sd["optimizer"])
                self._trg_cache = None
                if self._fv is not None:
                    self._fv.refresh_all()
            self._eff_cache = None
            # Capturing the update graphs runs warm-up updates, which would otherwise happen (and change the weights)
            # at the first update() after the resume: capture now, on the restored replay, then restore once more.
            if (passes == 1 and self._graphs_enabled() and self._graph1 is None
                    and self.experience.size >= self.params.train_batch_size):
                self.experience.sync_size()
                self._beta_host = float(self.params.beta_is(self.train_step))
                self._beta.fill_(self._beta_host)
                self._capture_update_graphs()
                passes = 2
                continue
            break
        if self.device.type == "cuda":
            torch.cuda.set_rng_state(sd["rng"].cpu(), self.device)
        else:
            torch.set_rng_state(sd["rng"].cpu())

    def save_checkpoint(self, path, include_replay=True):
        torch.save(self.checkpoint_state(include_replay), path)

    def load_checkpoint(self, path):
        self.load_checkpoint_state(torch.load(path, map_location="cpu", weights_only=True))

    def restore_weights(self, online_weights_file, trg_weights_file):
        self._eff_cache = None
        self.online.load_state_dict(torch.load(online_weights_file, map_location=self.device, weights_only=True))
        self.target.load_state_dict(torch.load(trg_weights_file, map_location=self.device, weights_only=True))
        if self._fl is not None:
            self._fl.refresh_effective()
        if self._fv is not None:
            self._fv.refresh_online()
        self._refresh_target_cache()
