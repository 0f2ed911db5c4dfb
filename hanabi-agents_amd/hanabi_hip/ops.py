"""Thin wrappers over the actor and replay-insert kernels (hb_policy_act, hb_actor_*, hb_replay_insert)."""
import ctypes
import os

import torch

from . import _capi as K

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def policy_act(logits, legal, support, epsilon, seed, draw, first_game_id=0, actions_out=None, q_out=None):
    """logits [N, >= A*K] (f32/bf16/f16, contiguous CUDA; extra columns are GEMM padding), legal [N, A] int8,
    support [K] f32 -> actions int32 [N]."""
    n, a = legal.shape
    k = support.numel()
    assert logits.is_cuda and logits.is_contiguous() and logits.shape[0] == n and logits.shape[1] >= a * k and logits.dtype in _DT
    assert legal.dtype == torch.int8 and legal.is_contiguous() and support.dtype == torch.float32
    actions = actions_out if actions_out is not None else torch.empty(n, dtype=torch.int32, device=logits.device)
    K.check(K.lib().hb_policy_act(K.dptr(logits), _DT[logits.dtype], K.dptr(legal), K.dptr(support), n, a, k,
                                  logits.shape[1], float(epsilon), int(seed), int(draw), int(first_game_id), K.dptr(actions), K.dptr(q_out),
                                  K.current_stream()))
    return actions


def obs_cast(obs, dtype, out=None):
    """int8 0/1 [N, L] -> bf16 / f16 [N, >= L] (one HBM pass). `out` may be wider than L (K padding of the first
    GEMM); its extra columns are not touched and must already be zero."""
    assert obs.dtype == torch.int8 and obs.is_cuda and obs.is_contiguous() and dtype in (torch.bfloat16, torch.float16)
    out = out if out is not None else torch.empty(obs.shape, dtype=dtype, device=obs.device)
    assert out.dtype == dtype and out.is_contiguous() and out.shape[0] == obs.shape[0] and out.shape[1] >= obs.shape[1]
    K.check(K.lib().hb_obs_cast(K.dptr(obs), K.dptr(out), _DT[dtype], obs.shape[0], obs.shape[1], out.shape[1],
                                K.current_stream()))
    return out


def replay_insert(last_obs, obs, legal, actions, rewards, step_types, ring, start):
    """ring: object with _obs_tm1_buf, _obs_t_buf, _act_tm1_buf, _lms_t_buf, _rew_t_buf, _terminal_t_buf, capacity."""
    odt = ring._obs_tm1_buf.dtype          # int8 rows of obs_len bytes, or bit-packed int32 rows: plain bytes to the kernel
    n, obs_len = obs.shape[0], obs.shape[1] * obs.element_size()
    assert last_obs.shape == obs.shape and ring._obs_tm1_buf.shape[1] == obs.shape[1]
    for t, dt in ((last_obs, odt), (obs, odt), (legal, torch.int8), (actions, torch.int32),
                  (rewards, torch.float32), (step_types, torch.int8)):
        assert t.is_cuda and t.is_contiguous() and t.dtype == dt, (t.dtype, dt)
    K.check(K.lib().hb_replay_insert(K.dptr(last_obs), K.dptr(obs), K.dptr(legal), K.dptr(actions), K.dptr(rewards),
                                     K.dptr(step_types), K.dptr(ring._obs_tm1_buf), K.dptr(ring._obs_t_buf),
                                     K.dptr(ring._act_tm1_buf), K.dptr(ring._lms_t_buf), K.dptr(ring._rew_t_buf),
                                     K.dptr(ring._terminal_t_buf), n, obs_len, legal.shape[1], ring.capacity, int(start),
                                     K.current_stream()))


def replay_insert_call(last_obs, obs, legal, actions, rewards, step_types, ring):
    """(launcher, fixed arguments, stream getter) of the replay_insert() call just made with these buffers: calling
    `launcher(*fixed, start, stream())` repeats it for another ring position without re-validating or re-converting anything."""
    n, obs_len = obs.shape[0], obs.shape[1] * obs.element_size()
    fixed = (last_obs.data_ptr(), obs.data_ptr(), legal.data_ptr(), actions.data_ptr(), rewards.data_ptr(), step_types.data_ptr(),
             ring._obs_tm1_buf.data_ptr(), ring._obs_t_buf.data_ptr(), ring._act_tm1_buf.data_ptr(), ring._lms_t_buf.data_ptr(),
             ring._rew_t_buf.data_ptr(), ring._terminal_t_buf.data_ptr(), n, obs_len, legal.shape[1], ring.capacity)
    fn = K.lib().hb_replay_insert

    def launch(*a):
        K.check(fn(*a))
    return launch, fixed, K.current_stream


class ActorMFMA:
    """The actor's forward pass on the hand-written MFMA kernels (csrc/actor.hip): packed (transposed) copies of
    the effective weights plus the scratch buffers; `pack` after every weight change, `act` per step."""

    def __init__(self, obs_len, hidden, n_actions, n_atoms, k_pad, device, n_sets=1, dtype=torch.bfloat16):
        assert k_pad % 64 == 0 and k_pad >= obs_len and hidden % 256 == 0 and 2 <= n_atoms <= 256
        assert dtype in (torch.bfloat16, torch.float16)
        self.obs_len, self.hidden, self.n_actions, self.n_atoms, self.k_pad = obs_len, hidden, n_actions, n_atoms, k_pad
        self.group_actions = 256 // n_atoms
        groups = -(-n_actions // self.group_actions)
        bf = dict(dtype=dtype, device=device)
        # operand type of the weights and hidden activations: bf16 (both kernel forms) or fp16 (the one-kernel form only — the
        # reference's own network dtype; csrc/actor.hip's two-kernel form exists for bf16)
        self.dtype, self._dt = dtype, 2 if dtype == torch.float16 else 1
        self.two_kernel = dtype == torch.bfloat16
        # n_sets > 1: double-buffered weights (params.actor_lag): the actor reads one set while the learner packs the other
        self.n_sets = int(n_sets)
        self.sets = [(torch.zeros(hidden, k_pad, **bf), torch.zeros(hidden, dtype=torch.float32, device=device),
                      torch.zeros(groups * 256, hidden, **bf), torch.zeros(groups * 256, dtype=torch.float32, device=device))
                     for _ in range(self.n_sets)] if self.two_kernel else []
        self.w1t, self.b1, self.w2t, self.b2 = self.sets[0] if self.two_kernel else (None, None, None, None)
        self.h = self.q = self.actions = None
        self._jobs = {}
        self._q_call = None   # q_values: cached argument addresses
        # True: selection inside the output-layer GEMM (hb_actor_q_select with ticket counters). Bit-identical actions; measured
        # r02: NOT faster (policy call 82 vs 75 us: the selecting workgroups hold their CU's LDS while they run a latency-bound
        # tail, which delays the GEMM's second round of workgroups), so the separate hb_policy_select launch stays the default
        self.fuse_select = False
        self._set_ptrs = [tuple(t.data_ptr() for t in st) for st in self.sets]
        # Round 3: the whole forward as ONE kernel (csrc/actor_fused.hip) for bit-packed observations on the reference topology
        # (hidden 512, 51 atoms): its own fragment-major weight copies per set. HB_ACTOR_FUSED=0 keeps the two-kernel form.
        self.fused = bool(K.lib().hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)) and os.environ.get("HB_ACTOR_FUSED", "1") != "0"
        self.fsets = []
        if self.fused:
            b1, b2, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
            K.check(K.lib().hb_actor_fused_sizes(obs_len, hidden, n_actions, n_atoms, ctypes.byref(b1), ctypes.byref(b2), ctypes.byref(nb)))
            self.fsets = [(torch.zeros(b1.value // 2, **bf), torch.zeros(hidden, dtype=torch.float32, device=device),
                           torch.zeros(b2.value // 2, **bf), torch.zeros(nb.value, dtype=torch.float32, device=device))
                          for _ in range(self.n_sets)]
        self._fset_ptrs = [tuple(t.data_ptr() for t in st) for st in self.fsets]
        self._src = [None] * self.n_sets          # the operands of the last pack() of each set
        self._two_stale = [False] * self.n_sets   # the two-kernel form's copies of that set are behind them (lazy_two_kernel)
        # one workgroup owns 128 rows from start to end, so below ~one workgroup per two CUs the two-kernel form (more, smaller
        # workgroups) is quicker: measured 58 vs 45 us at 4 096 rows, 56 vs 48 at 7 000, equal at 16 384, 49 vs 69 at 32 768
        self.fused_min_rows = int(os.environ.get("HB_ACTOR_FUSED_MIN_ROWS", "16385")) if self.two_kernel else 0
        if not self.two_kernel and not self.fused:
            raise ValueError("fp16 operands: only the one-kernel actor exists (hidden 512, 51 atoms)")

    def takes_fused(self, obs):
        """True when a policy call on `obs` runs on the one-kernel form (bit rows, enough of them)."""
        return self.fused and obs.dtype == torch.int32 and obs.shape[0] >= self.fused_min_rows and self.n_actions <= 64

    def accepts(self, obs):
        """True when this object can run a policy call on `obs` at all (else the caller's library-GEMM path takes it)."""
        if self.two_kernel:
            return obs.dtype in (torch.int8, torch.int32)
        return self.takes_fused(obs)

    def state_tensors(self):
        """Every packed weight copy of every set (checkpoints: with actor_lag the acting set is state of its own)."""
        return [t for st in self.sets for t in st] + [t for st in self.fsets for t in st]

    @staticmethod
    def supports(obs_len, hidden, n_atoms, k_pad, dtype, n_actions=1):
        if dtype == torch.float16:   # the one-kernel form only, and only where it also selects the moves
            return (k_pad % 64 == 0 and n_actions <= 64 and os.environ.get("HB_ACTOR_FUSED", "1") != "0"
                    and bool(K.lib().hb_actor_fused_supported(obs_len, hidden, n_actions, n_atoms)))
        return dtype == torch.bfloat16 and k_pad % 64 == 0 and hidden % 256 == 0 and 2 <= n_atoms <= 256

    def pack(self, w1, b1, w2, b2, s=0, lazy_two_kernel=False, thin=None):
        """w1 [>= obs_len, hidden] and w2 [hidden, >= A*K] (bf16, possibly padded GEMM operands), b1 [hidden], b2 [>= A*K];
        s: the weight set written. One launch for the one-kernel form's fragment-major copies; the transposed copies of the
        two-kernel form in a second launch — or, with lazy_two_kernel (only sound when the sources still hold the SAME weights
        whenever that form is next used: the synchronous agent's persistent `eff` operands), not until a policy call takes it.
        thin = (w1t, w1t_ld, w2t, w2t_ld): the same launch also writes the k-contiguous copies hb_thin_gemm reads (the learner's
        forward), which otherwise take a launch of hb_actor_pack_weights of their own."""
        self._src[s] = (w1, b1, w2, b2)
        if self.fused:
            f = self._fset_ptrs[s]
            assert w1.dtype == self.dtype and w2.dtype == self.dtype and b1.dtype == self.dtype and b2.dtype == self.dtype
            t = (thin[0].data_ptr(), int(thin[1]), thin[2].data_ptr(), int(thin[3])) if thin is not None else (None, 0, None, 0)
            K.check(K.lib().hb_actor_fused_pack_thin(w1.data_ptr(), w1.stride(0), b1.data_ptr(), w2.data_ptr(), w2.stride(0), b2.data_ptr(),
                                                     self.obs_len, self.hidden, self.n_actions, self.n_atoms, f[0], f[1], f[2], f[3],
                                                     t[0], t[1], t[2], t[3], self._dt, K.current_stream()))
        elif thin is not None:
            raise ValueError("thin copies ride on the one-kernel actor's packer")
        if not self.two_kernel:
            return
        if lazy_two_kernel and self.fused:
            self._two_stale[s] = True
        else:
            self._pack_two(s)

    def _pack_two(self, s):
        w1, b1, w2, b2 = self._src[s]
        key = (s, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w1.stride(0), w2.stride(0))
        jobs = self._jobs.get(key)
        if jobs is None:   # (the learner packs the same persistent operands after every update: build the job table once)
            ak = self.n_actions * self.n_atoms
            w1t, b1o, w2t, b2o = self.sets[s]
            jobs = (K.HbPackJob * 2)()
            for j, (w, b, wt, bo, k_rows, n_cols, group, kp) in enumerate((
                    (w1, b1, w1t, b1o, self.obs_len, self.hidden, 0, self.k_pad),
                    (w2, b2, w2t, b2o, self.hidden, ak, self.group_actions * self.n_atoms, self.hidden))):
                jobs[j].w, jobs[j].bias, jobs[j].wt, jobs[j].bias_out = w.data_ptr(), b.data_ptr(), wt.data_ptr(), bo.data_ptr()
                jobs[j].k_rows, jobs[j].n_cols, jobs[j].w_ld, jobs[j].group_cols, jobs[j].k_pad = k_rows, n_cols, w.stride(0), group, kp
            if len(self._jobs) < 16:
                self._jobs[key] = jobs
        K.check(K.lib().hb_actor_pack_weights(jobs, 2, K.current_stream()))   # both layers in one launch
        self._two_stale[s] = False

    def q_values(self, obs, support, s=0):
        """The two GEMMs of a policy call without the selection: q [N, A] fp32 (persistent buffer). For callers that fuse the
        selection into their next kernel (HanabiEnv.step_select)."""
        c = self._q_call
        if (c is None or c[0] != obs.data_ptr() or c[1] != obs.shape[0] or self.h is None or c[4] != self.h.data_ptr()
                or c[6] != support.data_ptr()):
            # a new operand (or re-allocated scratch): validate once, then reuse the converted arguments
            n = obs.shape[0]
            packed = obs.dtype == torch.int32
            assert obs.is_contiguous() and ((packed and obs.shape[1] == (self.obs_len + 31) // 32) or
                                            (obs.dtype == torch.int8 and obs.shape[1] == self.obs_len))
            if self.q is None or self.q.shape[0] != n:
                # (h: the hidden activations between the two kernels of the two-kernel form; a one-row stand-in without it)
                self.h = torch.empty(n if self.two_kernel else 1, self.hidden, dtype=self.dtype, device=obs.device)
                self.q = torch.empty(n, self.n_actions, dtype=torch.float32, device=obs.device)
                self.tickets = torch.zeros((n + 255) // 256, dtype=torch.int32, device=obs.device)   # hb_actor_q_select
            L = K.lib()
            c = self._q_call = (obs.data_ptr(), n, L.hb_actor_hidden_packed if packed else L.hb_actor_hidden, L.hb_actor_q,
                                self.h.data_ptr(), self.q.data_ptr(), support.data_ptr())
        _, n, hidden, qfn, hp, qp, sp = c
        st = K.current_stream()
        if self.fused and obs.dtype == torch.int32 and n >= self.fused_min_rows:
            f = self._fset_ptrs[s]
            K.check(K.lib().hb_actor_fused_q_dt(c[0], n, self.obs_len, f[0], f[1], f[2], f[3], sp, self.hidden, self.n_actions,
                                                self.n_atoms, qp, self._dt, st))
            return self.q
        if not self.two_kernel:
            raise ValueError("fp16 operands: the one-kernel actor takes bit-packed observation rows only")
        if self._two_stale[s]:
            self._pack_two(s)
        w1p, b1p, w2p, b2p = self._set_ptrs[s]
        K.check(hidden(c[0], n, self.obs_len, w1p, self.k_pad, b1p, self.hidden, hp, st))
        K.check(qfn(hp, n, self.hidden, w2p, b2p, sp, self.n_actions, self.n_atoms, qp, st))
        return self.q

    def act(self, obs, legal, support, epsilon, seed, draw, first_game_id=0, s=0, actions_out=None):
        n = obs.shape[0]
        packed = obs.dtype == torch.int32
        assert obs.is_contiguous() and ((packed and obs.shape[1] == (self.obs_len + 31) // 32) or
                                        (obs.dtype == torch.int8 and obs.shape[1] == self.obs_len))
        assert legal.dtype == torch.int8 and legal.is_contiguous() and legal.shape == (n, self.n_actions)
        if self.q is None or self.q.shape[0] != n:
            self.h = torch.empty(n if self.two_kernel else 1, self.hidden, dtype=self.dtype, device=obs.device)
            self.q = torch.empty(n, self.n_actions, dtype=torch.float32, device=obs.device)
            self.tickets = torch.zeros((n + 255) // 256, dtype=torch.int32, device=obs.device)   # hb_actor_q_select
        actions = actions_out if actions_out is not None else torch.empty(n, dtype=torch.int32, device=obs.device)
        if self.fused and packed and n >= self.fused_min_rows and self.n_actions <= 64:
            # ONE launch: forward + C51 expectation + the epsilon-greedy selection on the rows each workgroup has just written
            f = self._fset_ptrs[s]
            K.check(K.lib().hb_actor_fused_act_dt(obs.data_ptr(), legal.data_ptr(), n, self.obs_len, f[0], f[1], f[2], f[3],
                                                  support.data_ptr(), self.hidden, self.n_actions, self.n_atoms, self.q.data_ptr(),
                                                  float(epsilon), int(seed), int(draw), int(first_game_id), actions.data_ptr(),
                                                  self._dt, K.current_stream()))
            return actions
        if not self.two_kernel:
            raise ValueError("fp16 operands: the one-kernel actor takes bit-packed observation rows only")
        if self._two_stale[s]:
            self._pack_two(s)
        w1p, b1p, w2p, b2p = self._set_ptrs[s]
        K.check(K.lib().hb_actor_act(obs.data_ptr(), 1 if packed else 0, legal.data_ptr(), n, self.obs_len, w1p, self.k_pad, b1p,
                                     self.hidden, self.h.data_ptr(), w2p, b2p, support.data_ptr(), self.n_actions, self.n_atoms,
                                     self.q.data_ptr(), float(epsilon), int(seed), int(draw), int(first_game_id),
                                     actions.data_ptr(), self.tickets.data_ptr() if self.fuse_select else None,
                                     K.current_stream()))   # hidden GEMM; q GEMM + C51 expectation + selection
        return actions
