"""Thin wrappers over the actor and replay-insert kernels (hb_policy_act, hb_actor_*, hb_replay_insert)."""
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


class ActorMFMA:
    """The actor's forward pass on the hand-written MFMA kernels (csrc/actor.hip): packed (transposed) copies of
    the effective weights plus the scratch buffers; `pack` after every weight change, `act` per step."""

    def __init__(self, obs_len, hidden, n_actions, n_atoms, k_pad, device):
        assert k_pad % 64 == 0 and k_pad >= obs_len and hidden % 256 == 0 and 2 <= n_atoms <= 256
        self.obs_len, self.hidden, self.n_actions, self.n_atoms, self.k_pad = obs_len, hidden, n_actions, n_atoms, k_pad
        self.group_actions = 256 // n_atoms
        groups = -(-n_actions // self.group_actions)
        bf = dict(dtype=torch.bfloat16, device=device)
        self.w1t = torch.zeros(hidden, k_pad, **bf)
        self.b1 = torch.zeros(hidden, dtype=torch.float32, device=device)
        self.w2t = torch.zeros(groups * 256, hidden, **bf)
        self.b2 = torch.zeros(groups * 256, dtype=torch.float32, device=device)
        self.h = self.q = self.actions = None

    @staticmethod
    def supports(obs_len, hidden, n_atoms, k_pad, dtype):
        return dtype == torch.bfloat16 and k_pad % 64 == 0 and hidden % 256 == 0 and 2 <= n_atoms <= 256

    def pack(self, w1, b1, w2, b2):
        """w1 [>= obs_len, hidden] and w2 [hidden, >= A*K] (bf16, possibly padded GEMM operands), b1 [hidden], b2 [>= A*K]."""
        ak = self.n_actions * self.n_atoms
        jobs = (K.HbPackJob * 2)()
        for j, (w, b, wt, bo, k_rows, n_cols, group, kp) in enumerate((
                (w1, b1, self.w1t, self.b1, self.obs_len, self.hidden, 0, self.k_pad),
                (w2, b2, self.w2t, self.b2, self.hidden, ak, self.group_actions * self.n_atoms, self.hidden))):
            jobs[j].w, jobs[j].bias, jobs[j].wt, jobs[j].bias_out = w.data_ptr(), b.data_ptr(), wt.data_ptr(), bo.data_ptr()
            jobs[j].k_rows, jobs[j].n_cols, jobs[j].w_ld, jobs[j].group_cols, jobs[j].k_pad = k_rows, n_cols, w.stride(0), group, kp
        K.check(K.lib().hb_actor_pack_weights(jobs, 2, K.current_stream()))   # both layers in one launch

    def act(self, obs, legal, support, epsilon, seed, draw, first_game_id=0):
        n = obs.shape[0]
        packed = obs.dtype == torch.int32
        assert obs.is_contiguous() and ((packed and obs.shape[1] == (self.obs_len + 31) // 32) or
                                        (obs.dtype == torch.int8 and obs.shape[1] == self.obs_len))
        assert legal.dtype == torch.int8 and legal.is_contiguous() and legal.shape == (n, self.n_actions)
        if self.h is None or self.h.shape[0] != n:
            self.h = torch.empty(n, self.hidden, dtype=torch.bfloat16, device=obs.device)
            self.q = torch.empty(n, self.n_actions, dtype=torch.float32, device=obs.device)
        actions = torch.empty(n, dtype=torch.int32, device=obs.device)
        L, s = K.lib(), K.current_stream()
        hidden = L.hb_actor_hidden_packed if packed else L.hb_actor_hidden   # bit rows are unpacked while staged into LDS
        K.check(hidden(K.dptr(obs), n, self.obs_len, K.dptr(self.w1t), self.k_pad, K.dptr(self.b1), self.hidden, K.dptr(self.h), s))
        K.check(L.hb_actor_q(K.dptr(self.h), n, self.hidden, K.dptr(self.w2t), K.dptr(self.b2), K.dptr(support), self.n_actions,
                             self.n_atoms, K.dptr(self.q), s))
        K.check(L.hb_policy_select(K.dptr(self.q), K.dptr(legal), n, self.n_actions, float(epsilon), int(seed), int(draw),
                                   int(first_game_id), K.dptr(actions), s))
        return actions
