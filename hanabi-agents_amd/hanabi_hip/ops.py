"""Thin wrappers over the fused actor-tail and replay-insert kernels (hb_policy_act, hb_replay_insert)."""
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
    n, obs_len = obs.shape
    for t, dt in ((last_obs, torch.int8), (obs, torch.int8), (legal, torch.int8), (actions, torch.int32),
                  (rewards, torch.float32), (step_types, torch.int8)):
        assert t.is_cuda and t.is_contiguous() and t.dtype == dt, (t.dtype, dt)
    K.check(K.lib().hb_replay_insert(K.dptr(last_obs), K.dptr(obs), K.dptr(legal), K.dptr(actions), K.dptr(rewards),
                                     K.dptr(step_types), K.dptr(ring._obs_tm1_buf), K.dptr(ring._obs_t_buf),
                                     K.dptr(ring._act_tm1_buf), K.dptr(ring._lms_t_buf), K.dptr(ring._rew_t_buf),
                                     K.dptr(ring._terminal_t_buf), n, obs_len, legal.shape[1], ring.capacity, int(start),
                                     K.current_stream()))
