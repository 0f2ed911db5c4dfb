"""Loss arithmetic of the learner, in PyTorch (device-agnostic tensor math; runs on the MI355X).

Restates what `DQNLearning.update_q` computes (hanabi_agents/rlax_dqn/rlax_rainbow.py:152-217) and
the third-party rlax pieces it calls (SURVEY.md Appendix B — rlax is not in the reference tree):

  categorical_l2_project            rlax.categorical_l2_project (Cramér projection onto a support)
  categorical_double_q_td           rlax.categorical_double_q_learning, vmapped, with atoms[0] as both
                                    supports and a scalar discount (rlax_rainbow.py:172-185)
  is_weights                        (1/P)^beta / max (rlax_rainbow.py:188-189)
  double_q_td / scalar loss         rlax.double_q_learning + l2 + clip_gradient of the older agent
                                    (hanabi_agents/rlax_dqn/rlax_dqn.py:170-205), BASELINE config 2
"""
import torch
import torch.nn.functional as F


def categorical_l2_project(z_p: torch.Tensor, probs: torch.Tensor, z_q: torch.Tensor) -> torch.Tensor:
    """Project the distribution (z_p, probs) onto the support z_q.

    z_p [B, Kp] atom locations, probs [B, Kp], z_q [Kq] (sorted). Returns [B, Kq].
    Each target atom i receives sum_j probs_j * clip(1 - |clip(z_p_j) - z_q_i| / d, 0, 1) with d the
    spacing of z_q on the side z_p_j lies on; equals Dopamine's
    project_distribution (hanabi_agents/rainbow/rainbow_agent.py:252-404) on a uniform support.
    """
    kq = z_q.shape[0]
    d_pos = torch.cat([z_q, z_q[:1]])[1:]      # distance to the next atom (wraps, unused at the edge)
    d_neg = torch.cat([z_q[-1:], z_q])[:-1]    # previous atom
    z_p = torch.clamp(z_p, z_q[0], z_q[-1])[:, None, :]          # [B, 1, Kp]
    zq = z_q.view(1, kq, 1)
    d_pos = (d_pos - z_q).view(1, kq, 1)
    d_neg = (z_q - d_neg).view(1, kq, 1)
    delta = z_p - zq                                             # [B, Kq, Kp]
    sign = (delta >= 0).to(delta.dtype)
    delta_hat = sign * delta / d_pos - (1.0 - sign) * delta / d_neg
    return torch.sum(torch.clamp(1.0 - delta_hat, 0.0, 1.0) * probs[:, None, :], dim=-1)


def expected_q(logits: torch.Tensor, atoms: torch.Tensor) -> torch.Tensor:
    """q = mean(softmax(logits) * atoms, -1) — note MEAN, not sum (rlax_rainbow.py:117-118,176; App. C-3)."""
    return torch.mean(F.softmax(logits, dim=-1) * atoms, dim=-1)


def categorical_double_q_td(logits_tm1, a_tm1, r_t, discount, atoms, logits_t, logits_sel, terminal_t=None):
    """Per-sample C51 double-Q cross-entropy 'TD error' (rlax_rainbow.py:172-185).

    logits_* [B, A, K]; a_tm1 [B] int64; r_t [B]; atoms [A, K] (rows identical); discount scalar (the reference)
    or a [B] tensor (gamma^m per sample, n-step).
    terminal_t: None reproduces the reference (bootstraps through episode ends, App. C-5);
    a [B] 0/1 tensor multiplies the discount by (1 - terminal).
    """
    b = logits_tm1.shape[0]
    ar = torch.arange(b, device=logits_tm1.device)
    support = atoms[0]
    q_sel = expected_q(logits_sel, atoms)                        # no legal-move mask, as in the reference
    a_star = torch.argmax(q_sel, dim=-1)
    p_target = F.softmax(logits_t[ar, a_star], dim=-1)           # [B, K]
    if isinstance(discount, torch.Tensor):  # per-sample gamma^m of n-step transitions
        disc = discount.to(r_t.dtype)
    else:
        disc = torch.full((b,), float(discount), dtype=r_t.dtype, device=r_t.device)  # (a fill kernel: graph-capturable)
    if terminal_t is not None:
        disc = disc * (1.0 - terminal_t.to(r_t.dtype))
    target_z = r_t[:, None] + disc[:, None] * support[None, :]
    target = categorical_l2_project(target_z, p_target, support).detach()
    logp = F.log_softmax(logits_tm1[ar, a_tm1], dim=-1)
    return -torch.sum(target * logp, dim=-1)


def is_weights(prios: torch.Tensor, beta) -> torch.Tensor:
    """(1/P)^beta normalised by its max, float32 (rlax_rainbow.py:188-189; App. C-6). beta: float or 0-d tensor."""
    w = (1.0 / prios).to(torch.float32) ** beta
    return w / torch.max(w)


def double_q_td(q_tm1, a_tm1, r_t, discount, q_t_value, q_t_selector, terminal_t=None):
    """Scalar double-Q TD: r + g * q_t_value[argmax q_t_selector] - q_tm1[a] (rlax_dqn.py:170-181).
    The older agent zeroes q_t at terminal states (rlax_dqn.py:178) -> pass terminal_t."""
    b = q_tm1.shape[0]
    ar = torch.arange(b, device=q_tm1.device)
    if terminal_t is not None:
        q_t_value = torch.where(terminal_t.bool()[:, None], torch.zeros_like(q_t_value), q_t_value)
    target = r_t + discount * q_t_value[ar, torch.argmax(q_t_selector, dim=-1)]
    return target.detach() - q_tm1[ar, a_tm1]


class _ClipGradient(torch.autograd.Function):
    """rlax.clip_gradient: identity forward, cotangent clipped to [lo, hi] (rlax_dqn.py:203)."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.lo, ctx.hi = lo, hi
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return torch.clamp(g, ctx.lo, ctx.hi), None, None


def clip_gradient(x, lo=-1.0, hi=1.0):
    return _ClipGradient.apply(x, lo, hi)
