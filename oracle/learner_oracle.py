"""numpy float64 restatement of the learner arithmetic. TEST INFRASTRUCTURE ONLY.

Written independently of hanabi_agents/rlax_dqn/learning.py (classic floor/ceil C51 projection instead of
the dense clip formulation) so that agreement between the two means something.
Sources: hanabi_agents/rlax_dqn/rlax_rainbow.py:172-200 (loss), hanabi_agents/rlax_dqn/noisy_mlp.py:61-91,176-185
(network), SURVEY.md Appendix B (rlax.categorical_double_q_learning, categorical_l2_project, optix.adam),
hanabi_agents/rainbow/rainbow_agent.py:252-404 (Dopamine's project_distribution, the worked example at :262-266).
Parity status: the reference learner cannot be imported here (jax/haiku/rlax absent) => "parity unpinned"
against a reference run; pinned by the Dopamine example and hand KATs in tests/test_learner.py.
"""
import numpy as np


def softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def log_softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    return x - np.log(np.exp(x).sum(axis=-1, keepdims=True))


def project_uniform(z_p, probs, vmin, vmax, k):
    """Classic C51 projection onto a uniform support of k atoms: mass of each source atom is split
    between its two neighbouring target atoms (floor/ceil form)."""
    delta = (vmax - vmin) / (k - 1)
    out = np.zeros(k)
    for zp, p in zip(z_p, probs):
        b = (min(max(zp, vmin), vmax) - vmin) / delta
        lo, hi = int(np.floor(b)), int(np.ceil(b))
        if lo == hi:
            out[lo] += p
        else:
            out[lo] += p * (hi - b)
            out[hi] += p * (b - lo)
    return out


def project_general(z_p, probs, z_q):
    """Projection onto an arbitrary sorted support (Dopamine project_distribution semantics)."""
    z_q = np.asarray(z_q, float)
    out = np.zeros(len(z_q))
    for zp, p in zip(z_p, probs):
        zp = min(max(zp, z_q[0]), z_q[-1])
        j = np.searchsorted(z_q, zp, side="right") - 1
        if j >= len(z_q) - 1:
            out[-1] += p
        else:
            w = (zp - z_q[j]) / (z_q[j + 1] - z_q[j])
            out[j] += p * (1 - w)
            out[j + 1] += p * w
    return out


def noisy_mlp_forward(x, layers):
    """layers: list of dicts w,b,w_mu,b_mu,w_sigma,b_sigma,eps_w,eps_b (noisy_mlp.py:61-91,176-185)."""
    out = np.asarray(x, float)
    for i, l in enumerate(layers):
        plain = out @ l["w"] + l["b"]
        noisy = out @ (l["w_mu"] + l["w_sigma"] * l["eps_w"]) + (l["b_mu"] + l["b_sigma"] * l["eps_b"])
        out = plain + noisy
        if i < len(layers) - 1:
            out = np.maximum(out, 0.0)
    return out


def c51_double_q_td(logits_tm1, a_tm1, r_t, discount, support, logits_t, logits_sel, terminal=None):
    """Per-sample cross-entropy 'TD' of rlax_rainbow.py:172-185; logits [B, A, K]."""
    b, a, k = logits_tm1.shape
    td = np.zeros(b)
    for i in range(b):
        q_sel = (softmax(logits_sel[i]) * support[None]).mean(-1)   # mean, not sum (C-3)
        a_star = int(np.argmax(q_sel))
        p = softmax(logits_t[i, a_star])
        g = discount * (1.0 - terminal[i]) if terminal is not None else discount
        target = project_uniform(r_t[i] + g * support, p, support[0], support[-1], k)
        td[i] = -(target * log_softmax(logits_tm1[i, a_tm1[i]])).sum()
    return td


def is_weights(prios, beta):
    w = (1.0 / np.asarray(prios, float)) ** beta
    return w / w.max()


def adam_step(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=3.125e-5):
    """optix.adam: eps outside the square root; t counts from 1."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat, vhat = m / (1 - b1 ** t), v / (1 - b2 ** t)
    return p - lr * mhat / (np.sqrt(vhat) + eps), m, v
