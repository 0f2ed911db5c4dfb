"""GPU: hb_actor_fused_q / hb_actor_fused_pack (csrc/actor_fused.hip) — the whole policy forward in one kernel.

Checked against (a) an fp32 torch forward built from the SAME bf16 weights with the hidden activations rounded to bf16 (what the
kernel feeds its second layer), i.e. DQNPolicy's arithmetic (hanabi_agents/rlax_dqn/rlax_rainbow.py:113-122 over
noisy_mlp.py:176-185) at the kernel's precision contract; (b) the two-kernel form it replaces (hb_actor_hidden_packed +
hb_actor_q); (c) a Python restatement of the packer's physical column order. Parity vs the running reference is unpinned (jax /
haiku absent): the fp32 torch forward is the stated oracle for this kernel."""
import pytest

pytestmark = pytest.mark.gpu

# (obs_len, n_actions): Hanabi-Full 2..5 players, Hanabi-Small 2 players
SHAPES = {"full2": (658, 20), "full3": (783, 30), "full4": (908, 38), "full5": (1280, 48), "small2": (171, 11), "one": (40, 1)}


def _setup(obs_len, n_act, n, seed, scale2=0.2, dtype="bfloat16"):
    import torch

    from hanabi_agents.rlax_dqn import bitpack
    from hanabi_hip.ops import ActorMFMA

    dt = getattr(torch, dtype)
    H, K = 512, 51
    kp, np_ = (obs_len + 63) // 64 * 64, (n_act * K + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.35).to(torch.int8)
    w1 = torch.zeros(kp, H, device="cuda", dtype=dt)
    w1[:obs_len] = (torch.randn(obs_len, H, device="cuda", generator=g) * 0.04).to(dt)
    b1 = (torch.randn(H, device="cuda", generator=g) * 0.05).to(dt)
    w2 = torch.zeros(H, np_, device="cuda", dtype=dt)
    w2[:, :n_act * K] = (torch.randn(H, n_act * K, device="cuda", generator=g) * scale2).to(dt)
    b2 = torch.zeros(np_, device="cuda", dtype=dt)
    b2[:n_act * K] = (torch.randn(n_act * K, device="cuda", generator=g) * 0.5).to(dt)
    support = torch.linspace(-25, 25, K, device="cuda")
    act = ActorMFMA(obs_len, H, n_act, K, kp, "cuda", dtype=dt)
    assert act.fused, "the fused kernel must cover this shape"
    act.fused_min_rows = 0
    act.pack(w1, b1, w2, b2)
    return act, obs, bitpack.pack(obs), (w1, b1, w2, b2), support


def _reference_q(obs, w, support, obs_len, n_act):
    import torch

    w1, b1, w2, b2 = w
    h = torch.relu(obs.float() @ w1[:obs_len].float() + b1.float()).to(w1.dtype).float()
    lg = (h @ w2[:, :n_act * 51].float() + b2[:n_act * 51].float()).view(obs.shape[0], n_act, 51)
    return (torch.softmax(lg, -1) * support).sum(-1) / 51


@pytest.mark.parametrize("shape,n,dtype", [("full2", 32768, "bfloat16"), ("full2", 1000, "bfloat16"), ("full3", 777, "bfloat16"),
                                           ("full4", 515, "bfloat16"), ("full5", 4096 + 37, "bfloat16"), ("small2", 130, "bfloat16"),
                                           ("one", 1, "bfloat16"), ("full2", 127, "bfloat16"), ("full2", 129, "bfloat16"),
                                           ("full2", 32768, "float16"), ("full5", 4096 + 37, "float16"), ("small2", 130, "float16"),
                                           ("full2", 129, "float16")])
def test_fused_q_equals_fp32_forward(shape, n, dtype):
    """q from the fused kernel vs the fp32 forward of the same bf16 (fp16: hb_actor_fused_*_dt, the reference's own network dtype)
    weights. The kernel's logits are fp32 accumulators (never rounded); what is left is the summation order inside a row of H,
    which flips a rounding of H now and then: max |dq| 1e-3 on |q| <= 0.49 (measured 7e-7 .. 3e-4), and the same arg-max
    wherever the top-2 gap exceeds 2e-3."""
    import torch

    obs_len, n_act = SHAPES[shape]
    act, obs, bits, w, support = _setup(obs_len, n_act, n, seed=n + n_act, dtype=dtype)
    q = act.q_values(bits, support).clone()
    torch.cuda.synchronize()
    ref = _reference_q(obs, w, support, obs_len, n_act)
    assert torch.isfinite(q).all()
    assert float((q - ref).abs().max()) <= 1e-3, float((q - ref).abs().max())
    assert float((q - ref).abs().mean()) <= 2e-5
    if n_act > 1:
        top2 = ref.topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2e-3
        assert torch.equal(q.argmax(-1)[clear], ref.argmax(-1)[clear])
    # rows past a ragged end are neither read nor written
    if n > 200:
        act.h = act._q_call = None
        q2 = act.q_values(bits[:n - 77].contiguous(), support)
        assert torch.equal(q2, q[:n - 77])


def test_fused_equals_two_kernel_form_and_selects_legal_moves():
    """Same weights, bit rows in: hb_actor_fused_q against hb_actor_hidden_packed + hb_actor_q (whose logits pass through fp16 in
    LDS): q within 2e-3; act() through either form returns legal moves and, with epsilon = 0, the arg-max of its own q."""
    import torch

    obs_len, n_act = SHAPES["full2"]
    n = 3000
    act, obs, bits, w, support = _setup(obs_len, n_act, n, seed=5)
    legal = (torch.rand(n, n_act, device="cuda") < 0.5).to(torch.int8)
    legal[:, 3] = 1
    a_f = act.act(bits, legal, support, 0.0, 77, 9, 0).clone()
    q_f = act.q.clone()
    act.fused_min_rows = 1 << 30
    a_2 = act.act(bits, legal, support, 0.0, 77, 9, 0).clone()
    q_2 = act.q.clone()
    assert float((q_f - q_2).abs().max()) <= 2e-3
    for a, q in ((a_f, q_f), (a_2, q_2)):
        assert bool(legal.gather(1, a.long()[:, None]).all())
        masked = torch.where(legal.bool(), q, torch.full_like(q, float("-inf")))
        assert torch.equal(masked.gather(1, a.long()[:, None])[:, 0], masked.max(1).values)


@pytest.mark.parametrize("shape,n,dtype", [("full2", 32768, "bfloat16"), ("full5", 900, "bfloat16"), ("small2", 130, "bfloat16"),
                                           ("full2", 4099, "float16")])
def test_fused_act_selects_like_hb_policy_select(shape, n, dtype):
    """hb_actor_fused_act (selection inside the kernel, on the q rows the workgroup has just written) == hb_actor_fused_q followed
    by hb_policy_select: identical q, identical actions (same Philox draws, same tie rule), greedy and with exploration."""
    import torch

    from hanabi_hip import _capi as K

    obs_len, n_act = SHAPES[shape]
    act, obs, bits, w, support = _setup(obs_len, n_act, n, seed=n, dtype=dtype)
    g = torch.Generator(device="cuda").manual_seed(n + 1)
    for rep, eps in enumerate((0.0, 0.3, 1.0)):
        legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.5).to(torch.int8)
        legal[:, rep % n_act] = 1
        a_f = act.act(bits, legal, support, eps, 77, 100 + rep, 4096).clone()
        q_f = act.q.clone()
        q_s = act.q_values(bits, support).clone()
        a_s = torch.empty(n, dtype=torch.int32, device="cuda")
        K.check(K.lib().hb_policy_select(q_s.data_ptr(), legal.data_ptr(), n, n_act, eps, 77, 100 + rep, 4096, a_s.data_ptr(),
                                         K.current_stream()))
        assert torch.equal(q_f, q_s)
        assert torch.equal(a_f, a_s), f"eps {eps}: {(a_f != a_s).sum().item()} actions differ"
        assert bool(legal.gather(1, a_f.long()[:, None]).all())


def _logit_of_column(col, n_actions, n_pass):
    """Python restatement of the packer's physical column order (csrc/actor_fused.hip, fused_logit)."""
    slot, c = col >> 6, col & 63
    n, q, j = c >> 4, (c >> 2) & 3, c & 3
    r16, full_cap = 4 * n + j, 8 * n_pass
    if r16 < 12 or (r16 == 12 and q < 3):
        action, atom = slot, (4 * r16 + q if r16 < 12 else 48 + q)
    else:
        p, w = slot >> 3, slot & 7
        f = w & 3
        action = full_cap + 2 * p + (w >> 2)
        atom = 13 * f + 12 if r16 == 12 else 13 * f + 4 * (r16 - 13) + q
    return action * 51 + atom if (action < n_actions and atom < 51) else -1


@pytest.mark.parametrize("shape,dtype", [("full2", "bfloat16"), ("full3", "bfloat16"), ("full5", "bfloat16"), ("small2", "bfloat16"),
                                         ("full2", "float16")])
def test_pack_is_a_permutation_of_the_logit_columns(shape, dtype):
    """Every logit column of W2 / b2 lands in exactly one physical column of the fragment-major copy, unused columns are zero
    weights with a -1e30 bias, and W1 is the k-major fragment image of the (zero-padded) first layer."""
    import numpy as np
    import torch

    obs_len, n_act = SHAPES[shape]
    act, _, _, (w1, b1, w2, b2), _ = _setup(obs_len, n_act, 8, seed=3, dtype=dtype)
    n_pass = (n_act + 9) // 10
    w1f, b1f, w2f, b2f = (t.cpu() for t in act.fsets[0])
    cols = np.array([_logit_of_column(c, n_act, n_pass) for c in range(512 * n_pass)])
    used = cols[cols >= 0]
    assert sorted(used.tolist()) == list(range(n_act * 51))
    b2c = b2.float().cpu()
    assert torch.equal(b2f[torch.from_numpy(cols >= 0)], b2c[torch.from_numpy(used)])
    assert bool((b2f[torch.from_numpy(cols < 0)] == -1e30).all())
    # W2f [pass][s 16][n-tile 32][lane 64][8]: element i of lane (c, q) = W2[32 s + 8 q + i][logit of column 16 (32 pass + nt) + c]
    w2v = w2f.view(n_pass, 16, 32, 4, 16, 8).float()        # [pass, s, nt, q, c, i]
    w2c = w2.float().cpu()
    for p, s, nt, c in ((0, 0, 0, 0), (0, 5, 3, 7), (n_pass - 1, 15, 31, 15), (n_pass - 1, 9, 12, 3), (0, 2, 17, 12)):
        lg = cols[16 * (32 * p + nt) + c]
        for q in range(4):
            want = w2c[32 * s + 8 * q:32 * s + 8 * q + 8, lg] if lg >= 0 else torch.zeros(8)
            assert torch.equal(w2v[p, s, nt, q, c], want)
    s1 = 2 * ((obs_len + 63) // 64)
    w1v = w1f.view(s1, 32, 4, 16, 8).float()                # [s, nt, q, c, i]
    w1c = torch.zeros(s1 * 32, 512)
    w1c[:obs_len] = w1[:obs_len].float().cpu()
    for s, nt, c in ((0, 0, 0), (s1 - 1, 31, 15), (3, 9, 4)):
        for q in range(4):
            assert torch.equal(w1v[s, nt, q, c], w1c[32 * s + 8 * q:32 * s + 8 * q + 8, 16 * nt + c])
    assert torch.equal(b1f, b1.float().cpu())


def test_unsupported_shapes_are_refused():
    from hanabi_hip import _capi as K

    L = K.lib()
    assert L.hb_actor_fused_supported(658, 512, 20, 51) == 1
    assert L.hb_actor_fused_supported(658, 256, 20, 51) == 0       # hidden != 512
    assert L.hb_actor_fused_supported(658, 512, 20, 21) == 0       # atoms != 51
    with pytest.raises(K.HbError):
        K.check(L.hb_actor_fused_sizes(658, 256, 20, 51, None, None, None))
