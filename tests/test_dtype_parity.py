"""GPU: the arithmetic bench.py times (bf16 GEMM operands; fp16 = the reference's own network dtype,
hanabi_agents/rlax_dqn/rlax_rainbow.py:250-251) against the fp32 PyTorch-autograd path, with the STATED tolerances of
hanabi_agents/rlax_dqn/tolerance.py (include/hanabi_hip.h).

Learner: FusedLearner (hb_per_sample_gather / hb_replay_gather -> hb_thin_gemm forward with fp32 logits (bf16; library GEMMs for
fp16) -> hb_c51_loss_sparse -> hb_c51_backward -> dW1 GEMM -> hb_noisy_adam_multi) at compute_dtype bf16 / fp16 vs
DQNLearning.loss + torch.optim.Adam in fp32 (loss of rlax_rainbow.py:172-200) on the same batch, weights, sampling
probabilities: per-sample td, IS weights, loss, the four merged gradients after one step, the weights after five. Actor (bf16):
hb_actor_fused_q on bit rows (the benched form) and hb_actor_hidden + hb_actor_q on int8 rows; fp16: the library-GEMM fallback;
all vs DQNPolicy.q_values in fp32 at the 2-player and 5-player shapes, arg-max agreement wherever the fp32 top-2 gap is clear.
Measured errors are written to gpurun_out/dtype_parity.json next to the bounds."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {2: (658, 20), 5: (1280, 48)}


def _record(key, values):
    path = os.path.join(ROOT, "gpurun_out", "dtype_parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        data = json.load(open(path))
    except Exception:
        data = {}
    data[key] = values
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _make_agents(dtype, players, n, mask=True, n_step=1):
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    obs_len, n_act = SHAPES[players]
    base = RlaxRainbowParams(use_priority=False, train_batch_size=n, experience_buffer_size=n * n_step, target_update_period=3,
                             mask_terminal=mask, n_step=n_step)
    fused = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), base._replace(compute_dtype=dtype), device="cuda",
                     use_graphs=False, use_fused_learner=True)
    ref = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), base._replace(compute_dtype="float32"), device="cuda",
                   use_graphs=False, use_fused_learner=False)
    g = torch.Generator(device="cuda").manual_seed(11 + players)
    # weights a few hundred updates into training look like: non-zero biases and bias noise, target != online
    pert = [torch.randn(p.shape, device="cuda", generator=g) for p in fused.online.parameters()]
    for a in (fused, ref):
        with torch.no_grad():
            for layer in a.online.layers:
                layer.b_sigma.fill_(0.05)
                layer.b.fill_(0.02)
            a.target.load_state_dict(a.online.state_dict())
            for p, d in zip(a.target.parameters(), pert):
                p.add_(d * 0.01)
    return fused, ref, g, obs_len, n_act


def _fill(agents, g, n, obs_len, n_act, n_step=1):
    import torch

    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    st = torch.randint(1, 3, (n,), device="cuda", generator=g).to(torch.int8)
    for a in agents:
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, st)
        for extra in range(n_step - 1):
            a.add_experience_dense((None, (o1 if extra % 2 else o2, legal)), act, rew + extra, st)


@pytest.mark.parametrize("players", [2, 5])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_fused_learner_reduced_precision_vs_fp32_autograd(dtype, players):
    import torch

    from hanabi_agents.rlax_dqn import DQNLearning
    from hanabi_agents.rlax_dqn.tolerance import TOLERANCE

    tol = TOLERANCE[dtype]
    n = 256
    fused, ref, g, obs_len, n_act = _make_agents(dtype, players, n)
    _fill((fused, ref), g, n, obs_len, n_act)
    idx = torch.randperm(n, device="cuda", generator=g)
    pri = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.05) / n
    fused._sample_indices = lambda: (idx, pri)
    ref._sample_indices = lambda: (idx, pri)
    assert fused._fused_learner() is not None and ref._fused_learner() is None
    fl = fused._fl
    AK = n_act * 51

    # ---- fp32 reference quantities of the first update, straight from the loss of rlax_rainbow.py:172-200 --------------
    tr = ref.experience.gather_dev(idx)
    trf = tr._replace(observation_tm1=tr.observation_tm1.float(), observation_t=tr.observation_t.float())
    plist = list(ref.online.parameters())
    loss32, td32 = DQNLearning.loss(ref.online, ref.target, ref.atoms, trf, 0.99, pri, 0.4, True)
    grads = dict(zip([nm for nm, _ in ref.online.named_parameters()], torch.autograd.grad(loss32, plist)))
    w_before = torch.cat([p.detach().reshape(-1) for p in plist]).clone()
    from hanabi_agents.rlax_dqn import learning as L

    w32 = L.is_weights(pri, 0.4)
    # double-Q selection a* = argmax_a q_sel(obs_t) (rlax_rainbow.py:175-176): where two actions are nearly tied in fp32,
    # rounding may legitimately select the other one, which changes that sample's target distribution altogether
    with torch.no_grad():
        q_sel = L.expected_q(ref.online(trf.observation_t).view(n, n_act, 51), ref.atoms)
    top2 = q_sel.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > tol["selection_gap"]

    # ---- one update on both ------------------------------------------------------------------------------------------
    fused.update()
    ref.update()
    td_lp = fl.td.clone()
    err_td = (td_lp - td32).abs()
    bound = tol["td_abs"] + tol["td_rel"] * td32.abs()
    measured = {"td_max_abs_err": float(err_td.max()), "td_mean": float(td32.mean()),
                "loss_rel_err": float((fused.last_loss - loss32.detach()).abs() / loss32.detach().abs()),
                "is_weight_max_abs_err": float((fl.w_is - w32).abs().max())}
    got = {"layers.0.w": fl._gw1_out[:obs_len].float(), "layers.0.b": fl.g_b1, "layers.1.w": fl._gw2_out[:, :AK].float(),
           "layers.1.b": fl._gb2_pad[:AK]}
    for name, gl in got.items():
        measured["grad_rel_l2_" + name] = _rel_l2(gl, grads[name])
        # the merged gradient is routed (g, g, g * eps) to (w, w_mu, w_sigma): exact by construction, checked in fp32 elsewhere
    measured["td_worst_bound_ratio_clear_selection"] = float((err_td / bound)[clear].max())
    measured["td_max_abs_err_clear_selection"] = float(err_td[clear].max())
    measured["samples_with_clear_selection"] = float(clear.float().mean())
    measured["td_rel_l2"] = _rel_l2(td_lp, td32)
    _record(f"learner_{dtype}_{players}p", {"measured": measured, "tolerance": tol})
    assert measured["samples_with_clear_selection"] > 0.8, measured
    assert bool((err_td <= bound)[clear].all()), measured
    assert measured["loss_rel_err"] <= tol["loss_rel"], measured
    assert measured["is_weight_max_abs_err"] <= tol["is_weight_abs"], measured
    for name in got:
        assert measured["grad_rel_l2_" + name] <= tol["grad_rel_l2"], measured

    # ---- four more updates: weights --------------------------------------------------------------------------------------
    for _ in range(4):
        fused.update()
        ref.update()
    wa = torch.cat([p.detach().reshape(-1) for p in fused.online.parameters()])
    wb = torch.cat([p.detach().reshape(-1) for p in ref.online.parameters()])
    measured["weights5_rel_l2_of_delta"] = _rel_l2(wa - w_before, wb - w_before)
    measured["weights5_max_abs"] = float((wa - wb).abs().max())
    measured["weights5_rel_l2"] = _rel_l2(wa, wb)
    _record(f"learner_{dtype}_{players}p", {"measured": measured, "tolerance": tol})
    assert measured["weights5_rel_l2_of_delta"] <= tol["weights_after_5_steps_rel_l2_of_delta"], measured
    assert measured["weights5_max_abs"] <= tol["weights_after_5_steps_max_abs"], measured
    assert torch.isfinite(wa).all()


@pytest.mark.parametrize("players", [2, 5])
@pytest.mark.parametrize("dtype,form", [("bfloat16", "bits"), ("bfloat16", "int8"), ("float16", "bits"), ("float16", "int8")])
def test_actor_q_values_reduced_precision_vs_fp32_policy(dtype, form, players):
    """Bit rows: hb_actor_fused_act[_dt] (csrc/actor_fused.hip, the benched form; bf16 and — the reference's own network dtype —
    fp16 operands). bf16, int8 rows: hb_actor_hidden + hb_actor_q (csrc/actor.hip). fp16, int8 rows: hb_obs_cast + library GEMMs +
    hb_policy_act (the fallback `_act_fused` takes for what the MFMA kernels do not cover). All against DQNPolicy.q_values in fp32."""
    import torch

    from hanabi_agents.rlax_dqn import DQNPolicy, bitpack
    from hanabi_agents.rlax_dqn.tolerance import TOLERANCE
    from hanabi_hip import ops

    tol = TOLERANCE[dtype]
    n = 4096 + 37      # ragged: not a multiple of the 256-row tile
    fused, ref, g, obs_len, n_act = _make_agents(dtype, players, 64)
    obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8)
    legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.5).to(torch.int8)
    legal[:, 2] = 1
    with torch.no_grad():   # spread the q values (a fresh net's are nearly equal): scale the output layer
        for a in (fused, ref):
            for name in ("w", "w_mu", "w_sigma"):
                getattr(a.online.layers[1], name).mul_(4.0)
    fl = fused._fused_learner()
    fl.refresh_effective()
    q32 = DQNPolicy.q_values(ref.online, ref.atoms, obs.float(), legal)
    fused._draws = 3
    if form == "bits":
        assert fl.actor is not None and fl.actor.fused, "the fused actor kernel must cover the benched shape"
        fl.actor.fused_min_rows = 0
    arg = bitpack.pack(obs) if form == "bits" else obs
    act = fused._act_fused(arg, legal, 0.0)
    assert fl.actor is not None, "the MFMA actor must cover the benched shape"
    if dtype == "bfloat16" or form == "bits":
        assert fl.actor.accepts(arg) and fl.actor.two_kernel == (dtype == "bfloat16")
        q_lp = fl.actor.q.clone()
    else:
        assert not fl.actor.accepts(arg)
        (w1, b1), (w2, b2) = fl.eff
        x = torch.zeros(n, fl.Kp, dtype=fl.cd, device="cuda")
        ops.obs_cast(obs, fl.cd, out=x)
        logits = torch.addmm(b2, torch._addmm_activation(b1, x, w1, use_gelu=False), w2)
        q_lp = torch.empty(n, n_act, device="cuda")
        act2 = ops.policy_act(logits, legal, fused.atoms[0].contiguous(), 0.0, fused.params.seed + 0x9E3779B9, 3, 0, q_out=q_lp)
        assert torch.equal(act, act2)
    lm = legal.bool()
    q32f = torch.where(lm, q32, torch.zeros_like(q32))
    err = (q_lp - q32f).abs()[lm]
    top2 = q32.topk(2, dim=1).values
    gap = top2[:, 0] - top2[:, 1]
    clear = gap > tol["argmax_gap"]
    agree = act.long() == q32.argmax(1)
    measured = {"q_max_abs_err": float(err.max()), "q_mean_abs_err": float(err.mean()), "q_abs_mean": float(q32f[lm].abs().mean()),
                "rows_with_clear_gap": float(clear.float().mean()), "argmax_agreement_all_rows": float(agree.float().mean()),
                "median_top2_gap": float(gap.median())}
    _record(f"actor_{dtype}_{form}_{players}p", {"measured": measured, "tolerance": {k: tol[k] for k in ("q_abs", "argmax_gap")}})
    assert measured["q_max_abs_err"] <= tol["q_abs"], measured
    assert measured["rows_with_clear_gap"] > (0.55 if dtype == "bfloat16" else 0.85), measured
    assert bool(agree[clear].all()), measured
    assert bool(legal.gather(1, act.long()[:, None]).all())
