"""Learner arithmetic (torch, hanabi_agents.rlax_dqn.learning / noisy_mlp / rlax_rainbow) against the
independent numpy float64 restatement in oracle/learner_oracle.py and hand-worked known answers.
The reference learner itself cannot run here (jax/haiku/rlax absent): parity unpinned vs a reference run."""
import numpy as np
import pytest
import torch

from hanabi_agents.rlax_dqn import (ActionSpec, DQNAgent, DQNLearning, DQNPolicy, NoisyMLP, ObservationSpec,
                                    RlaxRainbowParams, learning as L)
from hanabi_agents.rlax_dqn.transition import Transition
from oracle import learner_oracle as LO


def test_projection_dopamine_worked_example():
    """hanabi_agents/rainbow/rainbow_agent.py:262-266,400-401: supports [[0,2,4,6,8],[1,3,4,5,6]], weights
    [[.1,.6,.1,.1,.1],[.1,.2,.5,.1,.1]], target support [4..8] -> [[.8,0,.1,0,.1],[.8,.1,.1,0,0]]."""
    z_p = torch.tensor([[0., 2, 4, 6, 8], [1, 3, 4, 5, 6]])
    w = torch.tensor([[.1, .6, .1, .1, .1], [.1, .2, .5, .1, .1]])
    z_q = torch.tensor([4., 5, 6, 7, 8])
    out = L.categorical_l2_project(z_p, w, z_q)
    assert torch.allclose(out, torch.tensor([[.8, 0, .1, 0, .1], [.8, .1, .1, 0, 0]]), atol=1e-6)
    for i in range(2):
        assert np.allclose(LO.project_general(z_p[i].numpy(), w[i].numpy(), z_q.numpy()), out[i].numpy(), atol=1e-6)


def test_projection_matches_floor_ceil_form_and_conserves_mass():
    rng = np.random.default_rng(0)
    support = np.linspace(-25, 25, 51)
    probs = rng.dirichlet(np.ones(51), 16)
    r = rng.integers(-3, 4, 16).astype(float)
    z_p = r[:, None] + 0.99 * support[None]
    out = L.categorical_l2_project(torch.tensor(z_p), torch.tensor(probs), torch.tensor(support)).numpy()
    ref = np.stack([LO.project_uniform(z_p[i], probs[i], -25, 25, 51) for i in range(16)])
    assert np.allclose(out, ref, atol=1e-12) and np.allclose(out.sum(1), 1.0)


def _net_and_layers(obs_len=30, hidden=16, out=4 * 5, seed=3, b_noise_rows=None):
    net = NoisyMLP(obs_len, (hidden, out), seed=seed, compute_dtype=torch.float64).double()
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():  # biases are zero-initialised like the reference; perturb them so they are exercised
        for layer in net.layers:
            for n in ("b", "b_mu", "b_sigma"):
                getattr(layer, n).copy_(torch.randn(getattr(layer, n).shape, generator=g, dtype=torch.float64) * 0.1)
    layers = [{n: getattr(l, n).detach().numpy() for n in ("w", "b", "w_mu", "b_mu", "w_sigma", "b_sigma", "eps_w", "eps_b")}
              for l in net.layers]
    return net, layers


def test_noisy_mlp_merged_equals_literal_and_numpy():
    net, layers = _net_and_layers()
    x = torch.tensor(np.random.default_rng(1).integers(0, 2, (7, 30)).astype(np.float64))
    merged, literal = net(x, merged=True), net(x, merged=False)
    assert torch.allclose(merged, literal, atol=1e-12)
    assert np.allclose(LO.noisy_mlp_forward(x.numpy(), layers), merged.detach().numpy(), atol=1e-12)
    # six parameters per layer, reference counts (2 582 004 for the 2-player net)
    assert NoisyMLP(658, (512, 20 * 51)).num_parameters() == 2582004
    assert NoisyMLP(1280, (512, 48 * 51)).num_parameters() == 5735088
    # explicit per-row bias noise [batch, out] (the reference draws eps_b after broadcasting, C-2)
    eb = torch.randn(7, 16, dtype=torch.float64)
    y = net.layers[0](x, eps_b=eb)
    l0 = layers[0]
    want = x.numpy() @ (l0["w"] + l0["w_mu"] + l0["w_sigma"] * l0["eps_w"]) + l0["b"] + l0["b_mu"] + l0["b_sigma"] * eb.numpy()
    assert np.allclose(y.detach().numpy(), want, atol=1e-12)


def test_c51_double_q_loss_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    b, a, k = 12, 4, 5
    atoms = torch.linspace(-2, 2, k, dtype=torch.float64).repeat(a, 1)
    lt, ls, l1 = (torch.tensor(rng.standard_normal((b, a, k))) for _ in range(3))
    a_tm1 = torch.tensor(rng.integers(0, a, b))
    r = torch.tensor(rng.integers(-1, 2, b).astype(float))
    term = torch.tensor(rng.integers(0, 2, b).astype(float))
    for mask in (None, term):
        td = L.categorical_double_q_td(l1, a_tm1, r, 0.99, atoms, lt, ls, mask)
        ref = LO.c51_double_q_td(l1.numpy(), a_tm1.numpy(), r.numpy(), 0.99, atoms[0].numpy(), lt.numpy(), ls.numpy(),
                                 None if mask is None else mask.numpy())
        assert np.allclose(td.numpy(), ref, atol=1e-10)
    pr = rng.random(b) + 0.01
    assert np.allclose(L.is_weights(torch.tensor(pr), 0.4).numpy(), LO.is_weights(pr, 0.4), rtol=1e-6)
    # q is the MEAN of probs*atoms (C-3)
    assert torch.allclose(L.expected_q(l1, atoms), (torch.softmax(l1, -1) * atoms).sum(-1) / k)


def test_policy_sampling_semantics():
    q = torch.tensor([[1.0, 3.0, 3.0, float("-inf")], [0.5, float("-inf"), 0.2, 0.1]])
    legal = torch.tensor([[1, 1, 1, 0], [1, 0, 1, 1]], dtype=torch.int8)
    # greedy: ties broken uniformly by u_pick, never an illegal move
    assert DQNPolicy.sample(q, legal, 0.0, torch.ones(2), torch.tensor([0.0, 0.0])).tolist() == [1, 0]
    assert DQNPolicy.sample(q, legal, 0.0, torch.ones(2), torch.tensor([0.99, 0.99])).tolist() == [2, 0]
    # exploring rows (u_explore < eps) pick the k-th LEGAL move
    assert DQNPolicy.sample(q, legal, 0.5, torch.tensor([0.1, 0.1]), torch.tensor([0.0, 0.99])).tolist() == [0, 3]
    # distribution check: eps-greedy frequencies
    n = 40000
    qq = q[:1].repeat(n, 1)
    ll = legal[:1].repeat(n, 1)
    g = torch.Generator().manual_seed(0)
    acts = DQNPolicy.sample(qq, ll, 0.3, torch.rand(n, generator=g), torch.rand(n, generator=g))
    freq = torch.bincount(acts.long(), minlength=4).double() / n
    want = torch.tensor([0.1, 0.45, 0.45, 0.0], dtype=torch.float64)   # (1-eps)*[0,.5,.5,0] + eps*[1/3,1/3,1/3,0]
    assert torch.allclose(freq, want, atol=0.01)


def test_update_is_adam_on_the_c51_loss_and_target_sync_cadence():
    """One DQNAgent.update() equals a hand-rolled step: loss gradient by autograd on the same batch, then
    optix.adam arithmetic in numpy (eps outside sqrt, 3.125e-5); target net syncs after step 0 (C-10)."""
    torch.manual_seed(0)
    n, obs_len, n_act = 8, 20, 3
    params = RlaxRainbowParams(use_priority=False, train_batch_size=8, experience_buffer_size=8, layers=[8], n_atoms=5,
                               atom_vmax=2, target_update_period=2)
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cpu")
    rng = np.random.default_rng(0)
    o1 = rng.integers(0, 2, (n, obs_len)).astype(np.int8)
    o2 = rng.integers(0, 2, (n, obs_len)).astype(np.int8)
    legal = np.ones((n, n_act), np.int8)
    agent.add_experience_first((None, (o1, legal)), np.zeros(n))
    agent.add_experience((None, (o2, legal)), rng.integers(0, n_act, n), rng.integers(0, 2, n).astype(float), np.ones(n))
    assert agent.experience.size == n
    # deterministic batch: sample all rows in order
    agent.experience.sample_indices_dev = lambda b: torch.arange(b)
    before = {k: v.detach().clone() for k, v in agent.online.named_parameters()}
    tr = agent.experience.gather_dev(torch.arange(n))
    tr = tr._replace(observation_tm1=tr.observation_tm1.float(), observation_t=tr.observation_t.float())
    loss, _ = DQNLearning.loss(agent.online, agent.target, agent.atoms, tr, 0.99, torch.ones(n, dtype=torch.float64), 0.4)
    grads = torch.autograd.grad(loss, list(agent.online.parameters()))
    agent.update()
    for (name, p), g in zip(agent.online.named_parameters(), grads):
        want, _, _ = LO.adam_step(before[name].numpy().astype(np.float64), g.numpy().astype(np.float64), 0.0, 0.0, 1)
        assert np.allclose(p.detach().numpy(), want, atol=1e-6), name
    # target synced after step 0 (train_step % period == 0), not after step 1
    assert all(torch.equal(a, b) for a, b in zip(agent.online.state_dict().values(), agent.target.state_dict().values()))
    agent.update()
    assert not all(torch.equal(a, b) for a, b in zip(agent.online.parameters(), agent.target.parameters()))
    agent.update()  # step 2 -> sync again
    assert all(torch.equal(a, b) for a, b in zip(agent.online.parameters(), agent.target.parameters()))
    assert agent.train_step == 3


def test_agent_api_bookkeeping_first_mid_last():
    """rlax_rainbow.py:292-308: FIRST rows seed last_obs; MID/LAST rows become transitions in row order,
    terminal = (step_type == 2); last_obs advances for those rows only."""
    n, obs_len, n_act = 6, 10, 4
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act),
                     RlaxRainbowParams(use_priority=False, experience_buffer_size=16, layers=[4]), device="cpu")
    o = [np.full((n, obs_len), k, np.int8) + np.arange(n, dtype=np.int8)[:, None] for k in (0, 10, 20)]
    legal = np.ones((n, n_act), np.int8)
    agent.add_experience_first((None, (o[0], legal)), np.zeros(n))
    st = np.array([1, 2, 0, 1, 0, 2])
    agent.add_experience_first((None, (o[1], legal)), st)     # rows 2, 4 are FIRST again
    agent.add_experience((None, (o[1], legal)), np.arange(n) % n_act, np.arange(n, dtype=float), st)
    buf = agent.experience
    assert buf.size == 4 and buf.oldest_entry == 4
    t = buf[np.arange(4)]
    rows = [0, 1, 3, 5]
    assert np.array_equal(t.observation_tm1, o[0][rows]) and np.array_equal(t.observation_t, o[1][rows])
    assert t.action_tm1[:, 0].tolist() == [0, 1, 3, 1] and t.reward_t[:, 0].tolist() == [0.0, 1.0, 3.0, 5.0]
    assert t.terminal_t[:, 0].tolist() == [False, True, False, True]
    assert t.observation_tm1.dtype == np.int8 and t.reward_t.dtype == np.float64 and t.terminal_t.dtype == bool
    assert np.array_equal(agent.last_obs.numpy(), o[1])       # FIRST rows via add_experience_first, others via add_experience
    assert agent.requires_vectorized_observation() is True
    assert repr(agent).startswith("<rlax_dqn.DQNAgent(params=")


def test_params_defaults_match_reference():
    p = RlaxRainbowParams()   # hanabi_agents/rlax_dqn/params.py:8-20
    assert (p.train_batch_size, p.target_update_period, p.discount, p.learning_rate, p.layers, p.use_double_q, p.use_priority,
            p.experience_buffer_size, p.seed, p.n_atoms, p.atom_vmax) == (256, 500, 0.99, 0.001, [512], True, True, 2 ** 19, 1234, 51, 25)
    assert p.epsilon(0) == 0.1 and p.beta_is(123) == 0.4
    assert p._fields[:13] == ("train_batch_size", "target_update_period", "discount", "epsilon", "learning_rate", "layers",
                              "use_double_q", "use_priority", "experience_buffer_size", "seed", "n_atoms", "atom_vmax", "beta_is")
    assert Transition._fields == ("observation_tm1", "action_tm1", "reward_t", "observation_t", "legal_moves_t", "terminal_t")


def test_save_restore_weights(tmp_path):
    a = DQNAgent(ObservationSpec((2, 12)), ActionSpec(3), RlaxRainbowParams(use_priority=False, layers=[4], experience_buffer_size=4),
                 device="cpu")
    a.save_weights(str(tmp_path), "x")
    assert (tmp_path / "rlax_rainbow_x_online.pkl").exists() and (tmp_path / "rlax_rainbow_x_target.pkl").exists()
    b = DQNAgent(ObservationSpec((2, 12)), ActionSpec(3), RlaxRainbowParams(use_priority=False, layers=[4], experience_buffer_size=4,
                                                                             seed=77), device="cpu")
    b.restore_weights(str(tmp_path / "rlax_rainbow_x_online.pkl"), str(tmp_path / "rlax_rainbow_x_target.pkl"))
    assert all(torch.equal(x, y) for x, y in zip(a.online.state_dict().values(), b.online.state_dict().values()))


def test_scalar_double_dqn_head():
    """BASELINE config 2 (vanilla DQN): rlax_dqn.py:170-205 semantics."""
    q_tm1 = torch.tensor([[1.0, 2.0], [0.5, 0.0]], requires_grad=True)
    td = L.double_q_td(q_tm1, torch.tensor([1, 0]), torch.tensor([1.0, 0.0]), 0.5, torch.tensor([[4.0, 6.0], [3.0, 9.0]]),
                       torch.tensor([[0.0, 1.0], [1.0, 0.0]]), torch.tensor([0.0, 1.0]))
    assert torch.allclose(td, torch.tensor([1 + 0.5 * 6 - 2, 0 + 0 - 0.5]))
    big = L.clip_gradient(q_tm1.sum() * 100.0)
    big.backward()
    assert torch.all(q_tm1.grad == 100.0)  # identity forward; the clip acts on the incoming cotangent (1.0 here)
    agent = DQNAgent(ObservationSpec((4, 12)), ActionSpec(3), RlaxRainbowParams(use_priority=False, distributional=False, layers=[8],
                                                                                 experience_buffer_size=8, train_batch_size=4), device="cpu")
    o = np.random.default_rng(0).integers(0, 2, (4, 12)).astype(np.int8)
    legal = np.ones((4, 3), np.int8)
    agent.add_experience_first((None, (o, legal)), np.zeros(4))
    agent.add_experience((None, (o, legal)), agent.explore((None, (o, legal))), np.ones(4), np.ones(4))
    agent.update()
    assert torch.isfinite(agent.last_loss)


def test_pending_tree_fills_are_coalesced_and_bounded():
    """split update: inserts only queue (start, rows) for the sum tree; a seat that inserts without training must not grow that
    list by one entry per step: contiguous ranges merge, a ring wrap starts a new range, and a whole ring's worth
    collapses to the whole ring."""
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    agent = DQNAgent(ObservationSpec((4, 171)), ActionSpec(11), RlaxRainbowParams(experience_buffer_size=64, train_batch_size=4, use_priority=False),
                     device="cpu")
    for start in (0, 4, 8, 12):
        agent._queue_fill(start, 4)
    assert agent._pending_fills == [(0, 16)]
    agent._queue_fill(60, 4)            # not contiguous with the previous range
    agent._queue_fill(0, 4)             # the ring wrapped: a new range
    assert agent._pending_fills == [(0, 16), (60, 4), (0, 4)]
    for start in range(4, 64, 4):
        agent._queue_fill(start, 4)
    assert agent._pending_fills == [(0, 64)]
    for _ in range(1000):               # a frozen seat in a long run
        agent._queue_fill(0, 64)
    assert len(agent._pending_fills) == 1
