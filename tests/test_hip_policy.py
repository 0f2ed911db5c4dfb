"""GPU: fused actor tail (hb_policy_act) and fused replay insert (hb_replay_insert) against plain
PyTorch fp32 references of the same ops (DQNPolicy.q_values / DQNPolicy.sample, ExperienceBuffer.add_transitions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def K_lib():
    import hanabi_hip

    return hanabi_hip.lib()


def C_byref(x):
    import ctypes

    return ctypes.byref(x)


@pytest.mark.parametrize("n,a,k", [(1000, 20, 51), (257, 48, 51), (64, 11, 51), (3, 20, 51), (500, 30, 7)])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_policy_q_values_and_greedy_choice(n, a, k, dtype):
    import torch

    from hanabi_agents.rlax_dqn import learning as L
    from hanabi_hip import ops

    g = torch.Generator(device="cuda").manual_seed(n + a)
    dt = getattr(torch, dtype)
    logits = (torch.randn(n, a * k, device="cuda", generator=g) * 2).to(dt)
    legal = (torch.rand(n, a, device="cuda", generator=g) < 0.6).to(torch.int8)
    legal[:, 0] = 1
    support = torch.linspace(-25, 25, k, device="cuda")
    q = torch.empty(n, a, device="cuda")
    act = ops.policy_act(logits, legal, support, 0.0, seed=7, draw=3, q_out=q)
    # the same logits inside a wider, padded row (the second GEMM's N is padded to a multiple of 64)
    wide = torch.full((n, (a * k + 63) // 64 * 64 + 64), 7.0, device="cuda", dtype=dt)
    wide[:, :a * k] = logits
    q_w = torch.empty(n, a, device="cuda")
    act_w = ops.policy_act(wide, legal, support, 0.0, seed=7, draw=3, q_out=q_w)
    assert torch.equal(act, act_w) and torch.equal(q, q_w)
    ref = L.expected_q(logits.float().view(n, a, k), support.repeat(a, 1))       # torch fp32 reference on the same inputs
    assert torch.allclose(q, ref, rtol=2e-5, atol=2e-6)
    masked = torch.where(legal.bool(), q, torch.full_like(q, float("-inf")))
    # greedy: the chosen move is legal and attains the maximum of the kernel's own q
    chosen = masked.gather(1, act.long()[:, None])[:, 0]
    assert torch.equal(chosen, masked.max(dim=1).values)
    assert bool(legal.gather(1, act.long()[:, None]).all())


def test_policy_epsilon_greedy_distribution_and_ties():
    import torch

    from hanabi_hip import ops

    n, a, k = 60000, 4, 5
    logits = torch.zeros(n, a * k, device="cuda")
    logits[:, 1 * k + 4] = 3.0   # action 1 and 2 tie for the best q, action 3 illegal
    logits[:, 2 * k + 4] = 3.0
    legal = torch.tensor([1, 1, 1, 0], dtype=torch.int8, device="cuda").repeat(n, 1).contiguous()
    support = torch.linspace(-2, 2, k, device="cuda")
    act = ops.policy_act(logits, legal, support, 0.3, seed=11, draw=1)
    freq = torch.bincount(act.long(), minlength=a).double().cpu() / n
    want = torch.tensor([0.1, 0.45, 0.45, 0.0], dtype=torch.float64)     # (1-eps)*[0,.5,.5,0] + eps*[1/3,1/3,1/3,0]
    assert torch.allclose(freq, want, atol=0.01)
    # deterministic given (seed, draw, game id); different draw -> different sample
    again = ops.policy_act(logits, legal, support, 0.3, seed=11, draw=1)
    other = ops.policy_act(logits, legal, support, 0.3, seed=11, draw=2)
    assert torch.equal(act, again) and not torch.equal(act, other)
    shifted = ops.policy_act(logits[:100], legal[:100], support, 0.3, seed=11, draw=1, first_game_id=50)
    assert torch.equal(shifted[:50], act[50:100])                           # keyed by GLOBAL game id


def test_agent_fused_explore_agrees_with_torch_policy():
    """DQNAgent on cuda (fused path, fp32 GEMMs) picks the arg-max of the torch-op q wherever the top-2 gap is clear."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, DQNPolicy, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 512, 658, 20
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act),
                     RlaxRainbowParams(experience_buffer_size=1024, epsilon=0.0), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.7).to(torch.int8)
    legal[:, 5] = 1
    act = agent.exploit((None, (obs, legal)))
    q = DQNPolicy.q_values(agent.online, agent.atoms, obs.float(), legal)
    top2 = q.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert clear.sum() > n // 2
    assert torch.equal(act[clear].long(), q.argmax(1)[clear])
    assert bool(legal.gather(1, act.long()[:, None]).all())
    # numpy in -> numpy out, like the reference
    out = agent.explore((None, (obs.cpu().numpy(), legal.cpu().numpy())))
    assert isinstance(out, np.ndarray) and out.shape == (n,)


@pytest.mark.parametrize("cap,n,rounds", [(1000, 96, 25), (64, 64, 3), (4096, 1000, 9)])
def test_fused_replay_insert_matches_torch_ring(cap, n, rounds):
    import torch

    from hanabi_agents.rlax_dqn.experience_buffer import ExperienceBuffer
    from hanabi_hip import ops

    obs_len, n_act = 658, 20
    fused = ExperienceBuffer(obs_len, n_act, 1, cap, device="cuda")
    plain = ExperienceBuffer(obs_len, n_act, 1, cap, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(cap)
    last = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.5).to(torch.int8)
    last_ref = last.clone()
    for r in range(rounds):
        obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.5).to(torch.int8)
        legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.5).to(torch.int8)
        act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
        rew = torch.randint(-3, 4, (n,), device="cuda", generator=g).float()
        st = torch.randint(1, 3, (n,), device="cuda", generator=g).to(torch.int8)
        ops.replay_insert(last, obs, legal, act, rew, st, fused, fused.oldest_entry)
        fused._advance(n)
        plain.add_transitions(last_ref, act.reshape(-1, 1), rew.reshape(-1, 1), obs, legal, (st == 2).reshape(-1, 1))
        last_ref = obs.clone()
        assert torch.equal(last, last_ref)
        assert (fused.oldest_entry, fused.size) == (plain.oldest_entry, plain.size)
    for name in ("_obs_tm1_buf", "_obs_t_buf", "_act_tm1_buf", "_lms_t_buf", "_rew_t_buf", "_terminal_t_buf"):
        assert torch.equal(getattr(fused, name), getattr(plain, name)), name


def test_graphed_update_equals_eager_update():
    """The HIP-graph replay of update() performs the same arithmetic as the eager launch sequence."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 256, 658, 20
    params = RlaxRainbowParams(use_priority=False, train_batch_size=256, experience_buffer_size=256, target_update_period=4)
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=g) for g in (True, False)]
    g = torch.Generator(device="cuda").manual_seed(1)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    for a in agents:
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, torch.ones(n, dtype=torch.int8, device="cuda"))
        a.experience.sample_indices_dev = lambda b: torch.arange(b, device="cuda")   # same batch, fixed order
    for step in range(6):    # (the capture's warm-up updates are rolled back: update() is exactly one gradient step either way)
        for a in agents:
            a.update()
    w = [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]) for a in agents]
    assert torch.allclose(w[0], w[1], rtol=1e-4, atol=1e-6)
    assert agents[0]._graph1 is not None and agents[1]._graph1 is None
    assert torch.allclose(agents[0].last_loss, agents[1].last_loss, rtol=1e-4)


def test_two_kernel_actor_follows_graph_replayed_updates():
    """bf16, a batch below ActorMFMA.fused_min_rows (the two-kernel actor form) and the update replayed from its HIP graph: the
    policy must act on the weights of the LAST update. (The two-kernel form's weight copies are refreshed lazily, by a mark the
    captured pack sets on the host — which a replayed graph does not repeat: FusedLearner.weights_updated renews it. Round 3
    regression: without that the form kept acting on the weights of the captured update and a 4 096-game run learned nothing.)"""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams, bitpack

    n, obs_len, n_act = 256, 658, 20
    params = RlaxRainbowParams(use_priority=False, train_batch_size=256, experience_buffer_size=256, target_update_period=50,
                               compute_dtype="bfloat16", learning_rate=0.02, packed_obs=True)
    a = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=True)
    g = torch.Generator(device="cuda").manual_seed(2)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    b1, b2 = bitpack.pack(o1), bitpack.pack(o2)
    a.add_experience_first((None, (b1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
    a.add_experience((None, (b2, legal)), act, rew, torch.ones(n, dtype=torch.int8, device="cuda"))
    fl = a._fused_learner()
    assert fl.actor is not None and fl.actor.fused and fl.actor.two_kernel and not fl.actor.takes_fused(b1)
    for _ in range(8):           # the first ones capture, the rest replay
        a.update()
        a.exploit((None, (b1, legal)))    # a policy call between updates, as in self-play (it consumes the lazy mark)
    assert a._graph1 is not None
    a.exploit((None, (b1, legal)))
    q_two = fl.actor.q.clone()            # the two-kernel form's q values of the call just made
    fl.actor.fused_min_rows = 0           # the one-kernel form on the same weights (its copies are packed inside the graph)
    q_one = fl.actor.q_values(b1, a.atoms[0].contiguous()).clone()
    torch.cuda.synchronize()
    assert float((q_two - q_one).abs().max()) <= 2e-3, float((q_two - q_one).abs().max())
    w1_now = fl.eff[0][0][:obs_len].float()
    assert float((w1_now - a.online.layers[0].effective()[0].float()).abs().max()) < 1e-2   # (eff follows the optimizer)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("players", [2, 5])
def test_adam_that_packs_equals_adam_plus_pack_launches(dtype, players):
    """hb_noisy_adam_multi_pack (optimizer step + the thin GEMMs' transposed online weights + the one-kernel actor's fragment-major
    copies and fp32 biases in ONE launch) against hb_noisy_adam_multi followed by hb_actor_pack_weights and hb_actor_fused_pack:
    identical parameters, moments and effective weights after 4 updates, and copies that equal, bit for bit, what the two packers
    make of the final effective weights. 2-player (658 -> 512 -> 20 x 51) and 5-player (1280 -> 512 -> 48 x 51) shapes."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    obs_len, n_act = {2: (658, 20), 5: (1280, 48)}[players]
    n = 256
    params = RlaxRainbowParams(use_priority=False, train_batch_size=n, experience_buffer_size=n, target_update_period=3,
                               compute_dtype=dtype, learning_rate=0.01)
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False) for _ in (0, 1)]
    g = torch.Generator(device="cuda").manual_seed(players)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    for a, fused_pack in zip(agents, (True, False)):
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, torch.ones(n, dtype=torch.int8, device="cuda"))
        a.experience.sample_indices_dev = lambda b: torch.arange(b, device="cuda")
        fl = a._fused_learner()
        assert fl.actor is not None and fl.actor.fused and fl.thin
        fl.pack_actor()
        fl.adam_pack = fused_pack     # (HB_ADAM_PACK=1 in a run; the default is the separate launches: the merged kernel is slower)
        for _ in range(4):      # (one of them is followed by a target sync)
            a.update()
    fa, fb = agents[0]._fl, agents[1]._fl
    assert fa.adam_pack is True and fb.adam_pack is False
    for pa, pb in zip(agents[0].online.parameters(), agents[1].online.parameters()):
        assert torch.equal(pa, pb)
    for ka in fa.state:
        assert torch.equal(fa.state[ka][0], fb.state[ka][0]) and torch.equal(fa.state[ka][1], fb.state[ka][1])
    for (wa, ba), (wb, bb) in zip(fa.eff, fb.eff):
        assert torch.equal(wa, wb) and torch.equal(ba, bb)
    # the copies Adam wrote == the packers' output for the same effective weights
    got = [fa.w1catT.clone(), fa.w2stT.clone()] + [t.clone() for t in fa.actor.fsets[0]]
    fa._transpose(0)
    (w1, b1), (w2, b2) = fa.eff
    fa.actor.pack(w1, b1, w2, b2, 0, lazy_two_kernel=True)
    want = [fa.w1catT, fa.w2stT] + list(fa.actor.fsets[0])
    for k, (x, y) in enumerate(zip(got, want)):
        assert torch.equal(x, y), f"copy {k} differs: {(x != y).sum().item()} elements"
    # one pack launch (hb_actor_fused_pack_thin: the thin GEMMs' transposed copies as extra outputs of the one-kernel actor's packer,
    # what every update runs by default) == the two launches
    want_t = [fa.w1catT.clone(), fa.w2stT.clone()]
    fa.w1catT[:fa.H].zero_()
    fa.w2stT[0].zero_()
    assert fa.pack_thin and fb.pack_thin
    fa.actor.pack(w1, b1, w2, b2, 0, lazy_two_kernel=True, thin=fa._thin_out())
    assert torch.equal(fa.w1catT, want_t[0]) and torch.equal(fa.w2stT, want_t[1])
    for x, y in zip(fa.actor.fsets[0], want[2:]):
        assert torch.equal(x, y)
    assert torch.equal(fb.w1catT, fa.w1catT) and torch.equal(fb.w2stT, fa.w2stT)   # (the other agent's, written during its updates)
    # and the unfused agent's copies, made by the packers during its updates, are the same
    fb.pack_actor()
    if fb.actor_stale is False and fb._packed_in_part2:
        for x, y in zip(fa.actor.fsets[0], fb.actor.fsets[0]):
            assert torch.equal(x, y)


def test_graph_warm_up_rolls_back_the_uniform_replay_generator():
    """Uniform replay (vanilla DQN, BASELINE config 2) draws its batch indices from the buffer's own generator. The graph
    capture's three warm-up updates consume draws; they are rolled back together with the weights, so after every
    update() the generator of a graph-enabled agent is in the state of the eager agent's: exactly one draw per update, none
    lost to the warm-up. (The VALUES drawn inside a replayed graph come from torch's capture-safe Philox path, which maps the
    same (seed, offset) to other numbers than the eager kernel: the two agents sample different, equally distributed batches.)"""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 512, 658, 20
    params = RlaxRainbowParams(distributional=False, use_priority=False, train_batch_size=64, experience_buffer_size=512,
                               target_update_period=4, compute_dtype="float32")
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=g) for g in (True, False)]
    g = torch.Generator(device="cuda").manual_seed(2)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    for a in agents:
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, torch.ones(n, dtype=torch.int8, device="cuda"))
    assert torch.equal(agents[0].experience._gen.get_state(), agents[1].experience._gen.get_state())
    w0 = torch.cat([p.detach().reshape(-1) for p in agents[0].online.parameters()]).clone()
    for step in range(5):
        for a in agents:
            a.update()
        assert torch.equal(agents[0].experience._gen.get_state(), agents[1].experience._gen.get_state()), step
    assert agents[0]._graph1 is not None and agents[1]._graph1 is None
    for a in agents:
        w = torch.cat([p.detach().reshape(-1) for p in a.online.parameters()])
        assert torch.isfinite(w).all() and float((w - w0).abs().max()) > 0


@pytest.mark.parametrize("fused", [False, True])
def test_graphed_nstep_update_follows_the_ring_after_capture(fused):
    """HIP graphs + n_step > 1: the n-step chain walk must read the ring's size / write pointer from device scalars, not
    from host ints frozen at capture time. Graphed and eager agents see the same inserts BETWEEN updates (so the ring
    advances after the capture) and must stay equal — non-fused (torch gather_nstep_dev) and fused (hb_replay_gather)."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 64, 658, 20
    params = RlaxRainbowParams(use_priority=False, train_batch_size=64, experience_buffer_size=64 * 7, target_update_period=4,
                               n_step=3, mask_terminal=True)
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=g,
                       use_fused_learner=fused) for g in (True, False)]
    g = torch.Generator(device="cuda").manual_seed(1)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")

    def insert(first=False):
        obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
        act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
        rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
        st = (torch.rand(n, device="cuda", generator=g) < 0.15).to(torch.int8) + 1
        for a in agents:
            if first:
                a.add_experience_first((None, (obs, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
            else:
                a.add_experience_dense((None, (obs, legal)), act, rew, st)

    insert(first=True)
    insert()
    insert()
    pick = [torch.randint(0, 10 ** 6, (64,), device="cuda", generator=g) for _ in range(12)]
    cur = torch.zeros(64, dtype=torch.int64, device="cuda")   # persistent: the captured graph reads it by address
    for a in agents:   # same sampled rows for both agents, drawn from the ring as it is NOW (device size scalar)
        a.experience.sample_indices_dev = lambda b, a=a: cur % a.experience._size_t.long()
    for step in range(12):
        cur.copy_(pick[step])
        for a in agents:
            a.update()
        insert()      # the ring grows, then wraps (capacity 7 inserts), while the captured graph keeps replaying
    assert agents[0]._graph1 is not None and agents[1]._graph1 is None
    w = [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]) for a in agents]
    assert torch.allclose(w[0], w[1], rtol=1e-4, atol=1e-6)


def test_batched_gpu_loss_equals_reference_loss_and_gradient():
    """DQNAgent._loss_batched (one 2B-row online pass, cached target weights) vs DQNLearning.loss (three module
    forwards) in fp32: same loss, same per-sample |td|, same gradient."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, DQNLearning, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 256, 658, 20
    for mask in (False, True):
        agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act),
                         RlaxRainbowParams(use_priority=False, experience_buffer_size=256, mask_terminal=mask), device="cuda",
                         use_graphs=False)
        with torch.no_grad():  # make target != online and biases non-zero
            for p in agent.target.parameters():
                p.add_(torch.randn_like(p) * 0.01)
            for layer in agent.online.layers:
                layer.b_sigma.normal_(0, 0.1)
                layer.b_mu.normal_(0, 0.1)
        g = torch.Generator(device="cuda").manual_seed(2)
        from hanabi_agents.rlax_dqn.transition import Transition

        tr = Transition((torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8),
                        torch.randint(0, n_act, (n, 1), device="cuda", generator=g).to(torch.int8),
                        torch.randint(-2, 3, (n, 1), device="cuda", generator=g).float(),
                        (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8),
                        torch.ones(n, n_act, dtype=torch.int8, device="cuda"),
                        torch.rand(n, 1, device="cuda", generator=g) < 0.3)
        prios = torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.01
        agent._beta.fill_(0.4)
        params = list(agent.online.parameters())
        loss_a, td_a = agent._loss_batched(tr, prios)
        grads_a = torch.autograd.grad(loss_a, params)
        trf = tr._replace(observation_tm1=tr.observation_tm1.float(), observation_t=tr.observation_t.float())
        loss_b, td_b = DQNLearning.loss(agent.online, agent.target, agent.atoms, trf, 0.99, prios, 0.4, mask)
        grads_b = torch.autograd.grad(loss_b, params)
        assert torch.allclose(loss_a, loss_b, rtol=1e-5)
        assert torch.allclose(td_a, td_b, rtol=1e-4, atol=1e-5)
        for ga, gb in zip(grads_a, grads_b):
            # fp32 GEMMs of different M (and whatever algorithm hipBLASLt picks on the box): the summation order differs. Measured
            # against the tensor's scale, not element by element (an element near zero has no relative error to speak of)
            scale = float(gb.abs().max())
            assert float((ga - gb).abs().max()) <= 1e-7 + 1e-3 * scale, (float((ga - gb).abs().max()), scale)
            assert float((ga.double() - gb.double()).norm() / gb.double().norm()) < 1e-4


@pytest.mark.parametrize("mask,priority,n_step", [(False, False, 1), (True, False, 1), (True, True, 1), (True, False, 3)])
def test_fused_learner_equals_autograd_learner(mask, priority, n_step):
    """FusedLearner (hb_replay_gather + hb_c51_loss_grad + hand-written backward + hb_noisy_adam) against the
    PyTorch-autograd fp32 reference path (DQNLearning.loss + torch.optim.Adam) on the same batches."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 256, 658, 20
    params = RlaxRainbowParams(use_priority=priority, train_batch_size=256, experience_buffer_size=256 * n_step,
                               target_update_period=3, mask_terminal=mask, n_step=n_step)
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False,
                       use_fused_learner=f) for f in (True, False)]
    g = torch.Generator(device="cuda").manual_seed(1)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    st = torch.randint(1, 3, (n,), device="cuda", generator=g).to(torch.int8)
    for a in agents:
        with torch.no_grad():
            for layer in a.online.layers:     # exercise the bias-noise path too
                layer.b_sigma.fill_(0.05)
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, st)
        for extra in range(n_step - 1):   # further dense inserts so that n-step chains have successors
            a.add_experience_dense((None, (o1 if extra % 2 else o2, legal)), act, rew + extra, st)
        a.target.load_state_dict(a.online.state_dict())
    idx = torch.randperm(n, device="cuda", generator=g)
    pri = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.05) / n
    fused, ref = agents
    fused._sample_indices = lambda: (idx, pri)
    ref._sample_indices = lambda: (idx, pri)
    for step in range(5):
        fused.update()
        ref.update()
        assert torch.allclose(fused.last_loss, ref.last_loss, rtol=2e-4), step
    for (na, pa), (nb, pb) in zip(fused.online.named_parameters(), ref.online.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=2e-5), na
    for pa, pb in zip(fused.target.parameters(), ref.target.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=2e-5)
    if priority:
        la = fused.experience.sum_tree.get_values(range(n))
        lb = ref.experience.sum_tree.get_values(range(n))
        assert np.allclose(la, lb, rtol=2e-3)
    # the actor sees the freshly written effective weights
    w_eff, _ = fused.online.layers[0].effective()
    got = fused._effective_weights()[0][0].float()
    assert torch.allclose(got[:obs_len], w_eff, rtol=1e-2, atol=1e-3) and not got[obs_len:].any()   # K padding stays zero


@pytest.mark.parametrize("shape", [(658, 20, 512, 256), (1280, 48, 512, 256), (171, 11, 32, 256), (658, 20, 40, 256),
                                   (658, 20, 512, 30), (171, 11, 32, 33)])   # (odd batches: the staged rows stay 16-byte aligned)
@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
def test_sparse_backward_equals_dense_chain(shape, dtype):
    """hb_c51_loss_sparse + hb_c51_backward (csrc/learner2.hip: compact dLoss/dlogits, one launch for dH / db1 / dW2 / db2)
    against the dense chain they replace (hb_c51_loss_grad + dW2 GEMM + hb_colsum + dH GEMM + hb_relu_bwd_colsum) on the same
    batch: same td and IS weights; in fp32 the gradients agree to summation order; in bf16 / f16 the sparse path keeps
    dLoss/dlogits in fp32 where the dense one rounds it, so they agree to that rounding."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    obs_len, n_act, hidden, n = shape
    params = RlaxRainbowParams(use_priority=False, train_batch_size=n, experience_buffer_size=n, mask_terminal=True,
                               compute_dtype=dtype, layers=[hidden])
    agents = [DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False) for _ in (0, 1)]
    g = torch.Generator(device="cuda").manual_seed(obs_len)
    o1 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    o2 = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.4).to(torch.int8)
    legal = torch.ones(n, n_act, dtype=torch.int8, device="cuda")
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    act[act == 3] = 2                                                   # action 3 is never taken: its dW2 / db2 slices must be zero
    if hidden == 512:
        act[:n * 200 // 256] = 7                                        # a greedy policy: most of the batch took ONE action
    rew = torch.randint(-1, 2, (n,), device="cuda", generator=g).float()
    st = torch.randint(1, 3, (n,), device="cuda", generator=g).to(torch.int8)
    idx = torch.randperm(n, device="cuda", generator=g)
    pri = (torch.rand(n, device="cuda", generator=g, dtype=torch.float64) + 0.05) / n
    for a, sparse in zip(agents, (True, False)):
        with torch.no_grad():
            for layer in a.online.layers:
                layer.b_sigma.fill_(0.05)
                layer.b.fill_(0.03)
        a.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
        a.add_experience((None, (o2, legal)), act, rew, st)
        fl = a._fused_learner()
        fl.refresh_effective()
        assert fl.sparse_backward
        fl.sparse_backward = sparse
        a.experience.sync_size()
        a._beta.fill_(0.4)
        fl.part1(idx, pri)
    s, d = agents[0]._fl, agents[1]._fl
    AK = n_act * 51
    # bf16: the sparse chain reads the thin GEMM's fp32 logits, the dense cross-check chain the same logits rounded to bf16
    # (a near-tie of the double-Q selection may then pick another action, which swaps that sample's whole target: tolerance.py)
    close = torch.ones_like(s.td, dtype=torch.bool)
    if dtype in ("bfloat16", "float16") and s.thin:   # (fp16: the same, 8 x finer)
        close = (s.td - d.td).abs() <= ((1.5e-2 + 2e-3 * d.td.abs()) if dtype == "bfloat16" else (2e-3 + 3e-4 * d.td.abs()))
        assert float(close.float().mean()) >= 0.98 and float((s.td - d.td).abs().max()) < 1.0, float((s.td - d.td).abs().max())
    else:
        assert torch.allclose(s.td, d.td, rtol=1e-5, atol=1e-6)
    assert torch.allclose(s.w_is, d.w_is, rtol=1e-6)
    assert float(s.step) == float(d.step) == 1.0
    # compact gradient == the populated slice of the dense one (dense is rounded to the GEMM dtype)
    ar = torch.arange(n, device="cuda")
    cols = s.act.long()[:, None] * 51 + torch.arange(51, device="cuda")[None]
    dense_slice = d.dlogits.float()[ar[:, None], cols]
    tol = dict(float32=(1e-5, 1e-9), bfloat16=(1e-2, 1e-7), float16=(2e-3, 1e-7))[dtype]
    assert torch.allclose(s.dl[close, :51], dense_slice[close], rtol=tol[0],
                          atol=tol[1] + 2e-3 * float(dense_slice.abs().max()) * (dtype != "float32"))
    assert not s.dl[:, 51:].any()
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    lim = dict(float32=2e-5, bfloat16=2e-2, float16=3e-2)[dtype]        # f16: the dense chain's dlogits go subnormal
    if not bool(close.all()):
        lim = 0.15                                                       # a flipped selection changes one sample's whole gradient
    assert rel(s._gw2_out[:, :AK].float(), d._gw2_out[:, :AK].float()) < lim
    assert rel(s._gb2_pad[:AK], d._gb2_pad[:AK]) < lim
    assert rel(s.g_b1, d.g_b1) < lim
    assert rel(s._gw1_out[:obs_len].float(), d._gw1_out[:obs_len].float()) < lim
    assert not s._gw2_out[:, 3 * 51:4 * 51].any() and not s._gb2_pad[3 * 51:4 * 51].any()      # the action nobody took
    # bit-reproducible: a second run of the sparse path gives identical bits
    keep = [t.clone() for t in (s.dh, s.g_b1, s._gw2_out, s._gb2_pad, s.dl)]
    s.part1(idx, pri)
    for x, y in zip(keep, (s.dh, s.g_b1, s._gw2_out, s._gb2_pad, s.dl)):
        assert torch.equal(x, y)


def test_relu_backward_with_column_sums_and_update_counter():
    """hb_relu_bwd_colsum == aten::threshold_backward + column sums (fp32, fixed order); the loss kernel
    advances the learner's step counter exactly once per update."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip import _capi as K

    g = torch.Generator(device="cuda").manual_seed(3)
    for dt, code in ((torch.bfloat16, 1), (torch.float32, 0), (torch.float16, 2)):
        dy = torch.randn(256, 512, device="cuda", generator=g).to(dt)
        act = torch.relu(torch.randn(256, 512, device="cuda", generator=g)).to(dt)
        want = torch.ops.aten.threshold_backward(dy, act, 0.0)
        out = torch.empty(512, device="cuda")
        got = dy.clone()
        K.check(K.lib().hb_relu_bwd_colsum(K.dptr(got), K.dptr(act), 0, code, 256, 512, K.dptr(out), K.current_stream()))
        assert torch.equal(got, want)
        ref = torch.empty(512, device="cuda")
        K.check(K.lib().hb_colsum(K.dptr(want), code, 256, 512, K.dptr(ref), K.current_stream()))
        assert torch.equal(out, ref) and torch.allclose(out, want.float().sum(0), rtol=1e-4, atol=1e-3)

    n = 64
    params = RlaxRainbowParams(train_batch_size=32, experience_buffer_size=256, layers=[32])
    a = DQNAgent(ObservationSpec((n, 171)), ActionSpec(11), params, device="cuda")
    obs = (torch.rand(n, 171, device="cuda", generator=g) < 0.3).to(torch.int8)
    legal = torch.ones(n, 11, dtype=torch.int8, device="cuda")
    a.add_experience_first((None, (obs, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
    a.add_experience((None, (obs, legal)), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.ones(n, device="cuda"),
                     torch.ones(n, dtype=torch.int8, device="cuda"))
    for _ in range(4):
        a.update()
    # (the graph capture's eager warm-up updates are rolled back: rlax_rainbow.py _capture_update_graphs)
    assert a._fl.step.item() == 4 and a.train_step == 4


@pytest.mark.parametrize("width", ["4", "2", "1"])
def test_adam_multi_vector_path_equals_scalar_kernel(width, monkeypatch):
    """hb_noisy_adam_multi (four / two / one element per thread — HB_ADAM_WIDTH, csrc/learner.hip —, gradient read from a
    padded bf16 GEMM output) is bit-identical to hb_noisy_adam (one element per thread, fp32 gradient) on the same data."""
    import torch

    monkeypatch.setenv("HB_ADAM_WIDTH", width)

    from hanabi_hip import _capi as K

    g = torch.Generator(device="cuda").manual_seed(9)
    rows, cols, ld = 40, 68, 128
    mk = lambda: torch.randn(rows, cols, device="cuda", generator=g)
    base = [mk() * 0.1 for _ in range(3)]                                      # w, w_mu, w_sigma
    noise, grad16 = mk(), torch.zeros(rows, ld, dtype=torch.bfloat16, device="cuda")
    grad16[:, :cols] = (mk() * 0.01).to(torch.bfloat16)
    moms = [torch.rand(rows, cols, device="cuda", generator=g) * 1e-3 for _ in range(6)]
    step = torch.tensor(6.0, device="cuda")
    L, s = K.lib(), K.current_stream()
    outs = []
    for vec in (True, False):
        p = [t.clone() for t in base]
        m = [t.clone() for t in moms]
        eff = torch.zeros(rows, ld, dtype=torch.bfloat16, device="cuda")
        if vec:
            tab = (K.HbAdamTensor * 1)()
            d = tab[0]
            d.w, d.w_mu, d.w_sigma = (t.data_ptr() for t in p)
            d.noise, d.grad, d.grad_dtype, d.grad_ld = noise.data_ptr(), grad16.data_ptr(), 1, ld
            d.m_w, d.v_w, d.m_mu, d.v_mu, d.m_sigma, d.v_sigma = (t.data_ptr() for t in m)
            d.eff, d.n, d.cols, d.eff_ld = eff.data_ptr(), rows * cols, cols, ld
            K.check(L.hb_noisy_adam_multi(tab, 1, K.dptr(step), 1.0, 1, 1e-3, 0.9, 0.999, 3.125e-5, s))
        else:
            g32 = grad16[:, :cols].float().contiguous()
            K.check(L.hb_noisy_adam(K.dptr(p[0]), K.dptr(p[1]), K.dptr(p[2]), K.dptr(noise), K.dptr(g32), K.dptr(m[0]), K.dptr(m[1]),
                                    K.dptr(m[2]), K.dptr(m[3]), K.dptr(m[4]), K.dptr(m[5]), K.dptr(step), K.dptr(eff), 1,
                                    rows * cols, cols, ld, 1e-3, 0.9, 0.999, 3.125e-5, s))
        outs.append(p + m + [eff])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not torch.equal(outs[0][0], base[0])


@pytest.mark.parametrize("n,players,game,atoms,hidden", [(1000, 2, "Hanabi-Full", 51, 512), (300, 5, "Hanabi-Full", 51, 512),
                                                        (700, 3, "Hanabi-Small", 21, 256), (65, 2, "Hanabi-Very-Small", 64, 256)])
def test_mfma_actor_equals_library_actor(n, players, game, atoms, hidden):
    """csrc/actor.hip (hb_actor_hidden + hb_actor_q + hb_policy_select) against cast + hipBLASLt GEMMs + hb_policy_act on
    the same agent: same hidden activations (bf16, bit for bit up to accumulation order), same actions except where
    two q values differ by rounding; ragged row counts (not a multiple of the 256-row tile) and 5-player shapes
    (obs 1280, 48 actions: 10 column groups, the last one partial)."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    cfg = hanabi_hip.make_config(game, players)
    L = K_lib().hb_obs_len(C_byref(cfg))
    A = K_lib().hb_num_actions(C_byref(cfg))
    params = RlaxRainbowParams(train_batch_size=32, experience_buffer_size=4096, layers=[hidden], compute_dtype="bfloat16",
                               n_atoms=atoms)   # 51 atoms: the unrolled epilogue; other counts: the generic one
    agent = DQNAgent(ObservationSpec((n, L)), ActionSpec(A), params, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    obs = (torch.rand(n, L, device="cuda", generator=g) < 0.35).to(torch.int8)
    legal = (torch.rand(n, A, device="cuda", generator=g) < 0.5).to(torch.int8)
    legal[:, 3] = 1
    with torch.no_grad():
        for layer in agent.online.layers:
            layer.b_sigma.fill_(0.05)
            layer.b.normal_(0, 0.1, generator=g)
    fl = agent._fused_learner()
    fl.refresh_effective()
    assert fl.actor is not None
    for eps in (0.0, 0.3):
        agent._draws = 10
        a_new = agent._act_fused(obs, legal, eps).clone()
        q_new, h_new = fl.actor.q.clone(), fl.actor.h.clone()
        agent._draws = 10
        agent.use_mfma_actor = False
        a_old = agent._act_fused(obs, legal, eps)
        agent.use_mfma_actor = True
        # library-path reference values
        (w1, b1), (w2, b2) = fl.eff
        x = torch.zeros(n, fl.Kp, dtype=torch.bfloat16, device="cuda")
        x[:, :L] = obs.to(torch.bfloat16)
        h_old = torch._addmm_activation(b1, x, w1, use_gelu=False)
        # fp32 logits from the same bf16 operands: the MFMA kernel keeps fp32 accumulators and stages them as fp16 (round 3), the
        # library path rounds its logits to bf16
        lg = (h_old.float() @ w2.float() + b2.float())[:, :A * atoms].view(n, A, atoms)
        q_old = (torch.softmax(lg, -1) * agent.atoms[0]).sum(-1) / atoms
        assert (h_new.float() - h_old.float()).abs().max().item() <= 0.02 * h_old.float().abs().max().item()
        assert (h_new != h_old).float().mean().item() < 1e-3
        assert torch.allclose(q_new, q_old, rtol=2e-3, atol=1.5e-3), float((q_new - q_old).abs().max())
        same = (a_new == a_old).float().mean().item()
        assert same > 0.985, same
        bad = (a_new != a_old).nonzero()[:, 0]
        if eps == 0.0 and bad.numel():  # the disagreements are ties at the library path's bf16 logit rounding between two legal moves
            qa = q_old[bad, a_new[bad].long()]
            qb = q_old[bad, a_old[bad].long()]
            assert (qa - qb).abs().max().item() < 2e-2
        assert (legal[torch.arange(n, device="cuda"), a_new.long()] == 1).all()


def test_adam_shared_moments_equal_separate_moments():
    """Passing the same moment arrays for w and w_mu (they see the same gradient, so their moments are equal) gives
    bit-identical parameters and effective weights to separate, equal moment arrays — vector and scalar kernels."""
    import torch

    from hanabi_hip import _capi as K

    g = torch.Generator(device="cuda").manual_seed(4)
    L, s = K.lib(), K.current_stream()
    for rows, cols in ((32, 64), (7, 5)):      # (vector path, scalar path: cols % 4 != 0)
        mk = lambda sc=1.0: torch.randn(rows, cols, device="cuda", generator=g) * sc
        base, noise, grad = [mk(0.1) for _ in range(3)], mk(), mk(0.01)
        m0, v0, ms, vs = mk(1e-3), mk(1e-3).abs(), mk(1e-3), mk(1e-3).abs()
        step = torch.tensor(3.0, device="cuda")
        outs = []
        for shared in (False, True):
            p = [t.clone() for t in base]
            m_w, v_w, m_s, v_s = m0.clone(), v0.clone(), ms.clone(), vs.clone()
            m_mu, v_mu = (m_w, v_w) if shared else (m0.clone(), v0.clone())
            eff = torch.zeros(rows, cols, device="cuda")
            tab = (K.HbAdamTensor * 1)()
            d = tab[0]
            d.w, d.w_mu, d.w_sigma = (t.data_ptr() for t in p)
            d.noise, d.grad, d.grad_dtype, d.grad_ld = noise.data_ptr(), grad.data_ptr(), 0, 0
            d.m_w, d.v_w, d.m_mu, d.v_mu, d.m_sigma, d.v_sigma = (t.data_ptr() for t in (m_w, v_w, m_mu, v_mu, m_s, v_s))
            d.eff, d.n, d.cols, d.eff_ld = eff.data_ptr(), rows * cols, cols, cols
            for _ in range(3):
                K.check(L.hb_noisy_adam_multi(tab, 1, K.dptr(step), 1.0, 0, 1e-3, 0.9, 0.999, 3.125e-5, s))
            outs.append(p + [m_w, v_w, m_mu, v_mu, m_s, v_s, eff])
        for a, b in zip(*outs):
            assert torch.equal(a, b)


def test_vanilla_double_dqn_gpu_equals_cpu_reference():
    """BASELINE config 2 (scalar double-DQN, uniform replay; spec hanabi_agents/rlax_dqn/rlax_dqn.py:170-205): the GPU agent's
    loss, per-sample TD errors, gradient and weights after an update equal the torch-CPU fp32 evaluation of the same
    arithmetic on the same weights and batch (the hand KAT of tests/test_learner.py pins that arithmetic itself)."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, DQNLearning, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 256, 658, 20
    params = RlaxRainbowParams(distributional=False, use_priority=False, train_batch_size=n, experience_buffer_size=n,
                               target_update_period=2, seed=11)
    gpu = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False)
    cpu = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cpu")
    rng = np.random.default_rng(0)
    o1, o2 = (rng.integers(0, 2, (n, obs_len)).astype(np.int8) for _ in range(2))
    legal = np.ones((n, n_act), np.int8)
    act, rew = rng.integers(0, n_act, n), rng.integers(-1, 3, n).astype(np.float32)
    st = rng.integers(1, 3, n)
    for a in (gpu, cpu):
        with torch.no_grad():
            for b in a.online.biases:
                b.fill_(0.05)
            for p, q in zip(a.target.parameters(), a.online.parameters()):
                p.copy_(q * 0.9)
        a.add_experience_first((None, (o1, legal)), np.zeros(n))
        a.add_experience((None, (o2, legal)), act, rew, st)
        a.experience.sample_indices_dev = lambda b, a=a: torch.arange(b, device=a.device)
    # loss / td / gradient of the first update
    outs = []
    for a in (gpu, cpu):
        tr = a.experience.gather_dev(torch.arange(n, device=a.device))
        tr = tr._replace(observation_tm1=tr.observation_tm1.float(), observation_t=tr.observation_t.float())
        loss, td = DQNLearning.loss(a.online, a.target, a.atoms, tr, 0.99, torch.ones(n, dtype=torch.float64, device=a.device), 0.4,
                                    distributional=False)
        g = torch.autograd.grad(loss, list(a.online.parameters()))
        outs.append((loss.detach().cpu(), td.cpu(), torch.cat([x.reshape(-1) for x in g]).cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-3, atol=1e-6 + 1e-4 * float(outs[1][2].abs().max()))
    terminal = st == 2   # the older agent zeroes the bootstrap at terminal states (rlax_dqn.py:178)
    assert terminal.any() and float(outs[1][1].max()) > 0
    for _ in range(3):
        gpu.update()
        cpu.update()
    for p, q in zip(gpu.online.parameters(), cpu.online.parameters()):
        assert torch.allclose(p.cpu(), q, rtol=1e-3, atol=2e-5)
    for p, q in zip(gpu.target.parameters(), cpu.target.parameters()):
        assert torch.allclose(p.cpu(), q, rtol=1e-3, atol=2e-5)


def test_vanilla_fast_actor_matches_torch_policy():
    """bf16 vanilla agent on the GPU: hidden layer on the MFMA kernel (int8 and bit-packed observations), output layer on the
    library GEMM, hb_policy_select — against DQNPolicy.q_values in fp32: q within the bf16 tolerance, arg-max wherever the
    top-2 gap is clear, identical actions for the two observation forms."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, DQNPolicy, ObservationSpec, RlaxRainbowParams, bitpack
    from hanabi_agents.rlax_dqn.tolerance import TOLERANCE

    n, obs_len, n_act = 1000, 658, 20
    params = RlaxRainbowParams(distributional=False, use_priority=False, experience_buffer_size=1024, compute_dtype="bfloat16",
                               epsilon=0.0)
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda")
    assert agent._plain_fast
    g = torch.Generator(device="cuda").manual_seed(4)
    obs = (torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8)
    legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.6).to(torch.int8)
    legal[:, 1] = 1
    with torch.no_grad():
        agent.online.biases[0].normal_(0, 0.1, generator=g)
        agent.online.biases[1].normal_(0, 0.1, generator=g)
    agent._draws = 5
    a8 = agent.exploit((None, (obs, legal))).clone()
    q_fast = agent._last_q.clone()
    agent._draws = 5
    ab = agent.exploit((None, (bitpack.pack(obs), legal)))
    assert torch.equal(a8, ab)
    net32 = lambda x: torch.relu(x @ agent.online.weights[0] + agent.online.biases[0]) @ agent.online.weights[1] + agent.online.biases[1]
    q32 = net32(obs.float())
    assert float((q_fast - q32).abs().max()) < 0.05 * float(q32.abs().max()) + TOLERANCE["bfloat16"]["q_abs"]
    masked = torch.where(legal.bool(), q32, torch.full_like(q32, float("-inf")))
    top2 = masked.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05
    assert clear.float().mean() > 0.3 and torch.equal(a8[clear].long(), masked.argmax(1)[clear])
    assert bool(legal.gather(1, a8.long()[:, None]).all())


def test_vanilla_actor_graph_replay_equals_eager_calls(monkeypatch):
    """Config 2's acting forward as ONE HIP graph (weight refresh from the learner's bf16 operands, hidden-layer kernel, cast, output
    GEMM: DQNAgent._act_plain, captured once the same observation buffer has been seen three times) against the same launches made
    one by one: identical q values and moves, before and after further updates."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams, bitpack

    n, obs_len, n_act = 4096, 658, 20
    params = RlaxRainbowParams(distributional=False, use_priority=False, experience_buffer_size=n, train_batch_size=256,
                               compute_dtype="bfloat16", epsilon=0.0, learning_rate=0.01, packed_obs=True)
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(6)
    o1 = bitpack.pack((torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8))
    o2 = bitpack.pack((torch.rand(n, obs_len, device="cuda", generator=g) < 0.3).to(torch.int8))
    legal = (torch.rand(n, n_act, device="cuda", generator=g) < 0.6).to(torch.int8)
    legal[:, 1] = 1
    act = torch.randint(0, n_act, (n,), device="cuda", generator=g, dtype=torch.int32)
    agent.add_experience_first((None, (o1, legal)), torch.zeros(n, dtype=torch.int8, device="cuda"))
    agent.add_experience((None, (o2, legal)), act, torch.ones(n, device="cuda"), torch.ones(n, dtype=torch.int8, device="cuda"))
    for _ in range(3):
        agent.update()
    assert agent._fv is not None and agent._plain_fast

    def call():
        agent._draws = 7
        a = agent.exploit((None, (o1, legal))).clone()
        return a, agent._last_q.clone()

    for _ in range(4):
        a_g, q_g = call()
    assert agent._plain_actor.graph is not None, "the forward must have been captured by now"
    monkeypatch.setenv("HB_PLAIN_ACT_GRAPH", "0")
    a_e, q_e = call()
    assert torch.equal(q_g, q_e) and torch.equal(a_g, a_e)
    monkeypatch.setenv("HB_PLAIN_ACT_GRAPH", "1")
    for _ in range(2):
        agent.update()
    a_g, q_g = call()              # replayed: must see the new weights
    monkeypatch.setenv("HB_PLAIN_ACT_GRAPH", "0")
    a_e2, q_e2 = call()
    assert torch.equal(q_g, q_e2) and torch.equal(a_g, a_e2)
    assert not torch.equal(q_e, q_e2)


@pytest.mark.parametrize("n,A,atoms,hidden", [(32768, 20, 51, 512), (1000, 20, 51, 512), (300, 48, 51, 512), (700, 11, 21, 256),
                                              (65, 6, 64, 256)])
def test_selection_fused_into_the_q_gemm_equals_the_separate_launch(n, A, atoms, hidden):
    """hb_actor_q_select (the last workgroup of each 256-row tile selects the tile's actions; ticket counters) against
    hb_actor_q + hb_policy_select: identical q values and identical actions, bit for bit, over repeated calls (the counters
    re-arm themselves), ragged row counts and the 5-player action count (10 column groups, the last one partial)."""
    import torch

    from hanabi_hip.ops import ActorMFMA

    L, kp = 200, 256
    g = torch.Generator(device="cuda").manual_seed(n + A)
    act = ActorMFMA(L, hidden, A, atoms, kp, "cuda")
    w1 = (torch.randn(kp, hidden, device="cuda", generator=g) * 0.08).to(torch.bfloat16)
    w2 = (torch.randn(hidden, A * atoms, device="cuda", generator=g) * 0.08).to(torch.bfloat16)
    b1 = (torch.randn(hidden, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    b2 = (torch.randn(A * atoms, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    act.pack(w1, b1, w2, b2)
    support = torch.linspace(-25, 25, atoms, device="cuda")
    for rep in range(4):
        obs = (torch.rand(n, L, device="cuda", generator=g) < 0.35).to(torch.int8)
        legal = (torch.rand(n, A, device="cuda", generator=g) < 0.5).to(torch.int8)
        legal[:, 1] = 1
        eps = 0.0 if rep % 2 == 0 else 0.25
        act.fuse_select = True
        a_f = act.act(obs, legal, support, eps, 77, 100 + rep, 4096).clone()
        q_f = act.q.clone()
        assert int(act.tickets.abs().sum()) == 0, "ticket counters were not re-armed"
        act.fuse_select = False
        a_s = act.act(obs, legal, support, eps, 77, 100 + rep, 4096)
        assert torch.equal(q_f, act.q)
        assert torch.equal(a_f, a_s), f"rep {rep}: {(a_f != a_s).sum().item()} actions differ"
        assert bool((legal.gather(1, a_f.long()[:, None]) == 1).all())


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_vanilla_learner_equals_autograd_learner(dtype):
    """FusedVanillaLearner (gather, 2 GEMMs, hb_dqn_loss_sparse, hb_c51_backward with the scalar head stored as a 2-atom head,
    dW1 GEMM, torch's fused Adam) against the torch-autograd form of the same update (use_fused_learner=False) on the same
    agent state: per-sample td, loss, every gradient and the weights after 5 updates — exact arithmetic up to summation
    order at fp32, within the bf16 tolerances at bf16."""
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_agents.rlax_dqn.tolerance import TOLERANCE

    n, obs_len, n_act = 256, 658, 20
    params = RlaxRainbowParams(distributional=False, use_priority=False, train_batch_size=n, experience_buffer_size=2 * n,
                               target_update_period=2, seed=11, compute_dtype=dtype)
    fused = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False)
    plain = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cuda", use_graphs=False, use_fused_learner=False)
    rng = np.random.default_rng(0)
    o1, o2 = (rng.integers(0, 2, (n, obs_len)).astype(np.int8) for _ in range(2))
    legal = np.ones((n, n_act), np.int8)
    act, rew = rng.integers(0, n_act, n), rng.integers(-1, 3, n).astype(np.float32)
    st = rng.integers(1, 3, n)
    for a in (fused, plain):
        with torch.no_grad():
            for b in a.online.biases:
                b.fill_(0.05)
            for p, q in zip(a.target.parameters(), a.online.parameters()):
                p.copy_(q * 0.9)
        a.add_experience_first((None, (o1, legal)), np.zeros(n))
        a.add_experience((None, (o2, legal)), act, rew, st)
        a.experience.sample_indices_dev = lambda b, a=a: torch.arange(b, device=a.device)
    assert fused._fused_vanilla() is not None and plain._fused_vanilla() is None
    fused._fv.refresh_all()                      # (the biases and the target were changed after construction)
    f32 = dtype == "float32"
    # (bf16 weights after 5 Adam steps: a gradient near zero may take the other sign under rounding and Adam's first steps
    # move every weight by ~lr whatever the gradient's size: the bound of tolerance.py, 2 * 5 * lr)
    tol = dict(td=1e-4, g=2e-3, w=2e-4) if f32 else dict(td=TOLERANCE["bfloat16"]["td_abs"] * 4, g=0.06,
                                                         w=TOLERANCE["bfloat16"]["weights_after_5_steps_max_abs"])
    # first update: td, loss and gradients (the plain agent leaves them in p.grad as well)
    fused._flat_grad.zero_()
    _, _, td_f = fused._update_part1()
    loss_p, _, td_p = plain._update_part1()
    td_f, td_p = td_f.clone(), td_p.clone()
    assert torch.allclose(td_f.abs(), td_p, atol=tol["td"], rtol=1e-3)
    assert abs(float(fused.last_loss) - float(loss_p)) <= 2e-3 * abs(float(loss_p)) + 1e-6
    for p, q in zip(fused.online.parameters(), plain.online.parameters()):
        scale = float(q.grad.abs().max())
        assert float((p.grad - q.grad).abs().max()) <= tol["g"] * scale + 1e-8, (p.shape, float((p.grad - q.grad).abs().max()), scale)
    fused._update_part2(None, None)
    plain._update_part2(None, None)
    for _ in range(4):
        fused.update()
        plain.update()
    for p, q in zip(list(fused.online.parameters()) + list(fused.target.parameters()),
                    list(plain.online.parameters()) + list(plain.target.parameters())):
        assert float((p - q).abs().max()) <= tol["w"] + 1e-3 * float(q.abs().max())
        assert float((p - q).abs().mean()) <= 0.05 * tol["w"] + 1e-6     # ... and on average they agree far better


@pytest.mark.parametrize("m,n,k,batch,relu", [(512, 1024, 704, 1, True), (512, 1024, 512, 2, False), (64, 48, 1280, 1, True),
                                              (512, 2496, 512, 2, False)])
def test_thin_gemm_equals_fp32_reference(m, n, k, batch, relu):
    """hb_thin_gemm (the learner's co-resident forward GEMM: one wavefront per 32 x 16 tile, fragments straight from global memory)
    against an fp32 product of the same bf16 operands: equal up to the bf16 rounding of the output; strided batches (the
    {online, target} pair reads the two halves of one activation matrix), bias + ReLU epilogue."""
    import torch

    from hanabi_hip import _capi as K

    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    ldx = batch * k                                       # batch b reads columns [b*k, (b+1)*k) of ONE row-major matrix
    x = (torch.randn(m, ldx, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    wt = (torch.randn(batch, n, k, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = (torch.randn(n, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    out = torch.full((batch, m, n), 7.0, dtype=torch.bfloat16, device="cuda")
    K.check(K.lib().hb_thin_gemm(K.dptr(x), K.dptr(wt), K.dptr(bias) if relu else None, K.dptr(out), m, n, k, ldx, k, n, batch, k, n * k,
                                 m * n, 1 if relu else 0, K.current_stream()))
    torch.cuda.synchronize()
    for b in range(batch):
        ref = x[:, b * k:(b + 1) * k].float() @ wt[b].float().t()
        if relu:
            ref = torch.relu(ref + bias.float())
        err = (out[b].float() - ref).abs()
        assert float(err.max()) <= 2 ** -8 * float(ref.abs().max()) + 1e-3, (b, float(err.max()))
        assert torch.equal(out[b], ref.to(torch.bfloat16)) or float((out[b] != ref.to(torch.bfloat16)).float().mean()) < 0.02
    # fp32 output (relu bit 1) with one bias row per batch entry: the accumulators + bias as they are (the learner's logits)
    bias2 = (torch.randn(batch, n, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    out32 = torch.full((batch, m, n), 7.0, dtype=torch.float32, device="cuda")
    K.check(K.lib().hb_thin_gemm(K.dptr(x), K.dptr(wt), K.dptr(bias2), K.dptr(out32), m, n, k, ldx, k, n, batch, k, n * k, m * n,
                                 (1 if relu else 0) | 2, K.current_stream()))
    for b in range(batch):
        ref = x[:, b * k:(b + 1) * k].float() @ wt[b].float().t() + bias2[b].float()
        if relu:
            ref = torch.relu(ref)
        assert float((out32[b] - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-5
    with pytest.raises(K.HbError):
        K.check(K.lib().hb_thin_gemm(K.dptr(x), K.dptr(wt), None, K.dptr(out), m + 1, n, k, ldx, k, n, 1, 0, 0, 0, 0, K.current_stream()))
