"""Bit-packed observation path: the env's native bit rows -> replay ring -> actor / learner kernels,
against the reference-shaped int8 path on the same data. Everything here is exact: 0/1 observations lose nothing when packed."""
import numpy as np
import pytest
import torch

from hanabi_agents.rlax_dqn import bitpack


def _np_pack(obs):
    n, L = obs.shape
    w = (L + 31) // 32
    b = np.zeros((n, w * 32), np.uint8)
    b[:, :L] = obs != 0
    return np.packbits(b, axis=1, bitorder="little").view(np.uint32).view(np.int32)


# ---- CPU: torch restatement of the layout, and the packed ring hands out what the int8 ring hands out ----------------------
@pytest.mark.parametrize("L", [658, 1280, 171, 64, 33, 32, 31, 1])
def test_bitpack_torch_roundtrip_and_layout(L):
    g = torch.Generator().manual_seed(L)
    obs = (torch.rand(37, L, generator=g) < 0.4).to(torch.int8)
    bits = bitpack.pack(obs)
    assert bits.dtype == torch.int32 and bits.shape == (37, bitpack.words_for(L))
    assert np.array_equal(bits.numpy(), _np_pack(obs.numpy()))             # element i = bit i & 31 of word i >> 5
    assert torch.equal(bitpack.unpack(bits, L), obs)
    assert bitpack.is_packed(bits, L) and (L <= 1 or not bitpack.is_packed(obs.to(torch.int32), L))


def test_packed_ring_hands_out_the_same_transitions_cpu():
    from hanabi_agents.rlax_dqn.experience_buffer import ExperienceBuffer

    L, A, cap = 171, 11, 7
    rng = np.random.default_rng(0)
    plain = ExperienceBuffer(L, A, 1, cap, device="cpu")
    packed = ExperienceBuffer(L, A, 1, cap, device="cpu", packed=True)
    assert packed._obs_t_buf.dtype == torch.int32 and packed._obs_t_buf.shape == (cap, 6)
    for k in range(3):       # 3 x 4 rows into a ring of 7: wraps (tests/rlax_dqn/test_experience_buffer.py:99-142)
        o1, o2 = (rng.integers(0, 2, (4, L)).astype(np.int8) for _ in range(2))
        rest = (rng.integers(0, A, (4, 1)), rng.random((4, 1)), o2, rng.integers(0, 2, (4, A)), rng.random((4, 1)) < 0.3)
        plain.add_transitions(o1, *rest)
        if k == 1:           # packed rows are accepted as they are
            rest = rest[:2] + (bitpack.pack(torch.as_tensor(o2)),) + rest[3:]
            packed.add_transitions(bitpack.pack(torch.as_tensor(o1)), *rest)
        else:
            packed.add_transitions(o1, *rest)
        assert (plain.oldest_entry, plain.size) == (packed.oldest_entry, packed.size)
    a, b = plain[np.arange(cap)], packed[np.arange(cap)]
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    t1, d1 = plain.gather_nstep_dev(torch.arange(cap), 1, 0.99)
    t2, d2 = packed.gather_nstep_dev(torch.arange(cap), 1, 0.99)
    assert torch.equal(t1.observation_t, t2.observation_t) and torch.equal(d1, d2)


def test_agent_api_accepts_both_forms_cpu():
    """An agent with packed_obs keeps bit rows internally and behaves exactly like the int8 agent (CPU, torch path)."""
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, L, A = 16, 171, 11
    base = RlaxRainbowParams(use_priority=False, train_batch_size=16, experience_buffer_size=16, layers=[16], seed=3)
    agents = [DQNAgent(ObservationSpec((n, L)), ActionSpec(A), base._replace(packed_obs=p), device="cpu") for p in (False, True)]
    rng = np.random.default_rng(1)
    o1, o2 = (rng.integers(0, 2, (n, L)).astype(np.int8) for _ in range(2))
    legal = np.ones((n, A), np.int8)
    act, rew = rng.integers(0, A, n), rng.random(n)
    for a, packed_in in zip(agents, (False, True)):
        f = (lambda o: bitpack.pack(torch.as_tensor(o))) if packed_in else (lambda o: o)
        a.add_experience_first((None, (f(o1), legal)), np.zeros(n))
        a.add_experience((None, (f(o2), legal)), act, rew, np.ones(n))
        a.experience.sample_indices_dev = lambda b: torch.arange(b)
        a.update()
    assert agents[1].last_obs.dtype == torch.int32 and agents[1].experience.packed
    for p, q in zip(agents[0].online.parameters(), agents[1].online.parameters()):
        assert torch.equal(p, q)
    ex = [a.exploit((None, (torch.as_tensor(o2) if i == 0 else bitpack.pack(torch.as_tensor(o2)), torch.as_tensor(legal))))
          for i, a in enumerate(agents)]
    assert torch.equal(ex[0], ex[1])


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,L", [(1000, 658), (257, 1280), (64, 171), (5, 33), (3, 1), (4096, 658)])
def test_hip_pack_unpack_match_numpy(n, L):
    g = torch.Generator(device="cuda").manual_seed(n + L)
    obs = (torch.rand(n, L, device="cuda", generator=g) < 0.35).to(torch.int8)
    obs[0, L - 1] = 1
    bits = bitpack.pack(obs)
    assert np.array_equal(bits.cpu().numpy(), _np_pack(obs.cpu().numpy()))
    out = torch.full((n, L), 7, dtype=torch.int8, device="cuda")
    assert torch.equal(bitpack.unpack(bits, L, out=out), obs)
    obs2 = obs * 5                                         # any nonzero byte packs to 1
    assert torch.equal(bitpack.pack(obs2), bits)


@pytest.mark.gpu
def test_packed_env_full_size_equals_unpacked_env():
    """BASELINE size (32 768 games, 2-player full Hanabi): the packed env and the int8 env run the same games; at every step
    obs_bits == pack(obs) and unpack(obs_bits) == obs (round trip), with everything else identical."""
    import hanabi_hip

    n = 32768
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    envs = [hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=1234, packed=p)
            for p in (False, True)]
    for t in range(120):
        act = envs[0].random_legal_actions(seed=4321, draw=t)
        for e in envs:
            e.step(act)
        if t % 10 == 0 or t == 119:
            assert torch.equal(envs[1].obs_bits, bitpack.pack(envs[0].obs))
            assert torch.equal(envs[1].obs, envs[0].obs) and torch.equal(envs[1].legal, envs[0].legal)
            assert torch.equal(envs[1].export_state(), envs[0].export_state())
            assert torch.equal(envs[1].agent_reward, envs[0].agent_reward)
    assert envs[1].stats() == envs[0].stats() and envs[1].stats()[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("players,n", [(2, 1000), (5, 300), (2, 33)])
def test_actor_hidden_from_bits_is_bit_identical(players, n):
    from hanabi_hip import _capi as K

    L = {2: 658, 5: 1280}[players]
    kp, hidden = (L + 63) // 64 * 64, 512
    g = torch.Generator(device="cuda").manual_seed(players)
    obs = (torch.rand(n, L, device="cuda", generator=g) < 0.3).to(torch.int8)
    bits = bitpack.pack(obs)
    w1t = (torch.randn(hidden, kp, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    w1t[:, L:] = 0
    b1 = torch.randn(hidden, device="cuda", generator=g) * 0.1
    h = [torch.empty(n, hidden, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    lib, s = K.lib(), K.current_stream()
    K.check(lib.hb_actor_hidden(K.dptr(obs), n, L, K.dptr(w1t), kp, K.dptr(b1), hidden, K.dptr(h[0]), s))
    K.check(lib.hb_actor_hidden_packed(K.dptr(bits), n, L, K.dptr(w1t), kp, K.dptr(b1), hidden, K.dptr(h[1]), s))
    assert torch.equal(h[0], h[1])
    ref = torch.relu(obs.float() @ w1t[:, :L].float().t() + b1)
    assert torch.allclose(h[1].float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("n_step", [1, 3])
def test_gather_from_packed_ring_equals_int8_ring(n_step):
    from hanabi_agents.rlax_dqn.experience_buffer import ExperienceBuffer
    from hanabi_hip import _capi as K

    L, A, n, cap, B = 658, 20, 64, 64 * 5, 128
    g = torch.Generator(device="cuda").manual_seed(n_step)
    rings = [ExperienceBuffer(L, A, 1, cap, device="cuda", packed=p) for p in (False, True)]
    for k in range(7):       # wraps
        o1, o2 = ((torch.rand(n, L, device="cuda", generator=g) < 0.4).to(torch.int8) for _ in range(2))
        rest = (torch.randint(0, A, (n, 1), device="cuda", generator=g), torch.randint(-1, 3, (n, 1), device="cuda", generator=g).float())
        tail = (torch.ones(n, A, dtype=torch.int8, device="cuda"), torch.rand(n, 1, device="cuda", generator=g) < 0.2)
        for r in rings:
            r.track_wp = True
            r.add_transitions(o1, *rest, o2, *tail)
            r.sync_size()
    idx = torch.randint(0, cap, (B,), device="cuda", generator=g)
    outs = []
    for r in rings:
        x = torch.zeros(2 * B, 704, dtype=torch.bfloat16, device="cuda")
        act = torch.empty(B, dtype=torch.int32, device="cuda")
        rew, term, disc = (torch.empty(B, device="cuda") for _ in range(3))
        fn = K.lib().hb_replay_gather_packed if r.packed else K.lib().hb_replay_gather
        K.check(fn(K.dptr(r._obs_tm1_buf), K.dptr(r._obs_t_buf), K.dptr(r._act_tm1_buf), K.dptr(r._rew_t_buf),
                   K.dptr(r._terminal_t_buf), K.dptr(idx), B, L, K.dptr(x), 1, 704, K.dptr(act), K.dptr(rew), K.dptr(term),
                   K.dptr(disc), n_step, 0.99, cap, n, K.dptr(r._size_wp), K.current_stream()))
        outs.append((x, act, rew, term, disc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][0][:, :L].sum() > 0 and not outs[0][0][:, L:].any()
    tr = rings[1].gather_dev(idx)
    assert torch.equal(outs[1][0][:B, :L], tr.observation_tm1.to(torch.bfloat16))


@pytest.mark.gpu
def test_selfplay_with_packed_observations_is_identical():
    """Two self-play sessions from the same seeds — int8 observations end to end vs bit-packed end to end (env output, last_obs,
    replay rings, MFMA actor input, learner gather): same actions every step, same replay contents, same weights."""
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n, steps = 512, 30
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    sessions = []
    for packed in (False, True):
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=5, packed=packed)
        params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=n * 8, compute_dtype="bfloat16", mask_terminal=True,
                                   target_update_period=4, packed_obs=packed)
        agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=20 + s), device="cuda")
                  for s in (0, 1)]
        sessions.append(SelfPlaySession(env, agents))
    for t in range(steps):
        for s in sessions:
            s.step()
        assert torch.equal(sessions[0].last_actions[t % 2], sessions[1].last_actions[t % 2]), t
    for s in sessions:
        s.flush()
    a0, a1 = sessions[0].agents[0], sessions[1].agents[0]
    assert a1._fl is not None and a1._fl.actor is not None and a1.experience.packed and a1.last_obs.dtype == torch.int32
    assert sessions[0].grad_steps == sessions[1].grad_steps > 0
    for p, q in zip(a0.online.parameters(), a1.online.parameters()):
        assert torch.equal(p, q)
    idx = np.arange(a0.experience.size)
    for x, y in zip(a0.experience[idx], a1.experience[idx]):
        assert np.array_equal(x, y)
    assert torch.equal(bitpack.unpack(a1.last_obs, a1.obs_len), a0.last_obs)
