"""End-to-end on the GPU: the self-play driver feeds two DQNAgents from the HIP env; transitions that reach the
replay are consistent with an independent replay of the same action sequence on the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap,lstream", [(False, False), (True, False), (False, True)])
def test_selfplay_transitions_match_oracle_replay(overlap, lstream):
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession
    from oracle import oracle_py as O

    n, steps = 96, 40
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=3)
    orc = O.OracleEnv(O.make_config("Hanabi-Small", 2, flags), n, seed=3)
    params = RlaxRainbowParams(train_batch_size=32, experience_buffer_size=n * steps, layers=[32], mask_terminal=True)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
              for s in (1, 2)]
    sess = SelfPlaySession(env, agents, updates_per_step=1, overlap_allreduce=overlap, learner_stream=lstream)  # overlap: update_finish one step later
    # oracle-side bookkeeping of what each seat's replay must contain
    last_obs = [None, None]
    last_act = [None, None]
    want = [[], []]
    out = orc.observe()
    for t in range(steps):
        seat = t % 2
        if last_obs[seat] is not None:
            want[seat].append((last_obs[seat], last_act[seat], out["agent_reward"].copy(), out["obs"].copy(),
                               out["legal"].copy(), out["agent_step_type"] == 2))
            assert (out["agent_step_type"] != 0).all()
        else:
            assert (out["agent_step_type"] == 0).all()
        last_obs[seat] = out["obs"].copy()
        sess.step()
        a = sess.last_actions[seat].cpu().numpy()
        legal_now = out["legal"]
        assert all(legal_now[g, a[g]] == 1 for g in range(n)), "agent picked an illegal move"
        last_act[seat] = a.copy()
        out = orc.step(a)
        assert np.array_equal(env.obs.cpu().numpy(), out["obs"])
    sess.flush()
    assert env.illegal_count() == 0 and all(a._pending is None for a in agents)
    for seat in (0, 1):
        buf = agents[seat].experience
        k = len(want[seat])
        assert buf.size == k * n
        tr = buf[np.arange(buf.size)]
        for j, (o1, act, rew, o2, lg, term) in enumerate(want[seat]):
            sl = slice(j * n, (j + 1) * n)
            assert np.array_equal(tr.observation_tm1[sl], o1) and np.array_equal(tr.observation_t[sl], o2)
            assert np.array_equal(tr.action_tm1[sl, 0], act) and np.array_equal(tr.reward_t[sl, 0], rew)
            assert np.array_equal(tr.legal_moves_t[sl], lg) and np.array_equal(tr.terminal_t[sl, 0], term)
    assert sess.grad_steps > 0 and torch.isfinite(agents[0].last_loss).item()
    assert sess.episodes > 0


def test_graft_smoke():
    import __graft_entry__ as g

    g.smoke()


@pytest.mark.parametrize("game,players,kwargs", [
    ("Hanabi-Full", 5, dict()),                                            # BASELINE config 4 shape: obs 1280, 48 moves
    ("Hanabi-Full", 2, dict(distributional=False, use_priority=False)),    # BASELINE config 2: vanilla double-DQN, uniform replay
    ("Hanabi-Small", 3, dict(compute_dtype="bfloat16")),
])
def test_other_configs_run_end_to_end(game, players, kwargs):
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n = 512
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players, flags), n_games=n, seed=11)
    params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=8192, mask_terminal=True, **kwargs)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
              for s in range(players)]
    sess = SelfPlaySession(env, agents, updates_per_step=1)
    sess.run(6 * players)
    assert env.illegal_count() == 0
    assert sess.grad_steps > 0
    for a in agents:
        assert torch.isfinite(a.last_loss).item()
        assert a.experience.size > 0
    assert int(env.current_player()[0]) == (6 * players) % players and bool((env.current_player() == env.current_player()[0]).all())


@pytest.mark.parametrize("dtype,graphs", [("float32", False), ("bfloat16", True)])
def test_learner_stream_gives_identical_training(dtype, graphs):
    """Running the updates on the second stream keeps every dependency of the sequential order: two sessions
    with the same seeds end with identical weights and replay contents. The bfloat16 case acts through the MFMA
    actor kernels, whose packed weights are rewritten by the learner stream inside its captured graph."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    def run(lstream):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=256, seed=5)
        params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=4096, mask_terminal=True, target_update_period=7,
                                   compute_dtype=dtype)
        agents = [DQNAgent(ObservationSpec((256, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda",
                           use_graphs=graphs) for s in (1, 2)]
        sess = SelfPlaySession(env, agents, learner_stream=lstream)
        sess.run(40)
        torch.cuda.synchronize()
        assert (agents[0]._fl.actor is not None) == (dtype == "bfloat16")
        return [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]) for a in agents], env.export_state()

    (w_a, st_a), (w_b, st_b) = run(False), run(True)
    assert torch.equal(st_a, st_b)
    for x, y in zip(w_a, w_b):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [0, 1])
def test_selection_fused_into_the_env_step_gives_identical_training(lag):
    """SelfPlaySession(fuse_select=True) (the env kernel picks the moves from the agent's q values) against the agent's own
    explore() + env.step(): same actions every step, identical weights, replay rings, sum trees and env rows at the end."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    def run(fuse):
        torch.manual_seed(0)
        flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=512, seed=5, packed=True)
        params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=8192, mask_terminal=True, target_update_period=7,
                                   compute_dtype="bfloat16", packed_obs=True, layers=[256], actor_lag=lag, learning_rate=0.01)
        agents = [DQNAgent(ObservationSpec((512, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
                  for s in (1, 2)]
        sess = SelfPlaySession(env, agents, fuse_select=fuse)
        assert sess.fuse_select == fuse
        acts = []
        for _ in range(30):
            sess.step()
            acts.append(sess.last_actions[(sess.t - 1) % 2].clone())
        sess.flush()
        torch.cuda.synchronize()
        out = [env.export_state(), env.obs_bits.clone()]
        for a in agents:
            out += [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]), a.experience.sum_tree.nodes(),
                    a.experience._obs_t_buf.clone(), a.experience._act_tm1_buf.clone()]
        return acts, out

    (acts_a, out_a), (acts_b, out_b) = run(False), run(True)
    for k, (x, y) in enumerate(zip(acts_a, acts_b)):
        assert torch.equal(x, y), f"actions differ at step {k}"
    for k, (x, y) in enumerate(zip(out_a, out_b)):
        assert torch.equal(x, y), f"item {k} differs"


@pytest.mark.parametrize("lag,dtype", [(0, "bfloat16"), (1, "bfloat16"), (0, "float16"), (1, "float16")])
def test_benched_wiring_matches_oracle_replay(lag, dtype, monkeypatch):
    """The configuration bench.py times — Hanabi-Full 2 players, bit-packed observation rows end to end, [512] net in bf16 on the
    one-kernel actor (selection inside the kernel), split update on per-agent learner streams, HIP graphs, one host call per step
    (hb_chain_run) — replayed move by move on the CPU oracle: every transition that reached the two replay rings equals the
    oracle's (observations, legal masks, per-seat rewards, terminals), synchronous actor and actor_lag = 1;
    the same with fp16 operands (bench.py --compute-dtype float16: the reference's own network dtype on the same kernels)."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession
    from oracle import oracle_py as O

    monkeypatch.setenv("HB_ACTOR_FUSED_MIN_ROWS", "0")     # (the one-kernel actor at this test's 2 048 rows)
    n, steps = 2048, 26
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=n, seed=3, packed=True)
    orc = O.OracleEnv(O.make_config("Hanabi-Full", 2, flags), n, seed=3)
    params = RlaxRainbowParams(train_batch_size=256, experience_buffer_size=n * 16, layers=[512], mask_terminal=True,
                               compute_dtype=dtype, packed_obs=True, actor_lag=lag, target_update_period=5)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
              for s in (1, 2)]
    sess = SelfPlaySession(env, agents)      # default options: what bench.py runs
    assert sess.fuse_select and sess.learner_stream is not None
    last_obs, last_act, want = [None, None], [None, None], [[], []]
    out = orc.observe()
    for t in range(steps):
        seat = t % 2
        if last_obs[seat] is not None:
            want[seat].append((last_obs[seat], last_act[seat], out["agent_reward"].copy(), out["obs"].copy(), out["legal"].copy(),
                               out["agent_step_type"] == 2))
        last_obs[seat] = out["obs"].copy()
        sess.step()
        a = sess.last_actions[seat].cpu().numpy()
        assert (out["legal"][np.arange(n), a] == 1).all(), "agent picked an illegal move"
        last_act[seat] = a.copy()
        out = orc.step(a)
        assert np.array_equal(env.obs.cpu().numpy(), out["obs"])
    sess.flush()
    torch.cuda.synchronize()
    fl = agents[0]._fl
    assert fl.actor is not None and fl.actor.fused and all(a.split_update for a in agents) and sess._stream_per_agent
    assert agents[0]._graph1 is not None and sess.grad_steps >= steps - 4
    assert sess.native_steps >= steps - 12, sess.native_steps     # most steps went through hb_chain_run
    assert env.illegal_count() == 0
    for seat in (0, 1):
        buf = agents[seat].experience
        k = len(want[seat])
        assert buf.size == k * n
        tr = buf[np.arange(buf.size)]
        for j, (o1, act, rew, o2, lg, term) in enumerate(want[seat]):
            sl = slice(j * n, (j + 1) * n)
            assert np.array_equal(tr.observation_tm1[sl], o1) and np.array_equal(tr.observation_t[sl], o2)
            assert np.array_equal(tr.action_tm1[sl, 0], act) and np.array_equal(tr.reward_t[sl, 0], rew)
            assert np.array_equal(tr.legal_moves_t[sl], lg) and np.array_equal(tr.terminal_t[sl, 0], term)


@pytest.mark.parametrize("lag,dtype", [(0, "bfloat16"), (1, "bfloat16"), (0, "float16")])
def test_one_host_call_per_step_gives_identical_training(lag, dtype, monkeypatch):
    """SelfPlaySession(native_chain=True): the step replayed from an hb_cmd array by ONE hb_chain_run call (csrc/chain.hip) against
    the ordinary path's ~25 host calls: same moves every step, identical weights, moments, replay rings, sum trees and env rows."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    monkeypatch.setenv("HB_ACTOR_FUSED_MIN_ROWS", "0")

    def run(native, n_step):
        torch.manual_seed(0)
        flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=1024, seed=5, packed=True)
        params = RlaxRainbowParams(train_batch_size=128, experience_buffer_size=1024 * 8, mask_terminal=True, target_update_period=6,
                                   compute_dtype=dtype, packed_obs=True, layers=[512], learning_rate=0.01, n_step=n_step,
                                   actor_lag=lag)
        agents = [DQNAgent(ObservationSpec((1024, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
                  for s in (1, 2)]
        sess = SelfPlaySession(env, agents, native_chain=native)
        acts = []
        for _ in range(44):          # (the ring wraps: 8 inserts per agent fill it)
            sess.step()
            acts.append(sess.last_actions[(sess.t - 1) % 2].clone())
        sess.flush()
        torch.cuda.synchronize()
        assert (sess.native_steps > 18) == native, sess.native_steps
        out = [env.export_state(), env.obs_bits.clone()]
        for a in agents:
            out += [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]), a.experience.sum_tree.nodes(),
                    a.experience._obs_t_buf.clone(), a.experience._act_tm1_buf.clone(), a._fl.step.clone()]
            out += [t.clone() for pair in a._fl.state.values() for t in pair]
        return acts, out, sess.grad_steps

    for n_step in (1, 3):
        (acts_a, out_a, g_a), (acts_b, out_b, g_b) = run(False, n_step), run(True, n_step)
        assert g_a == g_b
        for k, (x, y) in enumerate(zip(acts_a, acts_b)):
            assert torch.equal(x, y), f"n_step {n_step}: actions differ at step {k}"
        for k, (x, y) in enumerate(zip(out_a, out_b)):
            assert torch.equal(x, y), f"n_step {n_step}: item {k} differs"


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_self_play_learns_hanabi_small(dtype):
    """End to end, everything on the GPU (scripts/train_small.py in short): two agents in self-play on 2-player Hanabi-Small, 2 048
    games, 4 updates per step through the HIP graphs — the mean episode score over the last 500 of 4 000 steps rises from 0.0
    (random play bombs out) to ~3.5 (measured: bf16 3.51, fp16 3.4); required > 2. Not a parity claim: a wiring check that the
    policy acts on what the learner learns (bf16 at this batch size takes the two-kernel actor form, whose weight copies once went
    stale under graph replay: DESIGN section 8; fp16 takes the one-kernel form)."""
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n = 2048
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=1, packed=True)
    params = RlaxRainbowParams(compute_dtype=dtype, mask_terminal=True, experience_buffer_size=2 ** 18, learning_rate=2.5e-4,
                               epsilon=lambda ts: max(0.02, 1.0 - ts / 3000.0), target_update_period=200, atom_vmax=10, packed_obs=True)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=s), device="cuda")
              for s in (1, 2)]
    sess = SelfPlaySession(env, agents, updates_per_step=4)
    sess.run(3500)
    ep0, sc0 = env.stats()
    sess.run(500)
    ep1, sc1 = env.stats()
    assert agents[0]._graph1 is not None
    mean = (sc1 - sc0) / max(1, ep1 - ep0)
    assert mean > 2.0, mean
