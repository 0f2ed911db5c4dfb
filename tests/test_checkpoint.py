"""Checkpoint / resume. The reference stores network parameters only (rlax_rainbow.py:344-365,
kept as save_weights / restore_weights); a checkpoint adds optimizer moments, the replay ring with its sum tree, env
rows and every RNG state, and a resumed run must continue bit-for-bit."""
import numpy as np
import pytest


def _agent(device, **kw):
    import torch

    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    params = RlaxRainbowParams(train_batch_size=16, experience_buffer_size=256, layers=[24], n_atoms=11, atom_vmax=5,
                               use_priority=False, target_update_period=3, **kw)
    torch.manual_seed(0)
    return DQNAgent(ObservationSpec((8, 30)), ActionSpec(6), params, device=device)


def _feed(agent, rng, rounds):
    for r in range(rounds):
        obs = rng.integers(0, 2, (8, 30)).astype(np.int8)
        legal = np.ones((8, 6), np.int8)
        st = np.full(8, 1 if r else 0)
        agent.add_experience_first((None, (obs, legal)), st)
        a = agent.explore((None, (obs, legal)))
        agent.add_experience((None, (obs, legal)), a, rng.random(8).astype(np.float32), st)
        if agent.experience.size >= 16:
            agent.update()


def _digest(agent):
    return [v.detach().cpu().numpy().copy() for v in agent.online.state_dict().values()] + [
        v.detach().cpu().numpy().copy() for v in agent.target.state_dict().values()]


def test_agent_checkpoint_resumes_bit_exact_cpu(tmp_path):
    import torch

    a = _agent("cpu")
    _feed(a, np.random.default_rng(1), 6)
    path = tmp_path / "agent.ckpt"
    a.save_checkpoint(path)
    sd = torch.load(path, map_location="cpu", weights_only=True)     # plain tensors / numbers only
    assert sd["format"].endswith("agent/1") and sd["train_step"] == a.train_step and "optimizer" in sd
    _feed(a, np.random.default_rng(2), 5)
    want, want_step, want_size = _digest(a), a.train_step, a.experience.size

    b = _agent("cpu", seed=99)                                         # different initial weights / RNG streams
    b.load_checkpoint(path)
    _feed(b, np.random.default_rng(2), 5)
    assert b.train_step == want_step and b.experience.size == want_size
    for x, y in zip(want, _digest(b)):
        assert np.array_equal(x, y)
    tr_a, tr_b = a.experience[np.arange(want_size)], b.experience[np.arange(want_size)]
    for x, y in zip(tr_a, tr_b):
        assert np.array_equal(x, y)


def test_weights_only_checkpoint_and_errors(tmp_path):
    a = _agent("cpu")
    _feed(a, np.random.default_rng(1), 4)
    sd = a.checkpoint_state(include_replay=False)
    assert "data" not in sd["experience"]
    b = _agent("cpu", seed=5)
    b.load_checkpoint_state(sd)
    assert b.experience.size == 0 and b.train_step == a.train_step
    for x, y in zip(_digest(a), _digest(b)):
        assert np.array_equal(x, y)
    with pytest.raises(ValueError):
        b.load_checkpoint_state({"format": "something else"})
    sd["experience"]["capacity"] = 128
    with pytest.raises(ValueError):
        b.load_checkpoint_state(sd)


@pytest.mark.gpu
@pytest.mark.parametrize("n_step", [1, 3])
def test_session_checkpoint_resumes_bit_exact(tmp_path, n_step):
    """Self-play + PER + fused learner under HIP graphs and the learner stream: save after 14 steps, run 12 more; a
    fresh process-alike (new env, new agents with other seeds) that loads the file and runs the same 12 steps ends
    with identical weights, Adam moments, replay, sum tree, env rows and episode statistics."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n = 128
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT

    def make(agent_seed):
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=11)
        params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=n * 8, layers=[64], n_step=n_step,
                                   mask_terminal=True, target_update_period=4)
        agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=agent_seed + s),
                           device="cuda") for s in (0, 1)]
        return env, agents, SelfPlaySession(env, agents)

    def digest(env, agents, sess):
        out = [env.export_state().cpu().numpy(), env.obs.cpu().numpy(), np.array([sess.episodes, sess.t, sess.grad_steps])]
        for a in agents:
            out += [v.cpu().numpy() for v in a.online.state_dict().values()]
            out += [v.cpu().numpy() for v in a.target.state_dict().values()]
            out += [t.cpu().numpy() for mv in a._fl.state.values() for t in mv]
            out += [t.float().cpu().numpy() for pair in a._fl.eff for t in pair]
            out += [a.experience.sum_tree.nodes().cpu().numpy(), a.experience._obs_t_buf.cpu().numpy(),
                    a.experience._rew_t_buf.cpu().numpy(), np.array([a.experience.size, a.experience.oldest_entry, a.train_step])]
        return out

    torch.manual_seed(5)
    env, agents, sess = make(1)
    sess.run(14)
    path = tmp_path / "session.ckpt"
    sess.save_checkpoint(path)
    assert agents[0].experience.size == n * 6 and sess.grad_steps > 0
    sess.run(12)
    want = digest(env, agents, sess)
    mean_a = sess.mean_score()

    torch.manual_seed(77)
    env2, agents2, sess2 = make(50)
    sess2.load_checkpoint(path)
    assert sess2.t == 14 and agents2[0]._graph1 is not None      # graphs were captured during the load
    sess2.run(12)
    got = digest(env2, agents2, sess2)
    assert len(want) == len(got)
    for k, (x, y) in enumerate(zip(want, got)):
        assert np.array_equal(x, y), f"item {k} differs after resume"
    assert sess2.mean_score() == mean_a or (np.isnan(mean_a) and np.isnan(sess2.mean_score()))
    with pytest.raises(ValueError):
        env3 = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=12)
        SelfPlaySession(env3, agents2).load_checkpoint(path)
