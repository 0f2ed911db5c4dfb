"""Asynchronous actor / learner: `RlaxRainbowParams.actor_lag = 1`.

The reference's agent is synchronous: explore() always sees the parameters of the last update (rlax_rainbow.py:277-290,
310-339). With actor_lag=1 the actor's packed weights are double-buffered and a policy call reads the weights of the update
BEFORE the most recent one, so the acting stream never waits for the learner. The tests pin that this one-update staleness
is the ONLY difference:
  * the chosen actions and q-values equal those of an independent actor loaded with a snapshot of the weights taken two
    updates ago (and differ from the newest weights' q-values);
  * the run with the learner on its own HIP stream (events only, no waiting for updates) is bit-identical to the same
    lagged run executed on one stream — i.e. the relaxed ordering is race-free: weights, Adam moments, sum tree, rings;
  * a checkpoint taken mid-run resumes bit-for-bit (the lagging weight set is state of its own).
"""
import numpy as np
import pytest


def _make(n, lag, learner_stream, seed=3, game="Hanabi-Small", batch=64, **kw):
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, 2, flags), n_games=n, seed=11, packed=True)
    params = RlaxRainbowParams(train_batch_size=batch, experience_buffer_size=n * 8, layers=[256], mask_terminal=True,
                               target_update_period=4, compute_dtype="bfloat16", packed_obs=True, actor_lag=lag,
                               learning_rate=0.01, **kw)   # (a large step: consecutive weight sets differ visibly in bf16)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=seed + 17 * s),
                       device="cuda") for s in (0, 1)]
    return env, agents, SelfPlaySession(env, agents, learner_stream=learner_stream)


def _digest(env, agents):
    out = [env.export_state().cpu().numpy(), env.obs_bits.cpu().numpy()]
    for a in agents:
        out += [v.float().cpu().numpy() for v in a.online.state_dict().values()]
        out += [v.float().cpu().numpy() for v in a.target.state_dict().values()]
        out += [t.cpu().numpy() for mv in a._fl.state.values() for t in mv]
        out += [t.float().cpu().numpy() for pair in a._fl.eff for t in pair]
        out += [t.float().cpu().numpy() for st in a._fl.actor.sets for t in st]
        out += [a.experience.sum_tree.nodes().cpu().numpy(), a.experience._obs_t_buf.cpu().numpy(),
                a.experience._obs_tm1_buf.cpu().numpy(), a.experience._rew_t_buf.cpu().numpy(),
                np.array([a.experience.size, a.experience.oldest_entry, a.train_step, a._fl.n_packed])]
    return out


def test_actor_lag_is_an_addition_and_needs_the_hip_path():
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    assert RlaxRainbowParams().actor_lag == 0
    with pytest.raises(ValueError):   # no CPU implementation: the product path never falls back
        DQNAgent(ObservationSpec((8, 30)), ActionSpec(6), RlaxRainbowParams(actor_lag=1, layers=[24], use_priority=False), device="cpu")


@pytest.mark.gpu
def test_lagged_policy_acts_on_the_weights_of_the_update_before_last():
    import torch

    from hanabi_hip.ops import ActorMFMA

    n = 256
    env, agents, sess = _make(n, lag=1, learner_stream=False)
    snaps = {}
    for a in agents:
        fl = a._fused_learner()
        snaps[id(a)] = [[t.clone() for pair in fl.eff for t in pair]]          # [0]: the initial weights

        def finish(a=a, orig=a.update_finish):
            orig()
            snaps[id(a)].append([t.clone() for pair in a._fl.eff for t in pair])
        a.update_finish = finish
    probe = None
    checked = differs = 0
    for t in range(16):
        seat = sess.t % 2
        a = agents[seat]
        fl = a._fl
        obs, legal = env.net_obs.clone(), env.legal.clone()
        u = len(snaps[id(a)]) - 1                                            # updates this agent has made so far
        sess.step()
        torch.cuda.synchronize()
        if probe is None:
            probe = ActorMFMA(fl.L, fl.H, fl.A, fl.Kk, fl.Kp, "cuda")
        used = snaps[id(a)][max(u - 1, 0)]                                     # weights after update u - 2 (snapshot index u - 1)
        probe.pack(*used)
        want = probe.act(obs, legal, a.atoms[0].contiguous(), float(a.params.epsilon(0)), a.params.seed + 0x9E3779B9, a._draws,
                         a.first_game_id)
        assert torch.equal(want, sess.last_actions[seat]), f"step {t}: actions are not those of the lagged weights"
        assert torch.equal(probe.q, fl.actor.q), f"step {t}: q-values are not those of the lagged weights"
        checked += 1
        if u >= 1:
            probe.pack(*snaps[id(a)][u])                                       # the newest weights: what actor_lag=0 would use
            probe.act(obs, legal, a.atoms[0].contiguous(), 0.0, 1, 1)
            differs += int(not torch.equal(probe.q, fl.actor.q))
    assert checked == 16 and differs >= 5, "the lag never showed: the test would not notice a synchronous actor"
    assert all(len(v) >= 6 for v in snaps.values())


@pytest.mark.gpu
@pytest.mark.parametrize("game,n", [("Hanabi-Small", 256), ("Hanabi-Full", 2048)])
def test_async_streams_equal_the_lagged_run_on_one_stream(game, n):
    """Same seeds, same lag: learner stream + events (the acting stream waits only for `gathered` / `packed`) against the
    plain sequential order. Any race on the rings, the sum tree or the weight sets would show as a difference."""
    import torch

    runs = []
    for ls in (False, True):
        torch.manual_seed(5)
        env, agents, sess = _make(n, lag=1, learner_stream=ls, game=game, batch=256 if n >= 2048 else 64)
        acts = []
        for _ in range(40):
            sess.step()
            acts.append(sess.last_actions[(sess.t - 1) % 2].clone())
        sess.flush()
        torch.cuda.synchronize()
        assert sess.grad_steps >= 36 and (sess.learner_stream is not None) == ls
        runs.append(([x.cpu().numpy() for x in acts], _digest(env, agents)))
    for k, (x, y) in enumerate(zip(runs[0][0], runs[1][0])):
        assert np.array_equal(x, y), f"actions differ at step {k}"
    for k, (x, y) in enumerate(zip(runs[0][1], runs[1][1])):
        assert np.array_equal(x, y), f"state item {k} differs"


@pytest.mark.gpu
def test_lagged_session_checkpoint_resumes_bit_exact(tmp_path):
    import torch

    torch.manual_seed(5)
    env, agents, sess = _make(128, lag=1, learner_stream=True)
    sess.run(15)
    path = tmp_path / "lag.ckpt"
    sess.save_checkpoint(path)
    sess.run(11)
    torch.cuda.synchronize()
    want = _digest(env, agents)

    torch.manual_seed(9)
    env2, agents2, sess2 = _make(128, lag=1, learner_stream=True, seed=40)
    sess2.load_checkpoint(path)
    sess2.run(11)
    torch.cuda.synchronize()
    for k, (x, y) in enumerate(zip(want, _digest(env2, agents2))):
        assert np.array_equal(x, y), f"item {k} differs after resume"
    # a checkpoint written without the lag does not load into a lagging agent (its second weight set is missing)
    env3, agents3, sess3 = _make(128, lag=0, learner_stream=True)
    sess3.run(6)
    with pytest.raises(ValueError):
        agents2[0].load_checkpoint_state(agents3[0].checkpoint_state())
