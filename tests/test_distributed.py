"""N > 1 path on CPU: world-size-2 gloo processes. Checks the one collective the path has (flat gradient
all-reduce) and the rank-sharded game ids / Philox streams of the env."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    n, obs_len, n_act = 8, 24, 4
    params = RlaxRainbowParams(use_priority=False, train_batch_size=8, experience_buffer_size=8, layers=[8], n_atoms=5,
                               atom_vmax=2, seed=5)  # same seed -> identical initial weights on both ranks
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cpu")
    rng = np.random.default_rng(100 + rank)  # different data per rank (its own shard of games)
    o1, o2 = (rng.integers(0, 2, (n, obs_len)).astype(np.int8) for _ in range(2))
    legal = np.ones((n, n_act), np.int8)
    agent.add_experience_first((None, (o1, legal)), np.zeros(n))
    agent.add_experience((None, (o2, legal)), rng.integers(0, n_act, n), rng.integers(0, 2, n).astype(float), np.ones(n))
    agent.experience.sample_indices_dev = lambda b: torch.arange(b)
    # local gradient before the collective
    from hanabi_agents.rlax_dqn import DQNLearning

    tr = agent.experience.gather_dev(torch.arange(n))
    tr = tr._replace(observation_tm1=tr.observation_tm1.float(), observation_t=tr.observation_t.float())
    loss, _ = DQNLearning.loss(agent.online, agent.target, agent.atoms, tr, 0.99, torch.ones(n, dtype=torch.float64), 0.4)
    g = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, list(agent.online.parameters()))])
    agent.update()
    w = torch.cat([p.detach().reshape(-1) for p in agent.online.parameters()])
    extra = {}
    if rank == 0:
        # a purely local agent inside the distributed job (bench.py's CPU learner baseline lives on rank 0 only): with
        # process_group=False its update must not enter a collective — the other rank never would, and the job would hang
        solo = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cpu", process_group=False)
        solo.add_experience_first((None, (o1, legal)), np.zeros(n))
        solo.add_experience((None, (o2, legal)), rng.integers(0, n_act, n), rng.integers(0, 2, n).astype(float), np.ones(n))
        solo.update()
        extra["solo_world"] = solo._dp_world()
    torch.save({"grad": g, "weights": w, **extra}, os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_keeps_ranks_identical(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    assert not torch.allclose(r[0]["grad"], r[1]["grad"])            # shards differ
    assert torch.equal(r[0]["weights"], r[1]["weights"])              # replicas stay in lock step
    assert r[0]["solo_world"] == 1                                    # (and the rank-0-only local agent finished its update)
    # the applied step is Adam on the MEAN gradient: first Adam step = -lr * sign-ish(g_mean)
    from oracle import learner_oracle as LO

    sys.path.insert(0, os.path.join(ROOT, "hanabi-agents_amd"))
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    params = RlaxRainbowParams(use_priority=False, train_batch_size=8, experience_buffer_size=8, layers=[8], n_atoms=5,
                               atom_vmax=2, seed=5)
    fresh = DQNAgent(ObservationSpec((8, 24)), ActionSpec(4), params, device="cpu")
    w0 = torch.cat([p.detach().reshape(-1) for p in fresh.online.parameters()]).numpy().astype(np.float64)
    gmean = ((r[0]["grad"] + r[1]["grad"]) / 2).numpy().astype(np.float64)
    want, _, _ = LO.adam_step(w0, gmean, 0.0, 0.0, 1)
    assert np.allclose(r[0]["weights"].numpy(), want, atol=1e-6)


def _is_max_worker(rank, world, port, out):
    """params.global_is_max: each rank holds 8 transitions with ITS OWN sampling probabilities; the applied gradient must
    be the one a single process computes on the 16-row global batch with w /= max over all 16 (rlax_rainbow.py:188-189)."""
    for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, DQNLearning, ObservationSpec, RlaxRainbowParams
    from hanabi_agents.rlax_dqn import learning as L

    n, obs_len, n_act = 8, 24, 4
    params = RlaxRainbowParams(use_priority=False, train_batch_size=8, experience_buffer_size=8, layers=[8], n_atoms=5,
                               atom_vmax=2, seed=5, global_is_max=True)
    agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_act), params, device="cpu")
    data = []
    for r in range(world):   # every rank builds both shards: its own to train on, all of them for the single-process reference
        rng = np.random.default_rng(100 + r)
        o1, o2 = (rng.integers(0, 2, (n, obs_len)).astype(np.int8) for _ in range(2))
        data.append((o1, o2, rng.integers(0, n_act, n), rng.integers(0, 2, n).astype(float), (rng.random(n) + 0.05) / (1 + 3 * r)))
    o1, o2, act, rew, pri = data[rank]
    legal = np.ones((n, n_act), np.int8)
    agent.add_experience_first((None, (o1, legal)), np.zeros(n))
    agent.add_experience((None, (o2, legal)), act, rew, np.ones(n))
    agent._sample_indices = lambda: (torch.arange(n), torch.as_tensor(pri))
    # single-process reference on the global batch: w = (1/P)^beta / max over ALL rows; mean over 16 rows
    w_all = torch.cat([(1.0 / torch.as_tensor(d[4])).float() ** 0.4 for d in data])
    w_all = w_all / w_all.max()
    total = 0.0
    for r, d in enumerate(data):
        from hanabi_agents.rlax_dqn.transition import Transition

        tr = Transition(torch.as_tensor(d[0]).float(), torch.as_tensor(d[2])[:, None], torch.as_tensor(d[3])[:, None].float(),
                        torch.as_tensor(d[1]).float(), torch.ones(n, n_act), torch.zeros(n, 1, dtype=torch.bool))
        a, k = agent.atoms.shape
        td = L.categorical_double_q_td(agent.online(tr.observation_tm1).view(-1, a, k), tr.action_tm1[:, 0].long(),
                                       tr.reward_t[:, 0], 0.99, agent.atoms, agent.target(tr.observation_t).view(-1, a, k),
                                       agent.online(tr.observation_t).view(-1, a, k).detach())
        total = total + torch.sum(td * w_all[r * n:(r + 1) * n])
    g_ref = torch.cat([x.reshape(-1) for x in torch.autograd.grad(total / (world * n), list(agent.online.parameters()))])
    agent.experience.sync_size()
    agent._beta.fill_(0.4)
    agent._update_part1()
    agent._finish_allreduce(agent._allreduce_gradients(async_op=False))
    torch.save({"grad": agent._flat_grad.clone(), "ref": g_ref, "scale": (agent._is_max_local / agent._is_max_global).clone()},
               os.path.join(out, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_global_is_weight_normalisation_matches_single_global_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_is_max_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"m{i}.pt") for i in range(world)]
    assert torch.equal(r[0]["grad"], r[1]["grad"])
    assert torch.allclose(r[0]["grad"], r[0]["ref"], rtol=1e-4, atol=1e-7)
    scales = sorted(float(x["scale"]) for x in r)
    assert scales[1] == 1.0 and scales[0] < 0.9     # one rank held the global max, the other was rescaled


def test_rank_shards_use_disjoint_reproducible_decks():
    """Games are sharded by global id: rank r owns [r*N, (r+1)*N). The decks a rank deals are exactly the slice
    a single process would deal for those ids (checked on the oracle; the HIP side is compared to the oracle with
    first_game_id != 0 in tests/test_hip_env.py)."""
    from oracle import oracle_py as O

    cfg = O.make_config("Hanabi-Full", 2)
    whole = O.OracleEnv(cfg, 64, seed=1234, first_game_id=0).export_state()
    for rank in range(2):
        part = O.OracleEnv(cfg, 32, seed=1234, first_game_id=rank * 32).export_state()
        assert np.array_equal(part, whole[rank * 32:(rank + 1) * 32])
    assert len({bytes(r[16:29].tobytes()) for r in whole}) == 64   # all decks differ


def _gpu_worker(rank, world, port, out, lag=0):
    """Full self-play + fused learner (packed all-reduce bucket, HIP graphs around the collective, learner stream, MFMA
    actor) with two ranks sharing ONE GPU; gloo moves the CUDA gradient bucket (RCCL needs one GPU per rank)."""
    for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    torch.cuda.set_device(0)
    n = 128
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=9, first_game_id=rank * n,
                               packed=bool(lag))
    params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=n * 8, layers=[256], compute_dtype="bfloat16",
                               mask_terminal=True, target_update_period=5, actor_lag=lag, packed_obs=bool(lag))
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=40 + s), device="cuda")
              for s in (0, 1)]
    for a in agents:
        a.first_game_id = rank * n
    sess = SelfPlaySession(env, agents)
    assert sess.overlap_allreduce or sess.learner_stream is not None
    w0 = torch.cat([p.detach().reshape(-1) for p in agents[0].online.parameters()]).cpu()
    sess.run(24)
    torch.cuda.synchronize()
    fl = agents[0]._fl
    assert fl is not None and not fl.direct and fl.actor is not None and agents[0]._graph1 is not None
    w = [torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]).cpu() for a in agents]
    eff = torch.cat([t.float().reshape(-1) for pair in fl.eff for t in pair]).cpu()
    rows = env.export_state().cpu()
    torch.save({"w": w, "w0": w0, "eff": eff, "rows": rows, "grad_steps": sess.grad_steps, "illegal": env.illegal_count()},
               os.path.join(out, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [0, 1])
def test_two_ranks_on_one_gpu_stay_in_lock_step(tmp_path, lag):
    """lag = 1: the asynchronous actor (double-buffered weights, split update, per-agent learner streams) under data
    parallelism: the replicas still end bit-identical."""
    world, port = 2, _free_port()
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path), lag), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"g{i}.pt") for i in range(world)]
    assert r[0]["grad_steps"] == r[1]["grad_steps"] > 0 and r[0]["illegal"] == r[1]["illegal"] == 0
    assert not torch.equal(r[0]["rows"], r[1]["rows"])                     # different games on each rank (Philox by global id)
    for a, b in zip(r[0]["w"], r[1]["w"]):
        assert torch.equal(a, b)                                           # averaged gradients -> identical replicas
    assert torch.equal(r[0]["eff"], r[1]["eff"]) and not torch.equal(r[0]["w"][0], r[0]["w0"])


def _rccl_worker(rank, world, port, out, lag):
    """ONE rank, backend nccl (= RCCL): the multi-rank update path — flat gradient bucket, HIP graphs captured around the
    collective while RCCL's watchdog thread is alive, asynchronous all-reduce issued from the learner streams — on real RCCL."""
    for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n = 256
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    res = {}
    for force in (False, True, "graph"):
        # "graph": the all-reduce captured inside the update graph (HB_DP_GRAPH_COLLECTIVE=1, opt-in)
        os.environ["HB_DP_GRAPH_COLLECTIVE"] = "1" if force == "graph" else "0"
        torch.manual_seed(0)
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=9, packed=True)
        params = RlaxRainbowParams(train_batch_size=64, experience_buffer_size=n * 8, layers=[256], compute_dtype="bfloat16",
                                   mask_terminal=True, target_update_period=5, actor_lag=lag, packed_obs=True)
        agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=40 + s),
                           device="cuda") for s in (0, 1)]
        for a in agents:
            a.force_collective = bool(force)
        sess = SelfPlaySession(env, agents)
        sess.run(30)
        torch.cuda.synchronize()
        fl = agents[0]._fl
        assert fl.direct == (not force) and sess.grad_steps >= 26
        # two graphs: around the collective (force), or the early-update form of a synchronous split-update agent
        # (force == "graph": the collective is captured INSIDE the one graph)
        want2 = force is True or (force is False and bool(getattr(agents[0], "two_graphs", False) and not agents[0].actor_lag))
        assert (agents[0]._graph2 is not None) == want2, (force, lag, agents[0]._graph2 is not None, agents[0].two_graphs,
                                                          agents[0].split_update, sess.early_update)
        res[force] = dict(w=[torch.cat([p.detach().reshape(-1) for p in a.online.parameters()]).cpu() for a in agents],
                          loss=float(agents[0].last_loss), illegal=env.illegal_count())
    torch.save(res, os.path.join(out, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("lag", [0, 1])
def test_collective_path_runs_on_rccl_with_one_rank(tmp_path, lag):
    """The data-parallel code path cannot be run with two RCCL ranks on a one-GPU box (one GPU per rank); with ONE rank it can:
    same launches, same graphs around the same (here trivial) collectives. Its result must agree with the direct path: the only
    difference is the gradient's route (bf16 GEMM outputs read by Adam directly vs packed into the fp32 bucket first)."""
    port = _free_port()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path), lag), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt")
    assert r[False]["illegal"] == r[True]["illegal"] == 0 and np.isfinite(r[True]["loss"])
    for a, b, c in zip(r[False]["w"], r[True]["w"], r["graph"]["w"]):
        assert torch.isfinite(b).all()
        # same trajectory up to the rounding of the gradient route (30 steps of lr 1e-3: weights move by ~1e-2)
        assert (a - b).abs().max().item() < 2e-2
        assert torch.equal(b, c)        # the collective captured inside the graph: same arithmetic, same order
