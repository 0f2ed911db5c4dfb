#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native Hanabi self-play + Rainbow-DQN path.

    python bench.py --gpus N --steps K --warmup W
    N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or started plainly: `python bench.py --gpus N` with
    no WORLD_SIZE set starts its N ranks itself as child processes (one per GPU, RCCL over xGMI) BEFORE anything in the
    parent touches the GPU; rank 0 prints the JSON line, a failed rank makes the parent exit non-zero.

Workload (BASELINE.json configs[2]): 2-player full Hanabi, 32 768 parallel games PER GPU (weak
scaling: games shard embarrassingly, global game ids = rank * 32768 + local), two Rainbow agents
(C51 + noisy nets + PER on the GPU sum tree, batch 256), one per seat. A "step" is one lock-step
move of all the rank's games through the whole hot path:

    replay insert of the acting seat's transitions -> policy forward + legal eps-greedy sample
    -> fused env step / legal mask / canonical encoder kernel -> `--updates-per-step` learner
    updates (PER sample, 3 forwards + backward, Adam, priority update, RCCL grad all-reduce)

`value` = env-steps/s over all ranks inside the timed region; grad-steps/s is reported next to it.
Extra objects on the same JSON line:
  roofline      the env kernel (the north-star HBM-bound kernel): ALGORITHMIC bytes per launch
                (obs_len + A + 9 + 2*STATE_BYTES = 943 B per game, the reference's int8
                interface) / its average dispatch duration, measured live on every launch of the timed
                region with HIP events attached to the dispatch (hb_env_set_profile_events); the bytes
                the bit-packed kernel physically moves (369 B per game) and their fraction of the peak
                are the side key `physical`, the PMC traffic of the same kernel form is `traffic`
  roofline_qnet MFMA side: algorithmic FLOPs of actor forward + learner step / their event time
  cpu_baseline  the CPU oracle (oracle/, a port: the reference's env is not in its tree) on the
                host cores, bounded sample of the same workload, rank 0 only
"""
# HIP maps its streams onto at most GPU_MAX_HW_QUEUES hardware queues (ROCm default 4). The step interleaves an acting stream, one
# learner stream per agent and, data-parallel, RCCL's stream; which of them share a hardware queue changes the step time by up to
# 3x (measured r02, one MI355X: multi-rank update path 0.47 ms per step with 4 queues, 0.17 ms with 2; single-rank path 0.144 ms
# with either; 3 or 8 queues are pathological for one path or the other: profiles/r02/hw_queues.txt). Must be set before the HIP
# runtime initialises; an explicit setting in the environment wins.
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hanabi-agents_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"float32": 157.3, "bfloat16": 2500.0, "float16": 2500.0}


def parse():
    args = _parse()
    if args.games is None:
        args.games = 4096 if args.vanilla else 32768
    return args


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--actor-lag", type=int, default=0, choices=(0, 1),
                    help="1: asynchronous actor (RlaxRainbowParams.actor_lag): double-buffered actor weights, the acting stream "
                         "never waits for an update; 0 (default): the reference's synchronous semantics")
    ap.add_argument("--main-priority", type=int, default=0, help="HIP priority of the acting stream (-1: high; default: the default stream)")
    ap.add_argument("--event-every", type=int, default=1, help="attach the HIP timing events to every n-th env launch of the timed region")
    ap.add_argument("--no-async-variant", action="store_true", help="skip the second, asynchronous-actor measurement")
    ap.add_argument("--prime", type=int, default=24,
                    help="untimed SETUP steps before the W warm-up steps: the first updates capture the HIP graphs, pick the "
                         "hipBLASLt algorithms and grow the allocator pools (one-time work, the counterpart of a compile "
                         "step); reported as prime_steps")
    ap.add_argument("--games", type=int, default=None, help="games per GPU (default 32 768; 4 096 with --vanilla)")
    ap.add_argument("--vanilla", action="store_true",
                    help="BASELINE configs[1]: vanilla double-DQN (scalar Q head, rlax_dqn.py:170-205 spec) with uniform replay, "
                         "4 096 games per GPU")
    ap.add_argument("--players", type=int, default=2)
    ap.add_argument("--n-step", type=int, default=1, help="n-step returns assembled at sample time (SURVEY 8(f)-2; the reference's agent is 1-step)")
    ap.add_argument("--no-nstep-variant", action="store_true", help="skip the second synchronous measurement with n_step = 3")
    ap.add_argument("--no-fp16-variant", action="store_true", help="skip the synchronous measurement with fp16 GEMM operands")
    ap.add_argument("--updates-per-step", type=int, default=1)
    ap.add_argument("--compute-dtype", default="bfloat16", choices=list(MFMA_PEAK_TFLOPS))
    ap.add_argument("--games-per-wave", type=int, default=None, help="8/16/32/64; default: the library's choice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-games", type=int, default=4096)
    ap.add_argument("--cpu-sample-steps", type=int, default=150)
    ap.add_argument("--env-only", action="store_true", help="random-legal policy, no agents (kernel-only rate)")
    ap.add_argument("--learner-cus", type=int, default=0,
                    help="reserve this many CUs for the learner stream (CU-masked streams); 0 = no partition")
    ap.add_argument("--no-learner-stream", action="store_true", help="run the updates in order on the main stream")
    ap.add_argument("--learner-priority", type=int, default=-1, help="HIP stream priority of the learner stream (-1 = high)")
    ap.add_argument("--stream-per-agent", type=int, default=-1, help="1: one learner stream per agent, 0: one shared, -1: default")
    ap.add_argument("--unpacked-obs", action="store_true",
                    help="int8 [N, obs_len] observations end to end (the reference's layout) instead of the bit-packed rows")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rendezvous the ranks and all-reduce one number (works without a GPU: gloo); tests the launcher")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as children. The parent never initialises
    the GPU (no torch.cuda call happens before this point) and never re-execs; it waits, and if a rank fails it stops the
    others (exact PIDs) and exits with that rank's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:      # one rank died: the others would hang in the next collective
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc if rc >= 0 else 1


def launch_check(rank, world):
    """Rendezvous + one all-reduce, no hot path: what tests/test_bench_launch.py runs on a CPU-only box."""
    backend = os.environ.get("HB_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("HB_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "backend": backend, "sum": float(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def usable_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(args):
    """Oracle env (step + legal + encode) on the host: same game config, Philox decks and random-legal
    policy; 1 thread and all cores. Bounded to ~10-30 s."""
    from oracle import oracle_py as O

    n, steps = args.cpu_sample_games, args.cpu_sample_steps
    flags = O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT
    cores = usable_cores()
    out = {}
    for label, threads in (("1", 1), ("all", cores)):
        env = O.OracleEnv(O.make_config("Hanabi-Full", args.players, flags), n, seed=1234, threads=threads)
        legal = env.observe()["legal"]
        t0 = time.perf_counter()
        for t in range(steps):
            act = O.random_legal_actions(legal, 4321, t)
            legal = env.step(act)["legal"]
        out[label] = n * steps / (time.perf_counter() - t0)
    best = max(out.values())
    return {"value": best, "unit": "env-steps/s", "cores": cores if out["all"] >= out["1"] else 1, "kind": "port",
            "single_thread_value": out["1"], "all_cores_value": out["all"],
            "sample": f"{n} games x {steps} lock-step moves of the same 2-player full Hanabi workload "
                      f"(oracle/hanabi_oracle.c: step + legal mask + canonical encoder, random-legal policy)"}


def cpu_baseline_learner(args, obs_len, n_actions):
    """grad-steps/s half of the metric on the host cores: the fp32 PyTorch-autograd restatement of
    DQNLearning.update_q (rlax_rainbow.py:153-217: three NoisyMLP forwards, C51 double-Q cross-entropy, IS weights,
    backward, Adam eps 3.125e-5) at B = 256 on torch-CPU — the reference's own learner is JAX and cannot run here. Uniform
    sampling from a 4 096-row ring (the PER tree on the host is timed separately: cpu_baseline_sum_tree)."""
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams

    cores = usable_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        n = 4096
        params = RlaxRainbowParams(use_priority=False, experience_buffer_size=n, mask_terminal=True, seed=7)
        # process_group=False: this agent lives on rank 0 only and must not all-reduce with the job's other ranks
        agent = DQNAgent(ObservationSpec((n, obs_len)), ActionSpec(n_actions), params, device="cpu", process_group=False)
        g = torch.Generator().manual_seed(0)
        o1 = (torch.rand(n, obs_len, generator=g) < 0.3).to(torch.int8)
        o2 = (torch.rand(n, obs_len, generator=g) < 0.3).to(torch.int8)
        legal = torch.ones(n, n_actions, dtype=torch.int8)
        agent.add_experience_first((None, (o1, legal)), torch.zeros(n))
        agent.add_experience((None, (o2, legal)), torch.randint(0, n_actions, (n,), generator=g), torch.rand(n, generator=g),
                             torch.ones(n))
        for _ in range(3):
            agent.update()
        t0 = time.perf_counter()
        k = 0
        while k < 400 and time.perf_counter() - t0 < 8.0:
            agent.update()
            k += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    return {"value": k / dt, "unit": "grad-steps/s", "cores": cores, "kind": "port", "batch": 256, "dtype": "float32",
            "ms_per_update": dt / k * 1e3,
            "sample": f"{k} update() calls of the torch-CPU fp32 learner (DQNLearning.loss + torch.optim.Adam, NoisyMLP "
                      f"{obs_len}->512->{n_actions}x51, batch 256, uniform replay), torch.set_num_threads({cores})"}


def sum_tree_baseline(device):
    """The one hot-path piece of the reference that builds here: its C++ SumTree<float> (oracle/_ref, compiled from
    sum_tree/sum_tree/include/sum_tree.h by oracle/Makefile) timed on the host, next to the HIP tree on the GPU,
    for the three operations of the PER cycle at capacity 2^19 (priority_buffer.py:29-52)."""
    import numpy as np

    from oracle import oracle_py as O
    import hanabi_hip

    out = {"kind": "reference", "capacity": 2 ** 19, "unit": "us per call"}
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 300000, 256)
    val = rng.random(256).astype(np.float32)
    q = rng.random(256).astype(np.float32)
    if O.RefTree.available():
        import ctypes

        # one OpenMP thread is the reference's fastest setting (its per-node mutexes make more threads
        # slower; with the box's default of one thread per visible core it is ~40x slower still)
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        out["reference_threads"] = 1
        ref = O.RefTree(2 ** 19)
        ins = np.arange(32768)
        t0 = time.perf_counter(); ref.update(ins, np.full(32768, 0.6, np.float32)); ins_s = time.perf_counter() - t0
        ref.update(np.arange(32768, 300000), np.full(300000 - 32768, 0.6, np.float32))

        def tm(fn, reps=200):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return (time.perf_counter() - t0) / reps * 1e6

        out["reference_cpu"] = {"update_values_256": tm(lambda: ref.update(idx, val)), "get_indices_256": tm(lambda: ref.sample(q)),
                                "insert_32768": ins_s * 1e6}
    tree = hanabi_hip.SumTree(2 ** 19, device=device)
    mx = torch.tensor([0.6], device=device)
    mn = mx.clone()
    tree.fill_range_dev(0, 300000, mx)
    di, dv = torch.as_tensor(idx, device=device), torch.as_tensor(val, device=device)
    du = torch.as_tensor(rng.random(256), device=device)

    def tg(fn, reps=200):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    out["hip_gpu"] = {"update_values_256": tg(lambda: tree.per_update_dev(di, dv, 0.6, mx, mn)),
                      "get_indices_256": tg(lambda: tree.per_sample_dev(du, unit=True)),
                      "insert_32768": tg(lambda: tree.fill_range_dev(1000, 32768, mx))}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("HB_BENCH_WATCHDOG"):   # debugging aid: dump every thread's Python stack and exit after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["HB_BENCH_WATCHDOG"]), exit=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.launch_check:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        return launch_check(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    # stdout carries exactly ONE line, the JSON result. Libraries write there too (RCCL prints its version banner to fd 1 when the
    # first communicator is created): from here on fd 1 is stderr, and the result goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # rehearsal knobs (one-GPU box): HB_BENCH_DEVICE pins every rank to one device, HB_DIST_BACKEND=gloo replaces RCCL,
    # which needs one GPU per rank. The driver sets neither.
    if "HB_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["HB_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HB_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knob (one-GPU box): HB_BENCH_FORCE_COLLECTIVE=1 with --gpus 1 starts a ONE-rank RCCL group and routes every
    # update through the multi-rank path (gradient bucket, graphs around the all-reduce): its cost without the wire time
    force_coll = world == 1 and os.environ.get("HB_BENCH_FORCE_COLLECTIVE") == "1"
    if force_coll:
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    n = args.games
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", args.players, flags), n_games=n, seed=1234,
                               first_game_id=rank * n, games_per_wave=args.games_per_wave, device=device,
                               packed=not args.unpacked_obs)
    # obs + legal + action/reward/step-type (9) + state row read and written. In packed mode the observation
    # leaves the kernel as obs_words u32 instead of obs_len bytes (84 vs 658 for 2 players); both figures are reported.
    bytes_int8_form = env.obs_len + env.num_actions + 9 + 2 * env.state_words * 4
    bytes_per_step = (env.obs_words * 4 if env.packed else env.obs_len) + env.num_actions + 9 + 2 * env.state_words * 4

    agents = []
    session = None
    main_stream = None
    if not args.env_only:
        params = RlaxRainbowParams(compute_dtype=args.compute_dtype, mask_terminal=True, seed=1234 + rank, packed_obs=env.packed,
                                   actor_lag=args.actor_lag, n_step=args.n_step)
        if args.vanilla:
            params = params._replace(distributional=False, use_priority=False)
        agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions),
                           params._replace(seed=1234 + 17 * s), device=device) for s in range(args.players)]
        for a in agents:
            a.first_game_id = rank * n  # keys the exploration RNG by global game id
            a.force_collective = force_coll
        if world > 1:  # identical initial weights on every rank (data parallel)
            for a in agents:
                for t in list(a.online.parameters()) + list(a.online.buffers()):
                    dist.broadcast(t.data, 0)
                a.target.load_state_dict(a.online.state_dict())
        lstream = not args.no_learner_stream
        if args.learner_cus > 0 and lstream:
            from hanabi_hip.streams import masked_stream

            lstream = lambda: masked_stream(0, args.learner_cus, device)   # (a factory: one masked stream per agent)
            main_stream = masked_stream(args.learner_cus, 256, device)
        session = SelfPlaySession(env, agents, updates_per_step=args.updates_per_step, learner_stream=lstream,
                                  learner_priority=args.learner_priority,
                                  stream_per_agent=None if args.stream_per_agent < 0 else bool(args.stream_per_agent))

    act = torch.empty(n, dtype=torch.int32, device=device)
    draw = [0]
    if main_stream is None and args.main_priority != 0:
        main_stream = torch.cuda.Stream(device=device, priority=args.main_priority)
    if main_stream is not None:
        main_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_stream)

    def one_step():
        if session is not None:
            session.step()
        else:
            env.random_legal_actions(4321, draw[0], out=act)
            draw[0] += 1
            env.step(act)

    for _ in range(args.prime + args.warmup):
        one_step()
    if session is not None:
        session.flush()

    # live per-dispatch timing of the env kernel inside the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:  # torch creates the HIP event on first record; do it outside the timed region
        a.record()
        b.record()
    grad0 = session.grad_steps if session else 0
    sel0 = session.select_in_env_steps if session else 0
    native0 = session.native_steps if session else 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev_t0, ev_main_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_t0.record()
    for k in range(args.steps):
        if k % args.event_every == 0:
            env.set_profile_events(*ev[k])
        elif k % args.event_every == 1:
            env.set_profile_events(None, None)
        one_step()
    ev_main_end.record()   # the acting stream's last launch of the timed region; what follows is the learner streams' drain
    if session is not None:
        session.flush()  # the last update's Adam / priority half (deferred behind the all-reduce when data-parallel)
    host_enqueue_s = time.perf_counter() - t0  # host-side launch time of the timed region (GPU still draining)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.set_profile_events(None, None)
    grad_steps = (session.grad_steps - grad0) if session else 0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kernel_ms = sorted(a.elapsed_time(b) for a, b in ev[::args.event_every])
    kernel_avg_s = sum(kernel_ms) / len(kernel_ms) / 1e3
    # Which form of the kernel the loop ran. Round 3: the one-kernel actor picks the moves itself (hb_actor_fused_act), so the
    # env kernel takes plain 4-byte actions (hb_env_step_packed); only agents without that actor still have the selection fused
    # into the env kernel (hb_env_step_select_packed: +4A q and +A legal bytes read per game).
    bytes_alone = bytes_per_step
    fused_sel = bool(session is not None and session.select_in_env_steps > sel0)
    if fused_sel:
        bytes_per_step = bytes_per_step + 5 * env.num_actions
    # `achieved` / `frac` follow SURVEY 8(d): ALGORITHMIC bytes per env-step at the reference's interface (int8 observations:
    # obs_len + A + 9 + 2 x state bytes = 943 B for 2 players, 1 721 B for 5) x env-steps per launch / launch duration. The kernel
    # the loop runs keeps the observation as bits (84 B instead of 658 B), so what it physically moves is less: `physical` gives
    # those bytes and their fraction of the HBM peak, `traffic` the PMC-measured bytes of that same kernel form, and the kernel
    # that really writes the int8 layout is measured separately ("int8_form").
    achieved = n * bytes_int8_form / kernel_avg_s / 1e9
    phys = n * bytes_per_step / kernel_avg_s / 1e9
    line = {
        "metric": "env_steps_per_sec",
        "value": world * n * args.steps / dt,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prime_steps": args.prime,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8" if args.env_only else f"u8 env/encoder + {args.compute_dtype} Q-net GEMM (fp32 accumulate, fp32 master weights)",
        "data": "synthetic",
        "config": {
            "workload": (("2-player full Hanabi, 32 768 envs, Rainbow (PER sum_tree + noisy C51) on 1 MI355X"
                          if (n == 32768 and args.players == 2 and not args.vanilla) else
                          "2-player full Hanabi, 4 096 parallel envs, vanilla DQN (uniform replay) on 1 MI355X"
                          if (n == 4096 and args.players == 2 and args.vanilla) else
                          f"{args.players}-player full Hanabi, {n} envs per GPU" + (", vanilla DQN (uniform replay)" if args.vanilla else ", Rainbow"))
                         + (f" x {world} GPUs (weak scaling: the same per GPU)" if world > 1 else "")),
            "games_per_gpu": n, "players": args.players, "train_batch": 256, "updates_per_step": args.updates_per_step,
            "policy": "random-legal (env only)" if args.env_only else "agent eps-greedy (eps 0.1)",
            "actor_lag": args.actor_lag,
            "parallelism": f"dp{world}: games sharded, RCCL gradient all-reduce",
        },
        "grad_steps_per_sec": world * 0 + (grad_steps / dt if grad_steps else 0.0),
        "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
        # how the timed region splits: the acting stream's K steps, then the learner streams finishing the last update(s) plus
        # the final synchronisation (a fixed cost per run: 20-step runs read ~10 % slower per step than 200-step runs)
        "timed_region_ms": {"total": dt * 1e3, "acting_stream": ev_t0.elapsed_time(ev_main_end),
                            "drain_and_sync": dt * 1e3 - ev_t0.elapsed_time(ev_main_end)},
        "roofline": {"bound": "hbm", "kernel": "hb::env_kernel (step + legal mask + canonical encoder)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "bytes_per_env_step": bytes_int8_form,
                     "accounting": "SURVEY 8(d): obs_len + A + 9 + 2 x state bytes per env-step (the reference's int8 observation "
                                   "interface) x env-steps per launch / this launch's duration",
                     "physical": {"bytes_per_env_step": bytes_per_step, "achieved": phys, "frac": phys / HBM_PEAK_GBS,
                                  "note": "bytes this kernel form really has to move (observation rows leave as bits)"},
                     "observation_form": ("bit-packed u32 rows + the agent's eps-greedy selection from q (hb_env_step_select_packed: "
                                          "+4A q, +A legal bytes read per game)" if fused_sel else
                                          "bit-packed u32 rows (hb_env_step_packed)" if env.packed else "int8 [N, obs_len] (hb_env_step)"),
                     "env_steps_per_launch": n,
                     "avg_launch_us": kernel_avg_s * 1e6, "median_launch_us": kernel_ms[len(kernel_ms) // 2] * 1e3,
                     "kernel_only_env_steps_per_sec": n / kernel_avg_s},
    }
    if session is not None:
        line["host_calls"] = {"steps_through_hb_chain_run": session.native_steps - native0, "of": args.steps,
                              "note": "steps issued by ONE host call (csrc/chain.hip) instead of ~25"}
    # HBM traffic of the same kernel/config from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950
    # correction + WRITE_SIZE; profiles/): PMC counters cannot be read from inside this process
    # (one file per kernel form and player count; the form with the selection fused in has its own pass)
    stem = "env_kernel_pmc_traffic" + ("_select" if fused_sel else "_packed" if env.packed else "") + (f"_{args.players}p" if args.players != 2 else "")
    pmc = next((f for f in (os.path.join(ROOT, "profiles", r, stem + ".json") for r in ("r03", "r02", "r01")) if os.path.exists(f)), None)
    if n == 32768 and pmc is not None:
        line["roofline"]["traffic"] = json.load(open(pmc))["per_launch_bytes"]["total"]
        line["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes, of the kernel form named in observation_form)"
    if session is not None:
        line["mean_episode_score"] = session.mean_score()
        # the same kernel with the GPU to itself (random-legal policy, no agents): inside the loop it shares the chip with
        # the learner stream's kernels, which lengthens its launches
        sa = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for a, b in sa:
            a.record()
            b.record()
        torch.cuda.synchronize()
        for k in range(60):
            env.set_profile_events(*sa[k])
            env.random_legal_actions(4321, 10_000 + k, out=act)
            env.step(act)
        torch.cuda.synchronize()
        env.set_profile_events(None, None)
        alone = sorted(a.elapsed_time(b) for a, b in sa[10:])
        alone_s = sum(alone) / len(alone) / 1e3
        line["roofline"]["standalone"] = {"avg_launch_us": alone_s * 1e6, "achieved": n * bytes_int8_form / alone_s / 1e9,
                                          "frac": n * bytes_int8_form / alone_s / 1e9 / HBM_PEAK_GBS, "bytes_per_env_step": bytes_int8_form,
                                          "physical": {"bytes_per_env_step": bytes_alone, "achieved": n * bytes_alone / alone_s / 1e9,
                                                       "frac": n * bytes_alone / alone_s / 1e9 / HBM_PEAK_GBS},
                                          "note": "same kernel (plain actions in), env-only stepping after the timed region (50 launches)"}
        # env kernel + the deck-pool refill launch that follows every `refill_period`-th step, amortised per step: stream time of
        # 60 env-only steps minus the same 60 launches of the random-legal policy kernel alone
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for k in range(60):
            env.random_legal_actions(4321, 30_000 + k, out=act)
            env.step(act)
        e1.record()
        for k in range(60):
            env.random_legal_actions(4321, 30_000 + k, out=act)
        e2.record()
        torch.cuda.synchronize()
        per_step_us = (e0.elapsed_time(e1) - e1.elapsed_time(e2)) / 60 * 1e3
        line["roofline"]["env_plus_refill"] = {"us_per_step": per_step_us, "achieved": n * bytes_int8_form / (per_step_us * 1e-6) / 1e9,
                                               "frac": n * bytes_int8_form / (per_step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                               "note": "stream time per env-only step including hb::refill_kernel (every third step for "
                                                       "Hanabi-Full) and the launch gaps, same 8(d) accounting"}
        if env.packed:
            # the reference-shaped output of the same kernel (int8 [N, obs_len], SURVEY §8(d)'s 943 B per env-step), env-only
            env8 = hanabi_hip.HanabiEnv(config=env.cfg, n_games=n, seed=1234, first_game_id=rank * n,
                                        games_per_wave=args.games_per_wave, device=device, packed=False)
            for k in range(60):
                env8.set_profile_events(*sa[k])
                env8.random_legal_actions(4321, 20_000 + k, out=act)
                env8.step(act)
            torch.cuda.synchronize()
            env8.set_profile_events(None, None)
            t8 = sorted(a.elapsed_time(b) for a, b in sa[10:])
            t8_s = sum(t8) / len(t8) / 1e3
            line["roofline"]["int8_form"] = {"bytes_per_env_step": bytes_int8_form, "avg_launch_us": t8_s * 1e6,
                                             "achieved": n * bytes_int8_form / t8_s / 1e9,
                                             "frac": n * bytes_int8_form / t8_s / 1e9 / HBM_PEAK_GBS,
                                             "note": "hb_env_step writing int8 [N, obs_len] observations, env-only (50 launches)"}
            del env8
        # every rank runs it (the learner update inside contains the gradient all-reduce); rank 0's numbers are printed
        line["roofline_qnet"] = qnet_roofline(agents[0], env, args)
        if rank == 0:
            from hanabi_agents.rlax_dqn.tolerance import TOLERANCE

            if args.compute_dtype in TOLERANCE:
                line["tolerance"] = dict(TOLERANCE[args.compute_dtype], dtype=args.compute_dtype, reference="fp32 PyTorch-autograd path "
                                         "(DQNLearning.loss + Adam; DQNPolicy.q_values)", test="tests/test_dtype_parity.py")
    if (session is not None and not args.vanilla and args.actor_lag == 0 and args.compute_dtype in ("bfloat16", "float16")
            and not args.no_async_variant and env.packed):
        session.flush()
        line["async_actor"] = _variant(async_variant, args, rank, world, device, n,   # (every rank: it holds collectives)
                                       streams=list(dict.fromkeys(session._lstreams.values())))
    if (session is not None and not args.vanilla and args.actor_lag == 0 and args.n_step == 1 and not args.no_nstep_variant
            and env.packed and args.compute_dtype in ("bfloat16", "float16")):
        # north_star: "n-step double-DQN loss". The reference's rlax agent is 1-step (the headline); the n-step form it describes
        # (replay_memory.py:316-345) is timed beside it
        session.flush()
        line["n_step_3"] = _variant(async_variant, args, rank, world, device, n, streams=list(dict.fromkeys(session._lstreams.values())),
                                    actor_lag=0, n_step=3)
    if (session is not None and not args.vanilla and args.actor_lag == 0 and args.n_step == 1 and not args.no_fp16_variant
            and env.packed and args.compute_dtype == "bfloat16"):
        # the reference's own network dtype (rlax_rainbow.py:250-251: fp16) on the same kernels (v_mfma_f32_16x16x32_f16: the same
        # MFMA rate; 8 x finer operand rounding: tolerance.py's float16 row instead of the bfloat16 one)
        session.flush()
        line["fp16_operands"] = _variant(async_variant, args, rank, world, device, n, streams=list(dict.fromkeys(session._lstreams.values())),
                                         actor_lag=0, n_step=1, dtype="float16")
    if rank == 0 and not args.no_cpu_baseline:
        note = lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)   # progress on stderr; stdout carries the ONE JSON line
        note(f"timed region done ({dt / args.steps * 1e3:.4f} ms per step); timing the CPU baselines on rank 0")
        line["cpu_baseline"] = cpu_baseline(args)
        note("cpu_baseline (env port) done")
        if not args.env_only:
            line["cpu_baseline_learner"] = cpu_baseline_learner(args, env.obs_len, env.num_actions)
            note("cpu_baseline_learner done")
        line["cpu_baseline_sum_tree"] = sum_tree_baseline(device)
        note("cpu_baseline_sum_tree done")
    if world > 1 or force_coll:
        # proof that N ranks on N devices took part: gathered over the job's own process group
        info = {"rank": rank, "device_index": device.index, "hostname": socket.gethostname(),
                "uuid": str(getattr(torch.cuda.get_device_properties(device), "uuid", "")), "name": torch.cuda.get_device_name(device)}
        gathered = [None] * (world if world > 1 else 1)
        dist.all_gather_object(gathered, info)
        ver = getattr(torch.cuda.nccl, "version", lambda: None)()
        line["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "version": list(ver) if ver else None,
                        "ranks": gathered, "distinct_devices": len({(g["hostname"], g["uuid"] or g["device_index"]) for g in gathered})}
    if force_coll:
        line["config"]["parallelism"] += " [HB_BENCH_FORCE_COLLECTIVE: one-rank RCCL group, multi-rank update path]"
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    os.close(result_fd)
    if world > 1:
        dist.barrier()
    if world > 1 or force_coll:
        dist.destroy_process_group()


def _variant(fn, *a, **kw):
    """A side measurement must never cost the headline its JSON line: single rank, a failure is reported in the variant's place.
    (Data-parallel runs re-raise: the other ranks are inside the variant's collectives and would hang.)"""
    try:
        return fn(*a, **kw)
    except Exception as e:   # noqa: BLE001
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise
        print(f"[bench] variant failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return {"error": f"{type(e).__name__}: {e}"}


def async_variant(args, rank, world, device, n, streams=None, actor_lag=1, n_step=None, dtype=None):
    """The same workload with the asynchronous actor (RlaxRainbowParams.actor_lag = 1, one learner stream per
    agent), timed with the same protocol as the headline. Reported BESIDE the headline, which keeps the reference's
    synchronous semantics: here the policy acts on weights that are one update old (tests/test_async_actor.py)."""
    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_hip.selfplay import SelfPlaySession

    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", args.players, flags), n_games=n, seed=1234,
                               first_game_id=rank * n, games_per_wave=args.games_per_wave, device=device, packed=True)
    params = RlaxRainbowParams(compute_dtype=dtype or args.compute_dtype, mask_terminal=True, seed=1234 + rank, packed_obs=True,
                               actor_lag=actor_lag, n_step=args.n_step if n_step is None else n_step)
    agents = [DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params._replace(seed=1234 + 17 * s),
                       device=device) for s in range(args.players)]
    for a in agents:
        a.first_game_id = rank * n
        a.force_collective = world == 1 and os.environ.get("HB_BENCH_FORCE_COLLECTIVE") == "1"
    if world > 1:
        for a in agents:
            for t in list(a.online.parameters()) + list(a.online.buffers()):
                dist.broadcast(t.data, 0)
            a.target.load_state_dict(a.online.state_dict())
    # the learner streams of the headline session are reused: which HIP hardware queue a stream lands on changes the step time
    # by up to 3x, and the headline's streams are the ones whose placement has just been measured
    pool = iter(streams or [])
    factory = (lambda: next(pool, None) or torch.cuda.Stream(device=device, priority=args.learner_priority)) if streams else True
    session = SelfPlaySession(env, agents, updates_per_step=args.updates_per_step, learner_stream=factory,
                              learner_priority=args.learner_priority, stream_per_agent=True)
    for _ in range(args.prime + args.warmup):
        session.step()
    session.flush()
    g0, native0 = session.grad_steps, session.native_steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        session.step()
    session.flush()
    host_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    out = {"actor_lag": actor_lag, "n_step": params.n_step, "learner_streams": "one per agent", "ms_per_step": dt / args.steps * 1e3,
           "env_steps_per_sec": world * n * args.steps / dt, "grad_steps_per_sec": (session.grad_steps - g0) / dt,
           "host_enqueue_ms_per_step": host_s / args.steps * 1e3}
    if dtype:
        from hanabi_agents.rlax_dqn.tolerance import TOLERANCE

        out["compute_dtype"] = dtype
        out["host_calls"] = {"steps_through_hb_chain_run": session.native_steps - native0, "of": args.steps}
        out["tolerance"] = {k: TOLERANCE[dtype][k] for k in ("q_abs", "argmax_gap", "td_abs", "td_rel")}
        out["note"] = ("synchronous agent, the headline's protocol with fp16 GEMM operands (weights and hidden activations; fp32 "
                       "accumulation, master weights, softmax and loss as before): the reference's own network dtype")
    elif actor_lag:
        out["note"] = ("policy acts on the weights of the update before last (one-update staleness, tests/test_async_actor.py); "
                       "not the headline: the reference's agent is synchronous")
    else:
        out["note"] = ("synchronous agent, n-step returns assembled at sample time by walking the ring (hb_per_sample_gather; spec: "
                       "hanabi_agents/rainbow/replay_memory.py:316-345), same protocol as the headline")
    return out


def qnet_roofline(agent, env, args):
    """MFMA side: time the actor forward (N rows) and one learner update with events; FLOPs are the
    ALGORITHMIC ones of SURVEY §8(d) (plain + noisy GEMM per layer), although the merged-weight form
    executes half of the forward matrix FLOPs."""
    n = env.n
    hidden = agent.params.layers[0]
    if agent.distributional:
        fwd_flop = 4.0 * (env.obs_len * hidden + hidden * env.num_actions * agent.params.n_atoms)  # per sample
    else:  # vanilla scalar head: one GEMM per layer (0.69 MFLOP per sample for 2 players)
        fwd_flop = 2.0 * (env.obs_len * hidden + hidden * env.num_actions)
    obs = (None, (env.net_obs, env.legal))
    for _ in range(3):
        agent.explore(obs)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    reps = 10
    for _ in range(reps):
        agent.explore(obs)
    b.record()
    torch.cuda.synchronize()
    actor_s = a.elapsed_time(b) / 1e3 / reps
    a.record()
    for _ in range(reps):
        agent.update()
    b.record()
    torch.cuda.synchronize()
    learn_s = a.elapsed_time(b) / 1e3 / reps
    peak = MFMA_PEAK_TFLOPS[args.compute_dtype]
    actor_tf = fwd_flop * n / actor_s / 1e12
    per_kernel = None
    fl0 = getattr(agent, "_fl", None)
    if fl0 is not None and fl0.actor is not None and agent.use_mfma_actor:
        # each hand-written kernel alone (events around back-to-back launches; executed FLOPs of its GEMM)
        from hanabi_hip import _capi as K

        ac, L, s = fl0.actor, K.lib(), K.current_stream()
        obs8, legal8, support = env.net_obs, env.legal, agent.atoms[0].contiguous()
        acts = torch.empty(n, dtype=torch.int32, device=legal8.device)
        launches, flops = {}, {}
        if ac.two_kernel:   # (bf16 operands; fp16 has the one-kernel form only)
            if ac._two_stale[0]:
                ac._pack_two(0)   # (the two-kernel form's weight copies are refreshed lazily since round 3)
            hidden_fn = L.hb_actor_hidden_packed if env.packed else L.hb_actor_hidden
            launches = {
                "hb_actor_hidden": lambda: hidden_fn(K.dptr(obs8), n, ac.obs_len, K.dptr(ac.w1t), ac.k_pad, K.dptr(ac.b1), ac.hidden,
                                                     K.dptr(ac.h), s),
                "hb_actor_q": lambda: L.hb_actor_q(K.dptr(ac.h), n, ac.hidden, K.dptr(ac.w2t), K.dptr(ac.b2), K.dptr(support), ac.n_actions,
                                                   ac.n_atoms, K.dptr(ac.q), s),
                "hb_policy_select": lambda: L.hb_policy_select(K.dptr(ac.q), K.dptr(legal8), n, ac.n_actions, 0.1, 1, 1, 0, K.dptr(acts), s),
            }
            flops = {"hb_actor_hidden": 2.0 * ac.k_pad * ac.hidden * n, "hb_actor_q": 2.0 * ac.hidden * ac.w2t.shape[0] * n}
        if ac.takes_fused(obs8):
            # round 3: the whole forward + selection as ONE kernel (csrc/actor_fused.hip): what the loop runs
            f = ac._fset_ptrs[0]
            n_pass = (ac.n_actions + 9) // 10
            launches = dict({"hb_actor_fused_act": lambda: L.hb_actor_fused_act_dt(
                K.dptr(obs8), K.dptr(legal8), n, ac.obs_len, f[0], f[1], f[2], f[3], K.dptr(support), ac.hidden, ac.n_actions, ac.n_atoms,
                K.dptr(ac.q), 0.1, 1, 1, 0, K.dptr(acts), ac._dt, s)}, **launches)
            flops["hb_actor_fused_act"] = 2.0 * n * (ac.k_pad * ac.hidden + ac.hidden * 512 * n_pass)
        per_kernel = {}
        for name, fn in launches.items():
            for _ in range(3):
                fn()
            a.record()
            for _ in range(20):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            per_kernel[name] = {"avg_launch_us": us}
            if name in flops:
                tf = flops[name] / (us * 1e-6) / 1e12
                per_kernel[name].update(executed_gflop=flops[name] / 1e9, achieved=tf, frac=tf / peak)
    # what the actor kernels execute: merged weights (one GEMM per layer), K padded to a multiple of 64, output columns in
    # 256-column groups of whole actions (csrc/actor.hip)
    fl = getattr(agent, "_fl", None)
    mfma_actor = fl is not None and fl.actor is not None and agent.use_mfma_actor
    kp = -(-env.obs_len // 64) * 64
    ncols = (fl.actor.w2t.shape[0] if mfma_actor and fl.actor.two_kernel else
             512 * ((env.num_actions + 9) // 10) if mfma_actor else -(-env.num_actions * agent.params.n_atoms // 64) * 64)
    if not agent.distributional:
        ncols = env.num_actions
    exec_flop = 2.0 * (kp * hidden + hidden * ncols)
    exec_tf = exec_flop * n / actor_s / 1e12
    learn_tf = 5.0 * 256 * fwd_flop / learn_s / 1e12
    # learner: 3 forward passes of B rows (online on obs_tm1 and obs_t, target on obs_t; the merged layer-1 GEMM also computes
    # the unused target half of obs_tm1) + backward (dW2, dH, dW1) on the merged weights
    if fl is not None and getattr(fl, "sparse_backward", False):
        # sparse backward (csrc/learner2.hip): dH and dW2 touch only the K atoms of the action taken; dW1 stays a dense GEMM
        learn_exec_flop = 256 * (3 * exec_flop + 2.0 * (2 * hidden * 64 + kp * hidden))
    else:
        learn_exec_flop = 256 * (3 * exec_flop + 2.0 * (2 * hidden * ncols + kp * hidden))
    learn_exec_tf = learn_exec_flop / learn_s / 1e12
    # `achieved` / `frac` price the FLOPs the kernels EXECUTE (merged weights: one GEMM per layer); the literal two-GEMM
    # count of SURVEY §8(d) stays beside them as algorithmic_*
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "dtype": args.compute_dtype,
            "actor_forward": {"rows": n, "executed_gflop": exec_flop * n / 1e9, "ms": actor_s * 1e3, "achieved": exec_tf,
                              "frac": exec_tf / peak, "algorithmic_gflop": fwd_flop * n / 1e9, "algorithmic_achieved": actor_tf,
                              "algorithmic_frac": actor_tf / peak,
                              "kernels": ("hb_actor_fused_act: bit rows -> q values -> eps-greedy moves in ONE kernel (hand-written MFMA, "
                                          "csrc/actor_fused.hip); per_kernel also times the two-kernel form it replaces"
                                          if (mfma_actor and fl.actor.takes_fused(env.net_obs)) else
                                          "hb_actor_hidden + hb_actor_q + hb_policy_select (hand-written MFMA, csrc/actor.hip)"
                                          if mfma_actor else "hb_actor_hidden (MFMA) + library GEMM [N,H]x[H,A] + hb_policy_select"
                                          if getattr(agent, "_plain_fast", False) else "hb_obs_cast + hipBLASLt GEMMs + hb_policy_act"),
                              "per_kernel": per_kernel},
            "learner_update": {"batch": 256, "executed_gflop": learn_exec_flop / 1e9, "ms": learn_s * 1e3,
                               "achieved": learn_exec_tf, "frac": learn_exec_tf / peak,
                               "algorithmic_gflop": 5.0 * 256 * fwd_flop / 1e9, "algorithmic_achieved": learn_tf,
                               "algorithmic_frac": learn_tf / peak, "grad_steps_per_sec_alone": 1.0 / learn_s}}


if __name__ == "__main__":
    main()
