"""SelfPlaySession — minimal lock-step self-play driver.

The reference ships no driver: its agents are passive objects called by the authors' external
`hanabi_multiagent_framework` session. This is the smallest loop that exercises the
agent API the way that session does, entirely on the GPU:

    seat = t mod P                       (all N games act with the same seat: HB_FLAG_RESET_START_NEXT)
    agent[seat].add_experience_first(obs, step_types)     rows whose seat has no move pending yet
    agent[seat].add_experience(obs, last_actions[seat], agent_rewards, step_types)
    actions = agent[seat].explore(obs)
    env.step(actions)                    -> obs / legal / per-seat reward + step type for seat+1
    agent[seat].update()  x updates_per_step

Per-seat reward accumulation across the other seats' turns, FIRST/MID/LAST generation and the
re-deal of finished games happen inside the env kernel (include/hanabi_hip.h, hb_env_step), so
the driver itself moves no data. An agent's `obs_t` is its own next turn; when its episode ended in
between, step type is LAST and `obs_t` is already the first observation of the fresh game (which
the learner ignores when `mask_terminal` is on).
"""
import torch

from . import _capi as K


class SelfPlaySession:
    def __init__(self, env, agents, updates_per_step=1, min_replay=None, train_seats=None, overlap_allreduce=None,
                 learner_stream=True, learner_priority=-1, stream_per_agent=None, fuse_select=True, split_update=True):
        assert len(agents) == env.players, "one agent per seat"
        self.env = env
        self.agents = list(agents)
        self.updates_per_step = int(updates_per_step)
        self.t = 0
        self.last_actions = [torch.zeros(env.n, dtype=torch.int32, device=env.device) for _ in agents]
        self._act_buf = [torch.zeros(env.n, dtype=torch.int32, device=env.device) for _ in agents]
        self.fuse_select = bool(fuse_select) and getattr(env, "packed", False) and env.device.type == "cuda"
        self.min_replay = min_replay
        self.train_seats = set(range(env.players)) if train_seats is None else set(train_seats)
        if overlap_allreduce is None:  # only worth the reordering when there is a collective to hide
            import torch.distributed as dist

            overlap_allreduce = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.overlap_allreduce = bool(overlap_allreduce) and len(set(map(id, agents))) == len(agents)
        # Learner stream: seat A's update only has to finish before A acts again, so it is enqueued on a second HIP
        # stream right after A's policy forward and runs beside the env step and the NEXT seat's insert / policy /
        # env step (big GEMMs on the main stream, the ~30 small learner kernels on the side). Every dependency of the
        # sequential order is kept by events, so results are identical to running everything on one stream.
        self.learner_stream = None
        self._stream_factory = None
        if learner_stream and env.device.type == "cuda" and len(set(map(id, agents))) == len(agents) and len(agents) > 1:
            # (a torch.cuda.Stream / ExternalStream may be passed in, e.g. one restricted to a CU subset: streams.py)
            # (also accepted: a callable returning a new stream each time it is called — one per agent with per-agent streams)
            self._stream_factory = learner_stream if callable(learner_stream) and not isinstance(learner_stream, torch.cuda.Stream) else None
            self.learner_stream = (self._stream_factory() if self._stream_factory else
                                   learner_stream if isinstance(learner_stream, torch.cuda.Stream) else
                                   torch.cuda.Stream(device=env.device, priority=learner_priority))
        self._main, self._main_raw = None, -1
        import torch.distributed as dist

        self._dp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # one learner stream per agent: seat A's update (launched at step t, needed at t + P) runs beside seat B's (t + 1):
        # the two latency-bound kernel chains interleave instead of queueing one behind the other
        # (default: data-parallel runs, and asynchronous actors — there the acting stream waits for the seat's own `gathered`
        # event only, which on a shared learner stream would sit behind the OTHER seat's whole update: 0.137 vs 0.150 ms per step)
        lagging = any(getattr(a, "actor_lag", 0) for a in agents)
        self._stream_per_agent = (self._dp or lagging) if stream_per_agent is None else bool(stream_per_agent)
        self._lstreams = {}
        self.max_learner_streams = 2
        if self.learner_stream is not None and split_update:
            for a in agents:
                if hasattr(a, "set_split_update"):
                    a.set_split_update(True)
            if stream_per_agent is None and any(getattr(a, "split_update", False) for a in agents):
                # the acting stream waits for events INSIDE the seat's own update (rings read, weights written); on a shared
                # learner stream those would sit behind the other seat's whole update (r02: 0.145 vs 0.153 ms per step)
                self._stream_per_agent = True
        self._update_done = {}  # agent id -> event recorded on the learner stream after its last update
        self._acted_ev, self._done_ev = {}, {}   # persistent events (creating one costs more than recording it)
        self.env_steps = 0
        self.grad_steps = 0
        self._inflight = None  # agent whose update_begin() has run but not its update_finish() (data-parallel overlap)
        self._stats0 = env.stats()

    def step(self, train=True, explore=True):
        env = self.env
        seat = self.t % env.players
        agent = self.agents[seat]
        main = None
        if self.learner_stream is not None:
            raw = K.current_stream().value  # (torch.cuda.current_stream() costs ~8 us of Python: look it up only when it changed)
            if raw != self._main_raw:
                self._main, self._main_raw = torch.cuda.current_stream(), raw
            main = self._main
        if main is not None and id(agent) in self._update_done:
            if getattr(agent, "split_update", False):
                # split update: the acting stream waits only until the seat's update in flight has READ the replay rings (its
                # first launch) before it inserts; the sum tree is written on the learner stream alone; the policy call then
                # waits for the optimizer step's event (synchronous agent) or takes the other weight buffer (actor_lag = 1)
                self._update_done.pop(id(agent))
                if agent.gathered_ev is not None:
                    agent.gathered_ev.wait(raw)
            else:
                self._update_done.pop(id(agent)).wait(raw)  # its replay / weights are being written by that update
        # [0]: the rich observation source for agents with requires_vectorized_observation() False (rule-based
        # partners read the env's state rows); [1]: the vectorised (obs, legal) pair the DQN agents use
        observations = (env, (env.net_obs, env.legal))   # packed env: the bit rows, which the agents take as they are
        if self.t < env.players:
            # only during the first round can a seat be without a pending move (step type FIRST)
            agent.add_experience_first(observations, env.agent_step_type)
            agent.add_experience(observations, self.last_actions[seat], env.agent_reward, env.agent_step_type)
        else:
            agent.add_experience_dense(observations, self.last_actions[seat], env.agent_reward, env.agent_step_type)
        sel = agent.q_for_step(observations, explore) if (self.fuse_select and hasattr(agent, "q_for_step")) else None
        if sel is not None:
            # the network's q values go straight into the env kernel, which picks each game's move by the agent's own rule and
            # draws (identical actions) and applies it: no selection launch, no round trip of the actions
            actions = self._act_buf[seat]
            env.step_select(sel[0], sel[1], sel[2], sel[3], sel[4], actions_out=actions)
        else:
            actions = agent.explore(observations) if explore else agent.exploit(observations)
            env.step(actions)
        self.last_actions[seat] = actions
        acted = None
        if main is not None:
            # The policy has read the weights: the learner may overwrite them from here on. Recorded AFTER the env step:
            # releasing the learner before it is throughput-neutral (0.194 ms either way) but makes the HBM-bound env
            # kernel share the chip with the update's first kernels (18 us per launch instead of 14).
            acted = self._acted_ev.get(seat)   # (re-recording an event does not disturb waits already enqueued on it)
            if acted is None:
                acted = self._acted_ev[seat] = K.Event()
            acted.record(raw)
        self.env_steps += env.n
        if self.learner_stream is None:
            self._train_inline(agent, seat, train)
        elif train and seat in self.train_seats and self._ready(agent):
            ls = self._learner_stream_of(agent)
            lraw = ls.cuda_stream
            K.set_stream(ls)   # (the context manager costs ~20 us of Python per use)
            try:
                acted.wait(lraw)
                for _ in range(self.updates_per_step):
                    agent.update_begin()
                    agent.update_finish()
                    self.grad_steps += 1
                done = self._done_ev.get(id(agent))
                if done is None:
                    done = self._done_ev[id(agent)] = K.Event()
                if not getattr(agent, "split_update", False):   # (split update: agent.gathered_ev / weights_ev instead)
                    done.record(lraw)
            finally:
                K.set_stream(main)
            self._update_done[id(agent)] = done
        self.t += 1

    def _learner_stream_of(self, agent):
        """One learner stream with a single rank (measured best: 0.195 ms per step). Data-parallel: one per agent, so that
        seat A's gradient all-reduce (tens of microseconds of xGMI latency in the middle of its update) is in flight while
        seat B's update computes, instead of both queueing on one stream whose updates would then outlast a step."""
        if not self._stream_per_agent:
            return self.learner_stream
        ls = self._lstreams.get(id(agent))
        if ls is None:
            # at most max_learner_streams distinct streams, dealt round-robin in seat order: consecutive seats' updates never
            # share a stream (what "per agent" is for), and a 5-seat session does not open 5 high-priority streams
            made = list(dict.fromkeys(self._lstreams.values()))
            if len(made) >= self.max_learner_streams:
                ls = made[len(self._lstreams) % self.max_learner_streams]
            else:
                ls = (self.learner_stream if not made else self._stream_factory() if self._stream_factory else
                      torch.cuda.Stream(device=self.env.device, priority=self.learner_stream.priority))
            self._lstreams[id(agent)] = ls
        return ls

    def _ready(self, agent):
        if not hasattr(agent, "experience"):  # passive partner (rule-based): nothing to train
            return False
        need = self.min_replay if self.min_replay is not None else agent.params.train_batch_size
        return agent.experience.size >= need

    def _train_inline(self, agent, seat, train):
        # The previous seat's gradient all-reduce has been running behind this seat's insert / policy / env step:
        # apply it now. (An agent always finishes its update before it acts again: seats alternate.)
        if self._inflight is not None:
            self._inflight.update_finish()
            self._inflight = None
        if train and seat in self.train_seats and self._ready(agent):
            for k in range(self.updates_per_step):
                agent.update_begin()
                self.grad_steps += 1
                if k + 1 < self.updates_per_step or not self.overlap_allreduce or self.env.players == 1:
                    agent.update_finish()
                else:
                    self._inflight = agent

    def flush(self):
        """Complete an update left in flight by the last step() and rejoin the learner stream."""
        if self._inflight is not None:
            self._inflight.update_finish()
            self._inflight = None
        if self.learner_stream is not None:
            if any(getattr(a, "_pending_fills", None) for a in self.agents):
                # (actor_lag) leaves of rows inserted after the last update: set them where the tree is written
                cur = torch.cuda.current_stream()
                for a in self.agents:
                    if getattr(a, "_pending_fills", None):
                        ls = self._learner_stream_of(a)
                        ls.wait_stream(cur)
                        with torch.cuda.stream(ls):
                            a.apply_pending_fills()
            torch.cuda.current_stream().wait_stream(self.learner_stream)
            for ls in self._lstreams.values():
                torch.cuda.current_stream().wait_stream(ls)
            self._update_done.clear()
        else:
            for a in self.agents:
                if getattr(a, "_pending_fills", None):
                    a.apply_pending_fills()

    def run(self, steps, train=True):
        for _ in range(steps):
            self.step(train=train)
        self.flush()

    # ---- checkpoint / resume -----------------------------------------------------------------
    def checkpoint_state(self, include_replay=True):
        """Everything a bit-exact resume needs: env rows (the deck pool is re-derived from them), turn counter,
        pending actions, progress counters and one checkpoint per distinct agent."""
        self.flush()
        torch.cuda.synchronize(self.env.device)
        ep, sc = self.env.stats()
        uniq, slot = [], []
        for a in self.agents:
            if not any(a is u for u in uniq):
                uniq.append(a)
            slot.append(next(i for i, u in enumerate(uniq) if a is u))
        return dict(format="hanabi-agents_amd/session/1", t=self.t, env_steps=self.env_steps, grad_steps=self.grad_steps,
                    episodes=ep - self._stats0[0], score_sum=sc - self._stats0[1], games=self.env.n,
                    env_seed=self.env.seed, first_game_id=self.env.first_game_id,
                    env_rows=self.env.export_state().cpu(), last_actions=[a.cpu() for a in self.last_actions],
                    agent_slot=slot, agents=[u.checkpoint_state(include_replay) for u in uniq])

    def load_checkpoint_state(self, sd):
        if sd.get("format") != "hanabi-agents_amd/session/1":
            raise ValueError("not a session checkpoint")
        if sd["games"] != self.env.n or len(sd["agent_slot"]) != len(self.agents):
            raise ValueError("checkpoint was written for a different number of games / seats")
        if (sd["env_seed"], sd["first_game_id"]) != (self.env.seed, self.env.first_game_id):
            raise ValueError("the env's deck seed / first game id differ from the checkpoint's: future deals would diverge")
        self.flush()
        self.env.import_state(sd["env_rows"])
        self.env.observe()  # obs / legal / per-seat reward + step type of the seat about to act
        for dst, src in zip(self.last_actions, sd["last_actions"]):
            dst.copy_(src)
        done = set()
        for a, k in zip(self.agents, sd["agent_slot"]):
            if id(a) not in done:
                a.load_checkpoint_state(sd["agents"][k])
                done.add(id(a))
        self.t, self.env_steps, self.grad_steps = int(sd["t"]), int(sd["env_steps"]), int(sd["grad_steps"])
        ep, sc = self.env.stats()
        self._stats0 = (ep - sd["episodes"], sc - sd["score_sum"])
        self._update_done.clear()
        self._inflight = None

    def save_checkpoint(self, path, include_replay=True):
        torch.save(self.checkpoint_state(include_replay), path)

    def load_checkpoint(self, path):
        self.load_checkpoint_state(torch.load(path, map_location="cpu", weights_only=True))

    @property
    def episodes(self):
        """Episodes finished since this session started (counted inside the env kernel)."""
        return self.env.stats()[0] - self._stats0[0]

    def mean_score(self):
        ep, sc = self.env.stats()
        ep, sc = ep - self._stats0[0], sc - self._stats0[1]
        return sc / ep if ep else float("nan")
