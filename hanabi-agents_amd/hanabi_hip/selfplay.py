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
import os

import torch

from . import _capi as K


class SelfPlaySession:
    def __init__(self, env, agents, updates_per_step=1, min_replay=None, train_seats=None, overlap_allreduce=None,
                 learner_stream=True, learner_priority=-1, stream_per_agent=None, fuse_select=True, split_update=True,
                 native_chain=True, early_update=True):
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
        # (rounds 1-2: only data-parallel runs and asynchronous actors — there the acting stream waits for the seat's own `gathered`
        # event only, which on a shared learner stream would sit behind the OTHER seat's whole update: 0.137 vs 0.150 ms per step)
        # (round 3: also the plain synchronous agents — config 2's vanilla DQN went 0.163 -> 0.139 ms per step: each agent's update
        #  only has to finish before that agent acts again, and two 0.14 ms updates on ONE stream outlast two steps)
        self._stream_per_agent = True if stream_per_agent is None else bool(stream_per_agent)
        self._lstreams = {}
        self.max_learner_streams = int(os.environ.get("HB_MAX_LEARNER_STREAMS", "2"))   # (the variable: measurements)
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
        # One host call per step (hb_chain_run, csrc/chain.hip): once a seat's step has run through the ordinary path below and
        # every buffer, event and the update's graph exist, the same launches are replayed from a command array filled once.
        self.native_chain = bool(native_chain) and env.device.type == "cuda"
        # Early update (round 3, with the one-call step): an agent's update starts right after its replay INSERT — beside its own
        # policy kernel, which like the update's forward / loss / backward only reads the weights — and only the optimizer step
        # with the weight packs waits for the policy kernel (`acted`). The reference's order of effects is unchanged (insert,
        # act on W_t, update W_t -> W_t+1 from a ring that contains the inserted rows); what changes is that the update's head no
        # longer sits between one policy kernel and the next (0.115 -> see DESIGN section 8).
        self.early_update = (bool(early_update) and self.native_chain and self.learner_stream is not None
                             and os.environ.get("HB_EARLY_UPDATE", "1") != "0")   # (the variable: A/B measurements)
        if self.early_update:
            for a in agents:
                if hasattr(a, "set_two_graphs") and getattr(a, "split_update", False) and not getattr(a, "actor_lag", 0):
                    a.set_two_graphs(True)
        self._inserted_ev = {}
        self._acted_early = os.environ.get("HB_ACTED_BEFORE_ENV", "1") != "0"
        self.select_in_env_steps = 0   # steps whose moves were picked inside the env kernel (hb_env_step_select_packed)
        self._chains = {}        # seat -> _Chain
        self.native_steps = 0

    # ---- one host call per step ----------------------------------------------------------------------------------------------
    def _chain_for(self, seat, agent, train, raw):
        """The seat's command array, built when (and rebuilt if) the conditions of the replay hold; None -> ordinary path.
        Covered: packed env, one-kernel actor, synchronous agent with split update on its own learner stream, prioritized
        replay, the whole update captured in ONE graph (single rank), one update per step."""
        env = self.env
        if not (self.native_chain and train and self.learner_stream is not None and self.fuse_select and self.updates_per_step == 1
                and seat in self.train_seats and self.t >= env.players and getattr(env, "packed", False)):
            return None
        if not (hasattr(agent, "_fl") and getattr(agent, "split_update", False)):
            return None
        fl, p = agent._fl, agent.params
        lag = int(getattr(agent, "actor_lag", 0))
        if (fl is None or fl.actor is None or fl.actor_stale or not p.use_priority or p.resample_noise or not agent.use_mfma_actor
                or agent._graph1 is None or (agent._graph2 is not None) != bool(getattr(agent, "two_graphs", False) and not lag)
                or agent._pending is not None or agent._pending_fills
                or agent._dense_call is None or fl._sg_call is None or agent.gathered_ev is None or agent.weights_ev is None
                or seat not in self._acted_ev or not self._ready(agent) or not fl.actor.takes_fused(env.net_obs)):
            return None
        if agent.train_step % p.target_update_period == 0:
            return None   # (this update is followed by a target sync: torch copies, ordinary path)
        buf = agent.experience
        wset = fl.n_packed % 2 if lag else 0
        if lag and (fl.packed_ev[wset] is None or fl.actor._two_stale[wset]):
            return None   # (this weight set has not been through the ordinary path yet)
        ch = self._chains.get((seat, wset))
        ls = self._learner_stream_of(agent)
        key = (raw, ls.cuda_stream, wset, agent._dense_call[0], fl._sg_call[0], fl._sg_call[1], id(agent._graph1), buf.rows_per_insert,
               self.last_actions[seat].data_ptr(), self._act_buf[seat].data_ptr(), agent._g_idx.data_ptr(), agent._g_prios.data_ptr())
        if ch is None or ch.key != key:
            # the cached insert call must be the one the ordinary path would make NOW (same seven operands, add_experience_dense)
            now = (env.net_obs.data_ptr(), env.legal.data_ptr(), self.last_actions[seat].data_ptr(), env.agent_reward.data_ptr(),
                   env.agent_step_type.data_ptr(), agent.last_obs.data_ptr(), buf._obs_t_buf.data_ptr())
            if agent._dense_call[0] != now or agent._dense_call[1] != env.n or self.last_actions[seat] is not self._act_buf[seat]:
                return None
            ch = self._chains[(seat, wset)] = _Chain(self, seat, agent, raw, ls.cuda_stream, key, wset)
        return ch

    def _native_step(self, ch, seat, agent, explore):
        env, buf, fl = self.env, agent.experience, agent._fl
        n = env.n
        vi, vf = ch.vi, ch.vf
        start = buf.oldest_entry
        agent._draws += 1
        vi[0], vi[1], vi[2], vi[3] = start, agent._draws, start, n
        vi[4] = 1 if (self._update_done.pop(id(agent), None) is not None and agent.gathered_ev.recorded) else 0
        vi[5] = 1 if ((fl.packed_ev[ch.wset].recorded if ch.lag else agent.weights_ev.recorded)) else 0
        vf[0] = float(agent.params.epsilon(agent.train_step)) if explore else 0.0
        buf._advance(n)
        # device scalars the update reads: refreshed (on the learner stream, as update_begin does) only when they change
        beta = float(agent.params.beta_is(agent.train_step))
        if buf._synced != (buf.size, buf.oldest_entry) or beta != agent._beta_host:
            K.set_stream(ch.ls)
            try:
                buf.sync_size()
                if beta != agent._beta_host:
                    agent._beta.fill_(beta)
                    agent._beta_host = beta
            finally:
                K.set_stream(self._main)
        K.check(ch.run(ch.cmds, ch.count, vi, vf))
        agent.gathered_ev.recorded = agent.weights_ev.recorded = True
        self._acted_ev[seat].recorded = True
        if ch.lag:   # (weights_updated() of the ordinary path: the set just packed, and its event)
            fl.packed_ev[ch.wset].recorded = True
            fl.n_packed += 1
        env._obs_stale = True
        self.last_actions[seat] = self._act_buf[seat]
        agent._eff_cache = None
        if not ch.lag:
            fl.weights_updated()
        agent.train_step += 1
        self._update_done[id(agent)] = ch.done
        self.env_steps += n
        self.grad_steps += 1
        self.native_steps += 1
        self.t += 1

    def step(self, train=True, explore=True):
        env = self.env
        seat = self.t % env.players
        agent = self.agents[seat]
        if self.native_chain and self.learner_stream is not None:
            raw = K.current_stream().value
            if raw != self._main_raw:
                self._main, self._main_raw = torch.cuda.current_stream(), raw
            ch = self._chain_for(seat, agent, train, raw)
            if ch is not None:
                return self._native_step(ch, seat, agent, explore)
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
        early = self._acted_early and main is not None

        def record_acted():
            # The policy has read the weights: the learner may overwrite them from here on. Round 3: recorded BEFORE the env step,
            # so the update's head (tree fill, sample + gather, forward GEMMs: small kernels that fit beside the others) starts
            # while the env kernel runs: 0.124 -> 0.115 ms per step. (Rounds 1-2 recorded it after the env step: with the library
            # GEMMs of the time the update's first kernels slowed the env kernel by as much as they gained.)
            ev = self._acted_ev.get(seat)   # (re-recording an event does not disturb waits already enqueued on it)
            if ev is None:
                ev = self._acted_ev[seat] = K.Event()
            ev.record(raw)
            return ev

        acted = None
        actions = None
        if self.fuse_select and hasattr(agent, "act_for_step"):
            # one-kernel actor: forward + selection in ONE launch; the env kernel then takes plain moves (4 B per game instead of
            # the 100-byte q and legal rows the selection fused into the env step reads)
            actions = agent.act_for_step(observations, explore, actions_out=self._act_buf[seat])
        sel = None
        if actions is None and self.fuse_select and hasattr(agent, "q_for_step"):
            sel = agent.q_for_step(observations, explore)
        if actions is not None:
            if early:
                acted = record_acted()
            env.step(actions)
        elif sel is not None:
            # the network's q values go straight into the env kernel, which picks each game's move by the agent's own rule and
            # draws (identical actions) and applies it: no selection launch, no round trip of the actions
            actions = self._act_buf[seat]
            env.step_select(sel[0], sel[1], sel[2], sel[3], sel[4], actions_out=actions)
            self.select_in_env_steps += 1
        else:
            actions = agent.explore(observations) if explore else agent.exploit(observations)
            if early:
                acted = record_acted()
            env.step(actions)
        self.last_actions[seat] = actions
        if main is not None and acted is None:
            acted = record_acted()
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


class _Chain:
    """One seat's step as an hb_cmd array (include/hanabi_hip.h, hb_chain_run): acting stream — wait (rings read), replay insert,
    wait (weights written), policy kernel with the selection inside, env step, record `acted`; learner stream — wait `acted`, tree
    leaves of the inserted rows, sample + gather, record `gathered`, the captured update, record `weights`, priority write-back.
    The same launches in the same order on the same streams as SelfPlaySession.step's ordinary path."""

    def __init__(self, session, seat, agent, raw, lraw, key, wset=0):
        import ctypes as C

        env, buf, fl = session.env, agent.experience, agent._fl
        self.key, self.ls = key, session._learner_stream_of(agent)
        self.wset, self.lag = wset, bool(getattr(agent, "actor_lag", 0))
        cmds = (K.HbCmd * 20)()
        A, Ls = C.c_void_p(raw), C.c_void_p(lraw)
        n_cmd = [0]

        def put(op, stream, ptrs=(), ints=(), floats=(), var=-1, fvar=-1, cond=-1):
            c = cmds[n_cmd[0]]
            n_cmd[0] += 1
            c.op, c.var, c.fvar, c.cond, c.stream = op, var, fvar, cond, stream
            for j, v in enumerate(ptrs):
                c.p[j] = v.value if isinstance(v, C.c_void_p) else v
            for j, v in enumerate(ints):
                c.i[j] = int(v)
            for j, v in enumerate(floats):
                c.f[j] = float(v)

        acted = session._acted_ev[seat]
        done = session._done_ev.get(id(agent))
        if done is None:
            done = session._done_ev[id(agent)] = K.Event()
        self.done = done
        early = session.early_update and not self.lag   # (actor_lag = 1: measured no gain, 0.1147 vs 0.1132 ms per step)
        ins = agent._dense_call[3]      # hb_replay_insert's fixed arguments: 12 pointers, n, row bytes, n_actions, capacity
        act = fl.actor
        f = act._fset_ptrs[wset]
        if agent._support0 is None:
            agent._support0 = agent.atoms[0].contiguous()
        # ---- acting stream
        put(K.CMD_WAIT_EVENT, A, [agent.gathered_ev.h], cond=4)
        put(K.CMD_REPLAY_INSERT, A, ins[:12], ins[12:16], var=0)
        if early:
            inserted = session._inserted_ev.get(seat)
            if inserted is None:
                inserted = session._inserted_ev[seat] = K.Event()
            put(K.CMD_RECORD_EVENT, A, [inserted.h])
        # synchronous agent: the policy waits for the weights of this agent's last update; actor_lag = 1: for the launch that packed
        # the weight set it reads (update before last: long past)
        put(K.CMD_WAIT_EVENT, A, [(fl.packed_ev[wset] if self.lag else agent.weights_ev).h], cond=5)
        put(K.CMD_ACTOR_FUSED_ACT, A,
            [env.net_obs.data_ptr(), env.legal.data_ptr(), f[0], f[1], f[2], f[3], agent._support0.data_ptr(), act.q.data_ptr(),
             session._act_buf[seat].data_ptr()],
            [env.n, act.obs_len, act.hidden, act.n_actions, act.n_atoms, agent.params.seed + 0x9E3779B9, agent.first_game_id, act._dt],
            var=1, fvar=0)
        # `acted` (the policy has read its weights) is recorded BEFORE the env step unless HB_ACTED_BEFORE_ENV=0
        env_args = [env.h, session._act_buf[seat].data_ptr(), env.obs_bits.data_ptr(), None, env.legal.data_ptr(), env.reward.data_ptr(),
                    env.terminal.data_ptr(), env.agent_reward.data_ptr(), env.agent_step_type.data_ptr(), env.score.data_ptr()]
        if session._acted_early:
            put(K.CMD_RECORD_EVENT, A, [acted.h])
            put(K.CMD_ENV_STEP_PACKED, A, env_args)
        else:
            put(K.CMD_ENV_STEP_PACKED, A, env_args)
            put(K.CMD_RECORD_EVENT, A, [acted.h])
        # ---- learner stream. Early update: everything that only READS the weights starts as soon as the rows are in the ring —
        # beside this agent's own policy kernel — and only what WRITES them (the optimizer step and the actor's weight copies;
        # actor_lag: only the copies) waits for `acted`. Otherwise the whole update waits for `acted`.
        put(K.CMD_WAIT_EVENT, Ls, [(inserted if early else acted).h])
        put(K.CMD_TREE_FILL_RANGE, Ls, [buf.sum_tree.h, buf._max_priority.data_ptr()], var=2)
        g = fl._sg_call[3]              # hb_per_sample_gather's arguments in declaration order
        put(K.CMD_PER_SAMPLE_GATHER, Ls,
            [g[0], g[2], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[13], g[16], g[17], g[18], g[19], g[24]],
            [g[1], g[3], g[11], g[12], g[14], g[15], g[20], g[22], g[23]], [g[21]])
        put(K.CMD_RECORD_EVENT, Ls, [agent.gathered_ev.h])
        put(K.CMD_GRAPH_LAUNCH, Ls, [agent._graph1.raw_cuda_graph_exec()])
        if early:
            put(K.CMD_WAIT_EVENT, Ls, [acted.h])
        if agent._graph2 is not None:
            put(K.CMD_GRAPH_LAUNCH, Ls, [agent._graph2.raw_cuda_graph_exec()])
        if self.lag:
            # FusedLearner.weights_updated(): the update's result goes into the weight set the NEXT-but-one policy call reads
            # (both forms of the copies), followed by that set's event
            (w1, b1), (w2, b2) = fl.eff
            th = fl._thin_out()    # (pack_thin: the thin GEMMs' transposed online weights ride on this launch)
            put(K.CMD_ACTOR_FUSED_PACK, Ls, [w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), f[0], f[1], f[2], f[3]] +
                ([th[0].data_ptr(), th[2].data_ptr()] if th else [0, 0]),
                [w1.stride(0), w2.stride(0), act.obs_len, act.hidden, act.n_actions, act.n_atoms, act._dt] + ([th[1], th[3]] if th else [0, 0]))
            if act.two_kernel:   # (fp16 operands: the one-kernel form only)
                jobs = next(j for kk, j in act._jobs.items() if kk[0] == wset and kk[1] == w1.data_ptr())
                self._jobs = jobs   # (kept alive: the command holds its address)
                put(K.CMD_ACTOR_PACK_WEIGHTS, Ls, [C.addressof(jobs)], [2])
            put(K.CMD_RECORD_EVENT, Ls, [fl.packed_ev[wset].h])
        put(K.CMD_RECORD_EVENT, Ls, [agent.weights_ev.h])
        prios = agent._g_prios
        assert prios.dtype == torch.float32 and prios.is_contiguous() and agent._g_idx.dtype == torch.int64
        put(K.CMD_PER_UPDATE, Ls,
            [buf.sum_tree.h, agent._g_idx.data_ptr(), prios.data_ptr(), buf._max_priority.data_ptr(), buf._min_priority.data_ptr()],
            [agent._g_idx.numel()], [buf.alpha])
        self.cmds, self.count = cmds, n_cmd[0]
        self.vi = (C.c_int64 * 8)()
        self.vf = (C.c_double * 2)()
        self.run = K.lib().hb_chain_run
