"""RulebasedAgent — the reference's rule-walking agent (hanabi_agents/rule_based/rule_based.py:5-48) on the GPU.

Same surface: `RulebasedAgent(rules)`, `explore` / `exploit` (identical), `requires_vectorized_observation()`
False, no-op `add_experience_first` / `add_experience` / `update`, and the `histogram` / `totalCalls` counters of
which rule fired. The reference's session hands a non-vectorised agent the rich per-game observation objects; here
`observations[0]` is the `hanabi_hip.HanabiEnv` itself (hanabi_hip.selfplay passes it), whose state rows ARE
those observations: the agent reads them in place with one `hb_rule_act` launch and returns a device tensor of
move uids. Python's `random` is replaced by Philox keyed by (seed; call counter, global game id).
"""
import ctypes as C

import torch

from hanabi_hip import _capi as K

from .ruleset import Rule


class RulebasedAgent:
    def __init__(self, rules, seed=4321):
        rules = list(rules)
        if len(rules) > K.MAX_RULES:
            raise ValueError(f"at most {K.MAX_RULES} rules")
        for r in rules:
            if not isinstance(r, Rule):
                raise TypeError(f"{r!r} is not a Ruleset rule")
        self.rules = rules
        self._tab = (K.HbRule * max(len(rules), 1))()
        for i, r in enumerate(rules):
            self._tab[i].kind, self._tab[i].arg, self._tab[i].threshold = r.kind, r.arg, r.threshold
        self.seed = int(seed)
        self._draws = 0
        self._hist = None      # device int64 [len(rules) + 1]: how often each rule fired, last = random fallback
        self._fired = None
        self._actions = None

    # ---- acting (rule_based.py:13-34) -------------------------------------------------------------------------
    def get_moves(self, env):
        """Move uids int32 [N] (device) for the player to act in every game of `env`."""
        n = env.n
        if self._actions is None or self._actions.numel() != n or self._actions.device != env.device:
            self._actions = torch.empty(n, dtype=torch.int32, device=env.device)
            self._fired = torch.empty(n, dtype=torch.int32, device=env.device)
            self._hist = torch.zeros(len(self.rules) + 1, dtype=torch.int64, device=env.device)
            if getattr(self, "_hist_restore", None):   # counts from a checkpoint loaded before the first call
                self._hist.copy_(torch.tensor(self._hist_restore, dtype=torch.int64))
        self._draws += 1
        L = K.lib()
        K.check(L.hb_rule_act(C.byref(env.cfg), L.hb_env_state(env.h), n, env.first_game_id, self._tab, len(self.rules),
                              self.seed, self._draws, K.dptr(self._actions), K.dptr(self._fired), K.current_stream()))
        self._hist += torch.bincount(self._fired, minlength=len(self.rules) + 1)
        return self._actions

    def explore(self, observations):
        env = observations[0] if isinstance(observations, (tuple, list)) else observations
        if not hasattr(env, "h"):
            raise TypeError("RulebasedAgent reads the env's state rows: pass (env, (obs, legal)) or the HanabiEnv itself")
        return self.get_moves(env)

    def exploit(self, observations):
        return self.explore(observations)

    def requires_vectorized_observation(self):
        return False

    # ---- passive parts of the agent protocol (rule_based.py:40-48) ---------------------------------------------
    def add_experience_first(self, o, st):
        pass

    def add_experience(self, o, a, r, st):
        pass

    def add_experience_dense(self, o, a, r, st):
        pass

    def update(self):
        pass

    # ---- checkpointing (hanabi_hip.selfplay) ------------------------------------------------------------------------
    def checkpoint_state(self, include_replay=True):
        return dict(format="hanabi-agents_amd/rule_agent/1", seed=self.seed, draws=self._draws, histogram=self.histogram,
                    rules=[(r.kind, r.arg, r.threshold) for r in self.rules])

    def load_checkpoint_state(self, sd):
        if sd.get("format") != "hanabi-agents_amd/rule_agent/1":
            raise ValueError("not a rule-agent checkpoint")
        if [tuple(x) for x in sd["rules"]] != [(r.kind, r.arg, r.threshold) for r in self.rules]:
            raise ValueError("checkpoint was written by a different rule list")
        self.seed, self._draws = int(sd["seed"]), int(sd["draws"])
        if self._hist is not None:
            self._hist.copy_(torch.tensor(sd["histogram"], dtype=torch.int64))
        self._hist_restore = list(sd["histogram"])

    # ---- rule statistics (rule_based.py:9-11) ---------------------------------------------------------------------
    @property
    def histogram(self):
        if self._hist is None:
            return list(getattr(self, "_hist_restore", None) or [0] * (len(self.rules) + 1))
        return self._hist.cpu().tolist()

    @property
    def totalCalls(self):
        return sum(self.histogram)
