"""ctypes front-end of the CPU oracle (oracle/liboracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package never does (tests/test_no_oracle_in_product.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HbConfig(C.Structure):
    """Mirror of `hb_config` (include/hanabi_hip.h)."""

    _fields_ = [(n, C.c_int32) for n in ("players", "colors", "ranks", "hand_size", "max_info", "max_life", "flags")]


FLAG_AUTO_RESET, FLAG_RESET_START_NEXT, FLAG_LENIENT_REWARD = 1, 2, 4

GAME_TYPES = {
    # name -> (colors, ranks, hand_size(players), max_info, max_life); SURVEY App. A.1
    "Hanabi-Full": (5, 5, lambda p: 5 if p < 4 else 4, 8, 3),
    "Hanabi-Small": (2, 5, lambda p: 2, 3, 1),
    "Hanabi-Very-Small": (1, 5, lambda p: 2, 3, 1),
}


def make_config(game="Hanabi-Full", players=2, flags=0):
    colors, ranks, hs, info, life = GAME_TYPES[game]
    return HbConfig(players, colors, ranks, hs(players), info, life, flags)


def build(force=False):
    # HB_ORACLE_LIB: another build of the same sources, e.g. the sanitizer build `make -C oracle asan` (oracle/Makefile)
    if os.environ.get("HB_ORACLE_LIB"):
        return os.environ["HB_ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hanabi_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        p = C.c_void_p
        cfgp = C.POINTER(HbConfig)
        L.orc_env_create.restype = p
        L.orc_env_create.argtypes = [cfgp, C.c_int64, C.c_uint64, C.c_int64]
        L.orc_env_destroy.argtypes = [p]
        L.orc_env_set_decks.argtypes = [p, p]
        L.orc_env_reset.argtypes = [p, p, C.c_int32]
        L.orc_env_observe.argtypes = [p] + [p] * 4
        L.orc_env_step.argtypes = [p] + [p] * 8
        L.orc_env_illegal_count.restype = C.c_int64
        L.orc_env_illegal_count.argtypes = [p]
        L.orc_env_export_state.argtypes = [p, p]
        L.orc_env_set_threads.argtypes = [p, C.c_int]
        L.orc_random_legal_actions.argtypes = [p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint64, C.c_int64, p]
        L.orc_rule_act.argtypes = [p, p, C.c_int32, C.c_uint64, C.c_uint64, p, p]
        L.orc_philox4x32.argtypes = [p, p, p]
        L.orc_shuffled_deck.argtypes = [cfgp, C.c_uint64, C.c_uint64, C.c_uint32, p]
        for f in ("orc_num_actions", "orc_obs_len", "orc_deck_size", "orc_state_words"):
            getattr(L, f).argtypes = [cfgp]
            getattr(L, f).restype = C.c_int
        L.orc_tree_create.restype = p
        L.orc_tree_create.argtypes = [C.c_int64]
        L.orc_tree_destroy.argtypes = [p]
        L.orc_tree_capacity.restype = C.c_int64
        L.orc_tree_capacity.argtypes = [p]
        L.orc_tree_nodes.restype = C.POINTER(C.c_float)
        L.orc_tree_nodes.argtypes = [p]
        L.orc_tree_update.argtypes = [p, p, p, C.c_int64]
        L.orc_tree_fill_range.argtypes = [p, C.c_int64, C.c_int64, C.c_float]
        L.orc_tree_sample.argtypes = [p, p, p, p, C.c_int64]
        L.orc_tree_get.argtypes = [p, p, p, C.c_int64]
        L.orc_tree_total.restype = C.c_float
        L.orc_tree_total.argtypes = [p]
        L.orc_per_sample.argtypes = [p, p, C.c_int64, p, p]
        L.orc_per_update.argtypes = [p, p, p, C.c_int64, C.c_double, p, p]
        _LIB = L
    return _LIB


class HbRule(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arg", C.c_int32), ("threshold", C.c_float)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """N Hanabi games stepped one at a time on the CPU; same call shape as hanabi_hip.HanabiEnv."""

    def __init__(self, cfg, n_games, seed=1234, first_game_id=0, decks=None, start_player=0, threads=1):
        self.L = lib()
        self.cfg = cfg
        self.n = int(n_games)
        self.h = self.L.orc_env_create(C.byref(cfg), self.n, seed, first_game_id)
        self.num_actions = self.L.orc_num_actions(C.byref(cfg))
        self.obs_len = self.L.orc_obs_len(C.byref(cfg))
        self.deck_size = self.L.orc_deck_size(C.byref(cfg))
        self.state_words = self.L.orc_state_words(C.byref(cfg))
        self.L.orc_env_set_threads(self.h, threads)
        if decks is not None:
            self.set_decks(decks)
        self.reset(start_player=start_player)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_env_destroy(self.h)
            self.h = None

    def set_decks(self, decks):
        if decks is None:
            self.L.orc_env_set_decks(self.h, None)
            return
        d = np.ascontiguousarray(decks, dtype=np.uint8).reshape(self.n, self.deck_size)
        self.L.orc_env_set_decks(self.h, _ptr(d))

    def reset(self, mask=None, start_player=0):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.orc_env_reset(self.h, _ptr(m), start_player)

    def _bufs(self):
        return (np.empty((self.n, self.obs_len), np.int8), np.empty((self.n, self.num_actions), np.int8))

    def observe(self):
        obs, legal = self._bufs()
        ar, st = np.empty(self.n, np.float32), np.empty(self.n, np.int8)
        self.L.orc_env_observe(self.h, _ptr(obs), _ptr(legal), _ptr(ar), _ptr(st))
        return dict(obs=obs, legal=legal, agent_reward=ar, agent_step_type=st)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.n,)
        obs, legal = self._bufs()
        rew, ar = np.empty(self.n, np.float32), np.empty(self.n, np.float32)
        term, st, sc = (np.empty(self.n, np.int8) for _ in range(3))
        self.L.orc_env_step(self.h, _ptr(a), _ptr(obs), _ptr(legal), _ptr(rew), _ptr(term), _ptr(ar), _ptr(st), _ptr(sc))
        return dict(obs=obs, legal=legal, reward=rew, terminal=term, agent_reward=ar, agent_step_type=st, score=sc)

    def step_noobs(self, actions):
        """Step without encoding (used by the cpu_baseline to time the rules alone)."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        self.L.orc_env_step(self.h, _ptr(a), None, None, None, None, None, None, None)

    def illegal_count(self):
        return int(self.L.orc_env_illegal_count(self.h))

    def rule_act(self, rules, seed, draw):
        """rules: [(kind, arg, threshold)]. Returns (actions int32 [N], fired int32 [N]) of rule_oracle.c."""
        tab = (HbRule * max(len(rules), 1))()
        for i, (kind, arg, thr) in enumerate(rules):
            tab[i].kind, tab[i].arg, tab[i].threshold = kind, arg, thr
        act, fired = np.empty(self.n, np.int32), np.empty(self.n, np.int32)
        self.L.orc_rule_act(self.h, tab, len(rules), seed, draw, _ptr(act), _ptr(fired))
        return act, fired

    def export_state(self):
        rows = np.zeros((self.n, self.state_words), np.uint32)
        self.L.orc_env_export_state(self.h, _ptr(rows))
        return rows


def random_legal_actions(legal, seed, draw, first_game_id=0):
    legal = np.ascontiguousarray(legal, dtype=np.int8)
    n, a = legal.shape
    out = np.empty(n, np.int32)
    lib().orc_random_legal_actions(_ptr(legal), n, a, seed, draw, first_game_id, _ptr(out))
    return out


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().orc_philox4x32(_ptr(c), _ptr(k), _ptr(o))
    return o


def shuffled_deck(cfg, seed, game_id, episode):
    d = np.zeros(lib().orc_deck_size(C.byref(cfg)), np.uint8)
    lib().orc_shuffled_deck(C.byref(cfg), seed, game_id, episode, _ptr(d))
    return d


class OracleTree:
    """Flat-array sum tree with the semantics of the reference SumTreef."""

    def __init__(self, capacity):
        self.L = lib()
        self.h = self.L.orc_tree_create(capacity)
        self.cap = self.L.orc_tree_capacity(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_tree_destroy(self.h)
            self.h = None

    def nodes(self):
        return np.ctypeslib.as_array(self.L.orc_tree_nodes(self.h), shape=(2 * self.cap,)).copy()

    def leaves(self):
        return self.nodes()[self.cap:]

    def update(self, idx, val):
        i = np.ascontiguousarray(idx, np.int64)
        v = np.ascontiguousarray(val, np.float32)
        self.L.orc_tree_update(self.h, _ptr(i), _ptr(v), len(i))

    def fill_range(self, start, n, value):
        self.L.orc_tree_fill_range(self.h, start, n, value)

    def sample(self, q):
        q = np.ascontiguousarray(q, np.float32)
        idx, val = np.empty(len(q), np.int64), np.empty(len(q), np.float32)
        self.L.orc_tree_sample(self.h, _ptr(q), _ptr(idx), _ptr(val), len(q))
        return idx, val

    def get(self, idx):
        i = np.ascontiguousarray(idx, np.int64)
        v = np.empty(len(i), np.float32)
        self.L.orc_tree_get(self.h, _ptr(i), _ptr(v), len(i))
        return v

    def total(self):
        return float(self.L.orc_tree_total(self.h))

    def per_sample(self, u):
        u = np.ascontiguousarray(u, np.float64)
        idx, prob = np.empty(len(u), np.int64), np.empty(len(u), np.float64)
        self.L.orc_per_sample(self.h, _ptr(u), len(u), _ptr(idx), _ptr(prob))
        return idx, prob

    def per_update(self, idx, td, alpha, max_prio, min_prio):
        i = np.ascontiguousarray(idx, np.int64)
        t = np.ascontiguousarray(td, np.float32)
        mx, mn = np.array([max_prio], np.float32), np.array([min_prio], np.float32)
        self.L.orc_per_update(self.h, _ptr(i), _ptr(t), len(i), alpha, _ptr(mx), _ptr(mn))
        return float(mx[0]), float(mn[0])


class RefTree:
    """The REFERENCE SumTree<float> (oracle/_ref/libsumtree_ref.so, built by oracle/Makefile)."""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libsumtree_ref.so"))

    def __init__(self, capacity):
        L = C.CDLL(os.path.join(_HERE, "_ref", "libsumtree_ref.so"))
        p = C.c_void_p
        L.ref_tree_create.restype = p
        L.ref_tree_create.argtypes = [C.c_int64]
        L.ref_tree_destroy.argtypes = [p]
        L.ref_tree_capacity.restype = C.c_int64
        L.ref_tree_capacity.argtypes = [p]
        L.ref_tree_total.restype = C.c_float
        L.ref_tree_total.argtypes = [p]
        L.ref_tree_update_value.argtypes = [p, C.c_int64, C.c_float]
        L.ref_tree_update_values.argtypes = [p, p, p, C.c_int64]
        L.ref_tree_get_index.restype = C.c_int64
        L.ref_tree_get_index.argtypes = [p, C.c_float]
        L.ref_tree_get_indices.argtypes = [p, p, p, C.c_int64]
        L.ref_tree_get_values.argtypes = [p, p, p, C.c_int64]
        self.L = L
        self.h = L.ref_tree_create(capacity)
        self.cap = L.ref_tree_capacity(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_tree_destroy(self.h)
            self.h = None

    def update_value(self, i, v):
        self.L.ref_tree_update_value(self.h, i, v)

    def update(self, idx, val):
        i = np.ascontiguousarray(idx, np.int64)
        v = np.ascontiguousarray(val, np.float32)
        self.L.ref_tree_update_values(self.h, _ptr(i), _ptr(v), len(i))

    def get_index(self, q):
        return int(self.L.ref_tree_get_index(self.h, q))

    def sample(self, q):
        q = np.ascontiguousarray(q, np.float32)
        idx = np.empty(len(q), np.int64)
        self.L.ref_tree_get_indices(self.h, _ptr(q), _ptr(idx), len(q))
        return idx, self.get(idx)

    def get(self, idx):
        i = np.ascontiguousarray(idx, np.int64)
        v = np.empty(len(i), np.float32)
        self.L.ref_tree_get_values(self.h, _ptr(i), _ptr(v), len(i))
        return v

    def total(self):
        return float(self.L.ref_tree_total(self.h))
