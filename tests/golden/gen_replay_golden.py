"""Generates tests/golden/replay_ref.json by RUNNING the reference's own ExperienceBuffer /
PriorityBuffer (hanabi_agents/rlax_dqn/{experience_buffer,priority_buffer}.py) in this
container, on top of the reference SumTreef built into oracle/_ref by oracle/Makefile.

Only this script touches /root/reference; it runs here (never on the GPU box) and only its
OUTPUT (inputs + expected outputs, as JSON) is committed.

The two modules are loaded as files, bypassing hanabi_agents/rlax_dqn/__init__.py (which
imports the jax/haiku learner that cannot be imported here). Two names the files mention
but whose functionality the live classes never use are provided so the module bodies
execute: `np.int` (alias removed from numpy 2; experience_buffer.py:15) and an inert `jax`
module object (priority_buffer.py:3-5 imports it; the live class priority_buffer.py:13-52
calls nothing from it).
"""
import importlib.util
import json
import os

# One OpenMP thread: with duplicate indices in one update_values call the reference's parallel
# loop applies the leaf difference twice (read of the old leaf outside the lock, sum_tree.h:32-35;
# SURVEY §5 "latent race"). Sequential execution is the behaviour the goldens pin.
os.environ["OMP_NUM_THREADS"] = "1"
import sys
import types

import numpy as np

REF = "/root/reference/hanabi_agents/rlax_dqn"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))  # reference sum_tree pybind module


def load_reference():
    np.int = int
    jax = types.ModuleType("jax")
    jax.numpy = types.ModuleType("jax.numpy")
    jax.tree_util = types.ModuleType("jax.tree_util")
    for n in ("tree_flatten", "tree_unflatten"):
        setattr(jax.tree_util, n, None)
    jax.tree_util.register_pytree_node_class = lambda c: c
    sys.modules.update({"jax": jax, "jax.numpy": jax.numpy, "jax.tree_util": jax.tree_util})
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = [REF]
    sys.modules["refpkg"] = pkg
    mods = {}
    for name in ("transition", "experience_buffer", "priority_buffer"):
        spec = importlib.util.spec_from_file_location("refpkg." + name, os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["refpkg." + name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def batch(rng, n, obs_len, n_act):
    return dict(
        obs_tm1=rng.integers(0, 2, (n, obs_len)).astype(np.int8),
        act=rng.integers(0, n_act, (n, 1)).astype(np.int8),
        rew=rng.integers(-3, 4, (n, 1)).astype(np.float64),
        obs_t=rng.integers(0, 2, (n, obs_len)).astype(np.int8),
        lms=rng.integers(0, 2, (n, n_act)).astype(np.int8),
        term=rng.integers(0, 2, (n, 1)).astype(bool),
    )


def add(buf, b):
    buf.add_transitions(b["obs_tm1"], b["act"], b["rew"], b["obs_t"], b["lms"], b["term"])


def tolist(b):
    return {k: v.astype(int).tolist() if v.dtype != np.float64 else v.tolist() for k, v in b.items()}


def transition_lists(t):
    return dict(obs_tm1=t.observation_tm1.astype(int).tolist(), act=t.action_tm1.astype(int).tolist(),
                rew=t.reward_t.tolist(), obs_t=t.observation_t.astype(int).tolist(),
                lms=t.legal_moves_t.astype(int).tolist(), term=t.terminal_t.astype(int).tolist())


def main():
    mods = load_reference()
    EB = mods["experience_buffer"].ExperienceBuffer
    PB = mods["priority_buffer"].PriorityBuffer
    rng = np.random.default_rng(2026)
    out = {"experience": [], "priority": []}

    # --- ExperienceBuffer: plain add, exact fill, wrap-around (tests/rlax_dqn/test_experience_buffer.py:23-142)
    for cap, sizes in [(7, [3, 2]), (7, [7]), (7, [8]), (8, [5, 5, 5]), (6, [4, 2, 3])]:
        obs_len, n_act = 5, 4
        buf = EB(obs_len, n_act, 1, cap)
        case = dict(capacity=cap, obs_len=obs_len, n_act=n_act, steps=[])
        for n in sizes:
            upd = buf.get_update_indices(n)
            b = batch(rng, n, obs_len, n_act)
            add(buf, b)
            idx = list(range(buf.size))
            case["steps"].append(dict(batch=tolist(b), update_indices=upd, oldest_entry=buf.oldest_entry, size=buf.size,
                                      content=transition_lists(buf[np.array(idx, dtype=int)])))
        out["experience"].append(case)

    # --- PriorityBuffer: insert (incl. wrap), stratified sampling on given uniforms, priority update
    for cap, sizes, B in [(8, [3, 4, 3], 4), (16, [10, 10], 8), (64, [40, 40], 16)]:
        obs_len, n_act = 4, 3
        buf = PB(obs_len, n_act, 1, cap)
        case = dict(capacity=cap, obs_len=obs_len, n_act=n_act, alpha=buf.alpha, tree_capacity=buf.sum_tree.get_capacity(),
                    max_priority0=buf.max_priority, steps=[])
        for n in sizes:
            b = batch(rng, n, obs_len, n_act)
            add(buf, b)
            leaves = buf.sum_tree.get_values(list(range(buf.sum_tree.get_capacity())))
            step = dict(batch=tolist(b), oldest_entry=buf.oldest_entry, size=buf.size, leaves_after_add=leaves,
                        total_after_add=buf.sum_tree.get_total_val())
            # sample_batch draws onp.random.uniform(size=(B,), high=1/B): capture those exact uniforms
            np.random.seed(1000 + n + cap)
            u = np.random.uniform(size=(B,), high=1.0 / B)
            np.random.seed(1000 + n + cap)
            indices, prios, trans = buf.sample_batch(B)
            step.update(uniforms=u.tolist(), sample_indices=[int(i) for i in indices], sample_prios=[float(p) for p in prios],
                        sample_obs_t=trans.observation_t.astype(int).tolist())
            # dyadic |td| values keep every float sum exact, so tree internals are representation-independent
            # (a transition sampled twice yields the same |td|, so duplicates carry equal values: the
            # reference's OpenMP update loop is order-dependent for unequal duplicates, sum_tree.h:38-44)
            per_slot = (rng.integers(1, 64, cap) / 16.0).astype(np.float32)
            tds = per_slot[np.asarray(indices, dtype=int)]
            buf.update_priorities(indices, np.abs(tds))
            step.update(tds=tds.tolist(), max_priority=float(buf.max_priority), min_priority=float(buf.min_priority),
                        leaves_after_update=buf.sum_tree.get_values(list(range(buf.sum_tree.get_capacity()))),
                        total_after_update=buf.sum_tree.get_total_val())
            case["steps"].append(step)
        out["priority"].append(case)

    with open(os.path.join(HERE, "replay_ref.json"), "w") as f:
        json.dump(out, f)
    print("wrote replay_ref.json", os.path.getsize(os.path.join(HERE, "replay_ref.json")), "bytes")


if __name__ == "__main__":
    main()
