"""Replay ring / PER host logic against golden vectors captured from the reference's own classes
(tests/golden/replay_ref.json <- tests/golden/gen_replay_golden.py)."""
import json
import os

import numpy as np
import pytest

from hanabi_agents.rlax_dqn.experience_buffer import ExperienceBuffer

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay_ref.json")))


def _add(buf, b):
    buf.add_transitions(np.array(b["obs_tm1"], np.int8), np.array(b["act"], np.int8), np.array(b["rew"], np.float64),
                        np.array(b["obs_t"], np.int8), np.array(b["lms"], np.int8), np.array(b["term"], bool))


def _check_content(buf, content, n):
    t = buf[np.arange(n)]
    assert np.array_equal(t.observation_tm1, np.array(content["obs_tm1"], np.int8))
    assert np.array_equal(t.action_tm1, np.array(content["act"], np.int8))
    assert np.array_equal(t.reward_t, np.array(content["rew"], np.float64))
    assert np.array_equal(t.observation_t, np.array(content["obs_t"], np.int8))
    assert np.array_equal(t.legal_moves_t, np.array(content["lms"], np.int8))
    assert np.array_equal(t.terminal_t, np.array(content["term"], bool))


def test_experience_buffer_matches_reference_golden():
    """plain add, exact fill, wrap-around (tests/rlax_dqn/test_experience_buffer.py:23-142 cases)."""
    for case in GOLD["experience"]:
        buf = ExperienceBuffer(case["obs_len"], case["n_act"], 1, case["capacity"], device="cpu")
        assert buf._obs_tm1_buf.shape == (case["capacity"], case["obs_len"]) and buf._lms_t_buf.shape[1] == case["n_act"]
        for step in case["steps"]:
            assert buf.get_update_indices(len(step["batch"]["act"])) == step["update_indices"]
            _add(buf, step["batch"])
            assert (buf.oldest_entry, buf.size) == (step["oldest_entry"], step["size"])
            _check_content(buf, step["content"], step["size"])


def test_uniform_sample_with_replacement_within_size():
    buf = ExperienceBuffer(3, 2, 1, 10, device="cpu", seed=1)
    with pytest.raises(ValueError):
        buf.sample(4)
    buf.add_transitions(np.arange(12).reshape(4, 3) % 2, np.zeros((4, 1)), np.arange(4.0).reshape(4, 1), np.zeros((4, 3)),
                        np.ones((4, 2)), np.zeros((4, 1), bool))
    idx = buf.sample_indices_dev(1000)
    assert int(idx.min()) >= 0 and int(idx.max()) <= 3 and len(set(idx.tolist())) == 4
    t = buf.sample(16)
    assert t.observation_tm1.shape == (16, 3) and set(t.reward_t[:, 0]) <= {0.0, 1.0, 2.0, 3.0}


@pytest.mark.gpu
def test_priority_buffer_matches_reference_golden_on_gpu():
    """insert (incl. wrap) / stratified sample on the captured uniforms / priority update of the reference's
    PriorityBuffer, replayed on the GPU ring + HIP sum tree."""
    from hanabi_agents.rlax_dqn import PriorityBuffer

    for case in GOLD["priority"]:
        buf = PriorityBuffer(case["obs_len"], case["n_act"], 1, case["capacity"], device="cuda")
        assert buf.sum_tree.get_capacity() == case["tree_capacity"] and buf.max_priority == pytest.approx(0.6)
        for step in case["steps"]:
            _add(buf, step["batch"])
            assert (buf.oldest_entry, buf.size) == (step["oldest_entry"], step["size"])
            leaves = buf.sum_tree.get_values(range(case["tree_capacity"]))
            assert np.allclose(leaves, step["leaves_after_add"], rtol=2e-7, atol=0)
            idx, prios, tr = buf.sample_batch(len(step["uniforms"]), uniforms=np.array(step["uniforms"]))
            assert idx == step["sample_indices"]
            assert np.allclose(prios, step["sample_prios"], rtol=1e-6, atol=0)
            assert np.array_equal(tr.observation_t, np.array(step["sample_obs_t"], np.int8))
            buf.update_priorities(idx, np.abs(np.array(step["tds"], np.float32)))
            assert buf.max_priority == pytest.approx(step["max_priority"], rel=2e-7)
            assert buf.min_priority == pytest.approx(step["min_priority"], rel=2e-7)
            assert np.allclose(buf.sum_tree.get_values(range(case["tree_capacity"])), step["leaves_after_update"], rtol=2e-7, atol=0)


def _nstep_reference(rew, term, n_ins, cap, size, wp, i, n, gamma):
    """Plain-Python n-step walk (hanabi_agents/rainbow/replay_memory.py:316-345 semantics on our ring)."""
    ahead = (wp - 1 - i) % cap if size >= cap else size - 1 - i
    j, R, g, m = i, rew[i], gamma, 1
    while m < n and not term[j] and m * n_ins <= ahead:
        j = (j + n_ins) % cap
        R += g * rew[j]
        g *= gamma
        m += 1
    return j, R, g


@pytest.mark.parametrize("cap,n_ins,inserts", [(40, 8, 3), (40, 8, 5), (40, 8, 11), (37, 5, 20)])
def test_nstep_gather_matches_python_walk(cap, n_ins, inserts):
    import torch

    rng = np.random.default_rng(cap + inserts)
    buf = ExperienceBuffer(6, 3, 1, cap, device="cpu")
    for k in range(inserts):
        buf.add_transitions(rng.integers(0, 2, (n_ins, 6)), rng.integers(0, 3, (n_ins, 1)), rng.integers(-2, 3, (n_ins, 1)).astype(float),
                            np.full((n_ins, 6), k % 2), np.ones((n_ins, 3)), rng.random((n_ins, 1)) < 0.25)
    assert buf.rows_per_insert == n_ins
    rew = buf._rew_t_buf[:, 0].numpy().astype(np.float64)
    term = buf._terminal_t_buf[:, 0].numpy()
    idx = torch.arange(buf.size)
    for n in (1, 2, 3, 5):
        tr, disc = buf.gather_nstep_dev(idx, n, 0.9)
        for i in range(buf.size):
            j, R, g = _nstep_reference(rew, term, n_ins, cap, buf.size, buf.oldest_entry, i, n, 0.9)
            assert tr.reward_t[i, 0].item() == pytest.approx(R, rel=1e-6, abs=1e-6)
            assert disc[i].item() == pytest.approx(g, rel=1e-6)
            assert bool(tr.terminal_t[i, 0]) == bool(term[j])
            assert torch.equal(tr.observation_t[i], buf._obs_t_buf[j]) and torch.equal(tr.observation_tm1[i], buf._obs_tm1_buf[i])
    buf.add_transitions(np.zeros((3, 6)), np.zeros((3, 1)), np.zeros((3, 1)), np.zeros((3, 6)), np.ones((3, 3)), np.zeros((3, 1), bool))
    assert buf.rows_per_insert == -1
    with pytest.raises(ValueError):
        buf.gather_nstep_dev(idx[:4], 3, 0.9)
