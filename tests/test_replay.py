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
