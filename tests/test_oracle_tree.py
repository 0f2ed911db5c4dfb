"""Pins the CPU oracle's flat sum tree and PER arithmetic against the reference:
 - the gtest known answers (sum_tree/sum_tree/src/test_sum_tree.cc:45-117),
 - the array-heap vectors of the reference's stale Python test (tests/rlax_dqn/test_sum_tree.py:26,38),
 - golden vectors captured from the reference build (tests/golden/gen_*_golden.py),
 - the live reference build oracle/_ref/libsumtree_ref.so when present.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_gtest_ctor_capacity_rounds_up():  # test_sum_tree.cc:45-51
    assert O.OracleTree(7).cap == 8 and O.OracleTree(9).cap == 16 and O.OracleTree(8).cap == 8
    t = O.OracleTree(8)
    assert t.total() == 0 and not t.nodes().any()


def test_gtest_update_value_overwrites():  # test_sum_tree.cc:81-92
    t = O.OracleTree(4)
    t.update([0], [1.0])
    assert t.leaves()[0] == 1 and t.total() == 1
    t.update([2], [2.0])
    assert t.leaves()[2] == 2 and t.total() == 3
    t.update([2], [1.0])
    assert t.leaves()[2] == 1 and t.total() == 2


def test_gtest_update_values_and_get_index():  # test_sum_tree.cc:94-117
    t = O.OracleTree(4)
    t.update([0, 1, 2, 3], [1.0, 2.0, 3.0, 4.0])
    assert t.total() == 10 and list(t.leaves()) == [1, 2, 3, 4]
    qs = [0.0, 0.099, 0.1, 0.299, 0.3, 0.599, 0.6, 1.0]
    assert list(t.sample(qs)[0]) == [0, 0, 1, 1, 2, 2, 3, 3]


def test_array_heap_vectors_of_reference_python_test():  # tests/rlax_dqn/test_sum_tree.py:26,38
    t = O.OracleTree(8)
    t.update(range(8), [1.0] * 8)
    assert list(t.nodes()[1:]) == [8, 4, 4, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1]
    t.update(range(8), range(8))
    assert list(t.nodes()[1:]) == [28, 6, 22, 1, 5, 9, 13, 0, 1, 2, 3, 4, 5, 6, 7]


def test_duplicate_indices_last_wins_and_out_of_range_ignored():
    t = O.OracleTree(8)
    t.update([3, 3, 3, 99, -1], [1.0, 5.0, 2.0, 7.0, 7.0])
    assert t.leaves()[3] == 2 and t.total() == 2


def test_against_reference_golden():
    gold = json.load(open(os.path.join(GOLD, "sumtree_ref.json")))
    assert gold["gtest"]["indices"] == [0, 0, 1, 1, 2, 2, 3, 3]
    for case in gold["cases"]:
        t = O.OracleTree(case["requested_capacity"])
        assert t.cap == case["capacity"]
        for op in case["ops"]:
            t.update(op["update_idx"], op["update_val"])
            assert t.total() == op["total"]
            assert np.array_equal(t.leaves(), np.array(op["leaves"], np.float32))
            idx, val = t.sample(op["quantiles"])
            assert list(idx) == op["sample_idx"]
            assert np.array_equal(val, np.array(op["sample_val"], np.float32))
    t8 = O.OracleTree(8)
    t8.update([0, 1, 2, 3], [0.6] * 4)
    assert t8.sample([1.0])[0][0] == gold["c8"]["index_of_1"] == 7


@pytest.mark.skipif(not O.RefTree.available(), reason="oracle/_ref not built")
def test_against_live_reference_build():
    rng = np.random.default_rng(11)
    for cap in (16, 1000, 4096):
        ref, mine = O.RefTree(cap), O.OracleTree(cap)
        assert ref.cap == mine.cap
        for _ in range(5):
            n = int(rng.integers(1, mine.cap))
            idx = rng.permutation(mine.cap)[:n]
            val = (rng.integers(0, 1000, n) / 32.0).astype(np.float32)
            ref.update(idx, val)
            mine.update(idx, val)
            assert ref.total() == mine.total()
            q = rng.random(257).astype(np.float32)
            ri, rv = ref.sample(q)
            mi, mv = mine.sample(q)
            assert np.array_equal(ri, mi) and np.array_equal(rv, mv)
        assert np.array_equal(ref.get(np.arange(mine.cap)), mine.leaves())


def test_fill_range_wraps():
    t = O.OracleTree(8)
    t.fill_range(6, 4, 0.5)
    assert list(np.flatnonzero(t.leaves())) == [0, 1, 6, 7] and t.total() == 2.0


def test_per_arithmetic_against_reference_priority_buffer_golden():
    """priority_buffer.py:29-52 replayed on the oracle tree: ring insert at max priority,
    stratified sampling on the captured uniforms, (|td|+1e-10)^alpha update."""
    gold = json.load(open(os.path.join(GOLD, "replay_ref.json")))
    for case in gold["priority"]:
        cap = case["capacity"]
        t = O.OracleTree(cap)
        assert t.cap == case["tree_capacity"]
        max_p = min_p = case["max_priority0"]
        assert max_p == case["alpha"] == 0.6
        oldest = 0
        for step in case["steps"]:
            n = len(step["batch"]["act"])
            t.fill_range(oldest, n, max_p)
            oldest = (oldest + n) % cap
            assert oldest == step["oldest_entry"]
            assert np.array_equal(t.leaves(), np.array(step["leaves_after_add"], np.float32))
            # total: the reference accumulates float differences, the oracle rebuilds exactly (App. C-9)
            assert t.total() == pytest.approx(step["total_after_add"], rel=1e-6)
            idx, prob = t.per_sample(step["uniforms"])
            assert list(idx) == step["sample_indices"]
            assert np.allclose(prob, step["sample_prios"], rtol=1e-6, atol=0)
            max_p, min_p = t.per_update(idx, step["tds"], case["alpha"], max_p, min_p)
            assert max_p == pytest.approx(step["max_priority"], rel=2e-7)
            assert min_p == pytest.approx(step["min_priority"], rel=2e-7)
            # float32 pow: numpy's powf vs double pow rounded once may differ by 1 ulp
            ref_leaves = np.array(step["leaves_after_update"], np.float32)
            assert np.allclose(t.leaves(), ref_leaves, rtol=2e-7, atol=0)
            # continue from the reference's exact leaves so a 1-ulp pow difference cannot propagate
            t.update(np.arange(cap), ref_leaves[:cap])
            max_p, min_p = step["max_priority"], step["min_priority"]
