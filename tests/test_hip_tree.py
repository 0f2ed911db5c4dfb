"""GPU parity: the HIP sum tree / PER kernels against the oracle, the reference golden vectors
(tests/golden/*.json, captured from the reference build) and the gtest known answers."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(cap):
    import hanabi_hip

    return hanabi_hip.SumTree(cap)


def dev(a, dt):
    import torch

    return torch.as_tensor(np.asarray(a, dtype=dt)).cuda()


def test_gtest_known_answers():  # sum_tree/sum_tree/src/test_sum_tree.cc:45-117
    assert T(7).get_capacity() == 8 and T(9).get_capacity() == 16
    t = T(4)
    t.update_value(0, 1.0)
    assert t.get_value(0) == 1 and t.get_total_val() == 1
    t.update_value(2, 2.0)
    assert t.get_total_val() == 3
    t.update_value(2, 1.0)
    assert t.get_value(2) == 1 and t.get_total_val() == 2
    t = T(4)
    t.update_values([0, 1, 2, 3], [1.0, 2.0, 3.0, 4.0])
    assert t.get_total_val() == 10 and t.get_values([0, 1, 2, 3]) == [1, 2, 3, 4]
    assert t.get_indices([0.0, 0.099, 0.1, 0.299, 0.3, 0.599, 0.6, 1.0]) == [0, 0, 1, 1, 2, 2, 3, 3]
    assert "capacity=4" in repr(t)
    with pytest.raises(IndexError):
        t.get_value(17)


def test_reference_golden_vectors():
    gold = json.load(open(os.path.join(GOLD, "sumtree_ref.json")))
    for case in gold["cases"]:
        t = T(case["requested_capacity"])
        assert t.get_capacity() == case["capacity"]
        for op in case["ops"]:
            t.update_values(op["update_idx"], op["update_val"])
            assert t.get_total_val() == op["total"]
            assert t.get_values(range(case["capacity"])) == op["leaves"]
            idx, val = t.sample_dev(dev(op["quantiles"], np.float32))
            assert idx.cpu().tolist() == op["sample_idx"]
            assert np.array_equal(val.cpu().numpy(), np.array(op["sample_val"], np.float32))
    t8 = T(8)
    t8.update_values([0, 1, 2, 3], [0.6] * 4)
    assert t8.get_index(1.0) == 7


@pytest.mark.parametrize("cap", [1, 2, 64, 1024, 4096, 2 ** 19])
def test_random_ops_match_oracle_bit_exact(cap):
    rng = np.random.default_rng(cap)
    t, o = T(cap), O.OracleTree(cap)
    for rnd in range(6):
        n = int(rng.integers(1, min(cap, 3000) + 1))
        if rnd % 2:
            n = min(n, 256)                                   # the learner-sized single-workgroup path
        idx = rng.integers(0, cap, n)                         # duplicates allowed: last one wins
        val = rng.random(n).astype(np.float32) * 3            # arbitrary floats: internal sums must round alike
        t.update_dev(dev(idx, np.int64), dev(val, np.float32))
        o.update(idx, val)
        assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:]), f"round {rnd}"
        q = np.concatenate([rng.random(300), [0.0, 1.0]]).astype(np.float32)
        gi, gv = t.sample_dev(dev(q, np.float32))
        oi, ov = o.sample(q)
        assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gv.cpu().numpy(), ov)
    assert t.error_count() == 0


def test_out_of_range_indices_are_ignored_and_counted():
    t, o = T(16), O.OracleTree(16)
    idx, val = [3, 99, -1, 3, 5], [1.0, 2.0, 3.0, 4.0, 5.0]
    t.update_values(idx, val)
    o.update(idx, val)
    assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:])
    assert t.error_count() == 2
    assert t.get_values([5, 100]) == [5.0, 0.0]


def test_small_update_of_a_deep_tree_duplicates_and_bad_indices():
    """The per-subtree path (update_chunks): many entries in one 1024-leaf subtree with repeated leaves, entries
    at subtree borders, out-of-range indices, n from 1 to 1024; PER mode tracks max / min over ALL entries."""
    import torch

    cap = 2 ** 15
    rng = np.random.default_rng(11)
    t, o = T(cap), O.OracleTree(cap)
    bad_total = 0
    for n in (1, 2, 63, 256, 257, 1000, 1024):
        idx = np.concatenate([rng.integers(2048, 3072, n // 2),             # one subtree, plenty of duplicates
                              rng.choice([0, 1023, 1024, 2047, cap - 1], n - n // 2)]).astype(np.int64)
        rng.shuffle(idx)
        if n >= 63:
            idx[5], idx[17] = -3, cap                                        # ignored, counted
            bad_total += 2
        val = (rng.random(n).astype(np.float32) + 0.01) * 2
        t.update_dev(dev(idx, np.int64), dev(val, np.float32))
        o.update(idx, val)
        assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:]), n
    assert t.error_count() == bad_total
    mx = torch.tensor([0.6], dtype=torch.float32, device="cuda")
    mn = mx.clone()
    omx = omn = float(np.float32(0.6))
    for n in (5, 256, 700):
        idx = rng.integers(0, cap, n).astype(np.int64)
        idx[: n // 3] = idx[n // 3: 2 * (n // 3)]                            # repeated leaves with different values
        td = (rng.standard_normal(n) * 3).astype(np.float32)
        t.per_update_dev(dev(idx, np.int64), dev(td, np.float32), 0.6, mx, mn)
        omx, omn = o.per_update(idx, td, 0.6, omx, omn)
        assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:]), n
        assert float(mx.cpu()[0]) == np.float32(omx) and float(mn.cpu()[0]) == np.float32(omn)


@pytest.mark.parametrize("cap,start,n", [(8, 6, 4), (2 ** 19, 2 ** 19 - 100, 32768), (2 ** 19, 0, 32768),
                                         (4096, 1000, 4096), (1024, 1023, 2), (512, 10, 100)])
def test_fill_range_matches_oracle(cap, start, n):
    import torch

    rng = np.random.default_rng(1)
    t, o = T(cap), O.OracleTree(cap)
    idx = rng.permutation(cap)[: cap // 2]
    val = rng.random(len(idx)).astype(np.float32)
    for a in range(0, len(idx), 1000):
        t.update_dev(dev(idx[a:a + 1000], np.int64), dev(val[a:a + 1000], np.float32))
    o.update(idx, val)
    v = torch.tensor([0.6], dtype=torch.float32, device="cuda")
    t.fill_range_dev(start, n, v)
    o.fill_range(start, n, np.float32(0.6))
    assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:])


def test_per_sample_and_update_match_oracle_and_reference_golden():
    import torch

    gold = json.load(open(os.path.join(GOLD, "replay_ref.json")))
    for case in gold["priority"]:
        cap = case["capacity"]
        t, o = T(cap), O.OracleTree(cap)
        mx = torch.tensor([case["max_priority0"]], dtype=torch.float32, device="cuda")
        mn = mx.clone()
        omx = omn = case["max_priority0"]
        oldest = 0
        for step in case["steps"]:
            n = len(step["batch"]["act"])
            t.fill_range_dev(oldest, n, mx)
            o.fill_range(oldest, n, np.float32(omx))
            oldest = (oldest + n) % cap
            u = np.array(step["uniforms"], np.float64)
            gi, gp = t.per_sample_dev(dev(u, np.float64))
            oi, op = o.per_sample(u)
            assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gp.cpu().numpy(), op)   # vs oracle: exact
            assert gi.cpu().tolist() == step["sample_indices"]                                       # vs reference
            assert np.allclose(gp.cpu().numpy(), step["sample_prios"], rtol=1e-6, atol=0)
            td = np.array(step["tds"], np.float32)
            t.per_update_dev(gi, dev(td, np.float32), case["alpha"], mx, mn)
            omx, omn = o.per_update(oi, td, case["alpha"], omx, omn)
            assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:])
            assert float(mx.cpu()[0]) == np.float32(omx) and float(mn.cpu()[0]) == np.float32(omn)
            ref = np.array(step["leaves_after_update"], np.float32)
            assert np.allclose(t.nodes().cpu().numpy()[cap:], ref, rtol=2e-7, atol=0)  # numpy powf vs pow: 1 ulp
            # resynchronise all three on the reference's exact leaves
            t.update_dev(dev(np.arange(cap), np.int64), dev(ref[:cap], np.float32))
            o.update(np.arange(cap), ref[:cap])
            omx, omn = step["max_priority"], step["min_priority"]
            mx.fill_(omx)
            mn.fill_(omn)
            omx, omn = float(np.float32(omx)), float(np.float32(omn))


@pytest.mark.parametrize("path", ["", "two"])
def test_learner_sized_per_cycle_at_full_capacity(path, monkeypatch):
    """cap 2^19, B=256: stratified sample -> update, 50 rounds, exact vs the oracle. Both forms of the small priority write-back
    (csrc/sum_tree.hip: the transform inside update_chunks, or per_transform + update_chunks<false>: HB_TREE_UPDATE_PATH=two)."""
    if path:
        monkeypatch.setenv("HB_TREE_UPDATE_PATH", path)
    rng = np.random.default_rng(3)
    cap = 2 ** 19
    t, o = T(cap), O.OracleTree(cap)
    import torch

    mx = torch.tensor([0.6], dtype=torch.float32, device="cuda")
    mn = mx.clone()
    omx = omn = float(np.float32(0.6))
    t.fill_range_dev(0, 300000, mx)
    o.fill_range(0, 300000, np.float32(0.6))
    for _ in range(50):
        u = rng.random(256) / 256
        gi, gp = t.per_sample_dev(dev(u, np.float64))
        oi, op = o.per_sample(u)
        assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gp.cpu().numpy(), op)
        td = (rng.standard_normal(256) * 2).astype(np.float32)
        t.per_update_dev(gi, dev(td, np.float32), 0.6, mx, mn)
        omx, omn = o.per_update(oi, td, 0.6, omx, omn)
    assert np.array_equal(t.nodes().cpu().numpy()[1:], o.nodes()[1:])
    assert float(mx.cpu()[0]) == np.float32(omx) and float(mn.cpu()[0]) == np.float32(omn)


def test_per_sample_philox_draws_match_host_philox():
    """hb_per_sample_philox == hb_per_sample / the oracle on uniforms rebuilt on the host from the same Philox
    words (bits53 = word0>>5 << 26 | word1>>6), for two counter values; strata stay stratified."""
    import torch

    cap, batch, seed = 4096, 256, 0xABCDEF12345
    t, o = T(cap), O.OracleTree(cap)
    rng = np.random.default_rng(3)
    vals = rng.random(cap).astype(np.float32) ** 2
    t.update_dev(dev(np.arange(cap), np.int64), dev(vals, np.float32))
    o.update(np.arange(cap), vals)
    for counter in (0.0, 7.0, 123456.0):
        c = torch.tensor(counter, dtype=torch.float32, device="cuda")
        gi, gp = t.per_sample_philox_dev(seed, c, batch)
        u = np.empty(batch, np.float64)
        for i in range(batch):
            w = O.philox([i, 0, int(counter), 0], [seed & 0xFFFFFFFF, seed >> 32])
            u[i] = float((int(w[0]) >> 5) << 26 | (int(w[1]) >> 6)) * 2.0 ** -53 / batch
        assert (u >= 0).all() and (u < 1.0 / batch).all()
        oi, op = o.per_sample(u)
        assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gp.cpu().numpy(), op)
        hi, hp = t.per_sample_dev(dev(u, np.float64))
        assert torch.equal(hi, gi) and torch.equal(hp, gp)
    a, _ = t.per_sample_philox_dev(seed, torch.tensor(1.0, device="cuda"), batch)
    b, _ = t.per_sample_philox_dev(seed, torch.tensor(2.0, device="cuda"), batch)
    assert not torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("packed,n_step", [(False, 1), (True, 1), (True, 3)])
def test_fused_sample_gather_equals_the_two_separate_launches(packed, n_step):
    """hb_per_sample_gather == hb_per_sample_philox followed by hb_replay_gather(_packed): same indices, probabilities and
    GEMM operand, bit for bit."""
    import torch

    from hanabi_agents.rlax_dqn.priority_buffer import PriorityBuffer
    from hanabi_hip import _capi as K

    L, A, n, cap, B = 658, 20, 64, 64 * 6, 128
    g = torch.Generator(device="cuda").manual_seed(7 + n_step)
    buf = PriorityBuffer(L, A, 1, cap, device="cuda", packed=packed)
    buf.track_wp = True
    for k in range(8):
        o1, o2 = ((torch.rand(n, L, device="cuda", generator=g) < 0.4).to(torch.int8) for _ in range(2))
        buf.add_transitions(o1, torch.randint(0, A, (n, 1), device="cuda", generator=g), torch.randint(-1, 3, (n, 1), device="cuda", generator=g).float(),
                            o2, torch.ones(n, A, dtype=torch.int8, device="cuda"), torch.rand(n, 1, device="cuda", generator=g) < 0.2)
        buf.update_priorities_dev(torch.randint(0, cap, (50,), device="cuda", generator=g), torch.rand(50, device="cuda", generator=g))
    buf.sync_size()
    counter = torch.tensor(5.0, device="cuda")
    lib, s = K.lib(), K.current_stream()
    outs = []
    for fused in (False, True):
        idx = torch.empty(B, dtype=torch.int64, device="cuda")
        prob = torch.empty(B, dtype=torch.float64, device="cuda")
        x = torch.zeros(2 * B, 704, dtype=torch.bfloat16, device="cuda")
        act = torch.empty(B, dtype=torch.int32, device="cuda")
        rew, term, disc = (torch.empty(B, device="cuda") for _ in range(3))
        if fused:
            K.check(lib.hb_per_sample_gather(buf.sum_tree.h, 99, K.dptr(counter), B, K.dptr(idx), K.dptr(prob), K.dptr(buf._obs_tm1_buf),
                                             K.dptr(buf._obs_t_buf), K.dptr(buf._act_tm1_buf), K.dptr(buf._rew_t_buf),
                                             K.dptr(buf._terminal_t_buf), L, 1 if packed else 0, K.dptr(x), 1, 704, K.dptr(act), K.dptr(rew),
                                             K.dptr(term), K.dptr(disc), n_step, 0.99, cap, n, K.dptr(buf._size_wp), s))
        else:
            K.check(lib.hb_per_sample_philox(buf.sum_tree.h, 99, K.dptr(counter), B, K.dptr(idx), K.dptr(prob), s))
            fn = lib.hb_replay_gather_packed if packed else lib.hb_replay_gather
            K.check(fn(K.dptr(buf._obs_tm1_buf), K.dptr(buf._obs_t_buf), K.dptr(buf._act_tm1_buf), K.dptr(buf._rew_t_buf),
                       K.dptr(buf._terminal_t_buf), K.dptr(idx), B, L, K.dptr(x), 1, 704, K.dptr(act), K.dptr(rew), K.dptr(term),
                       K.dptr(disc), n_step, 0.99, cap, n, K.dptr(buf._size_wp), s))
        outs.append((idx, prob, x, act, rew, term, disc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][2].any() and len(set(outs[0][0].tolist())) > B // 2


@pytest.mark.gpu
def test_lazy_top_levels_give_identical_results():
    import torch

    from hanabi_hip import _capi as K
    import hanabi_hip

    cap, B, L = 1 << 14, 128, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    trees = [hanabi_hip.SumTree(cap), hanabi_hip.SumTree(cap)]
    trees[1].set_lazy_top(True)
    mx, mn = [torch.tensor([0.6], device="cuda") for _ in trees], [torch.tensor([0.6], device="cuda") for _ in trees]
    ring = torch.zeros(cap, L, dtype=torch.int8, device="cuda")
    ring_act = torch.zeros(cap, 1, dtype=torch.int8, device="cuda")
    ring_rew = torch.zeros(cap, 1, device="cuda")
    ring_term = torch.zeros(cap, 1, dtype=torch.bool, device="cuda")
    size_wp = torch.tensor([cap, 0], dtype=torch.int64, device="cuda")
    counter = torch.tensor(1.0, device="cuda")
    lib, s = K.lib(), K.current_stream()
    for rnd in range(6):
        idx = torch.randint(0, cap, (256,), device="cuda", generator=g)
        td = torch.rand(256, device="cuda", generator=g)
        start = int(torch.randint(0, cap, (1,), generator=torch.Generator().manual_seed(rnd)).item())
        outs = []
        for t, a, b in zip(trees, mx, mn):
            t.fill_range_dev(start, 3000, a)              # ring insert (wraps for late starts)
            t.per_update_dev(idx, td, 0.6, a, b)          # 256 entries: the per-subtree path
            i_out = torch.empty(B, dtype=torch.int64, device="cuda")
            p_out = torch.empty(B, dtype=torch.float64, device="cuda")
            x = torch.zeros(2 * B, L, device="cuda")
            act = torch.empty(B, dtype=torch.int32, device="cuda")
            rew, term, disc = (torch.empty(B, device="cuda") for _ in range(3))
            K.check(lib.hb_per_sample_gather(t.h, 5, K.dptr(counter), B, K.dptr(i_out), K.dptr(p_out), K.dptr(ring), K.dptr(ring),
                                             K.dptr(ring_act), K.dptr(ring_rew), K.dptr(ring_term), L, 0, K.dptr(x), 0, L, K.dptr(act),
                                             K.dptr(rew), K.dptr(term), K.dptr(disc), 1, 0.99, cap, 1, K.dptr(size_wp), s))
            outs.append((i_out, p_out))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), rnd
        counter += 1
    u = torch.rand(B, dtype=torch.float64, device="cuda", generator=g)
    a, b = trees[0].per_sample_dev(u, unit=True), trees[1].per_sample_dev(u, unit=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert trees[0].get_total_val() == trees[1].get_total_val()
    assert torch.equal(trees[0].nodes(), trees[1].nodes())
