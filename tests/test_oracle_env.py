"""Pins the CPU oracle of the Hanabi env/encoder (oracle/hanabi_oracle.c).

The upstream env is absent from the reference, so these are hand-worked
known-answer tests and invariants from SURVEY.md App. A.6/A.7 — "parity unpinned" against
upstream HLE, pinned against the written spec.
"""
import numpy as np
import pytest

from oracle import oracle_py as O

FULL2 = dict(hands=(0, 127), board=(127, 203), discards=(203, 253), last=(253, 308), know=(308, 658))


def canonical_deck(cfg):
    deck = []
    for c in range(cfg.colors):
        for r in range(cfg.ranks):
            copies = 3 if r == 0 else (1 if r == cfg.ranks - 1 else 2)
            deck += [c * cfg.ranks + r] * copies
    return np.array(deck, np.uint8)


def test_sizes():
    # obs 658 / 1280 / 171, actions 20 / 48 / 11
    for game, players, obs_len, n_act, deck in [
        ("Hanabi-Full", 2, 658, 20, 50),
        ("Hanabi-Full", 3, 5 * 2 * 25 + 3 + (50 - 15) + 25 + 8 + 3 + 50 + (3 + 4 + 3 + 5 + 5 + 5 + 5 + 25 + 2) + 3 * 5 * 35, 30, 50),
        ("Hanabi-Full", 5, 1280, 48, 50),
        ("Hanabi-Small", 2, 171, 11, 20),
    ]:
        env = O.OracleEnv(O.make_config(game, players), 1)
        assert (env.obs_len, env.num_actions, env.deck_size) == (obs_len, n_act, deck)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert list(O.philox([0, 0, 0, 0], [0, 0])) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_shuffle_is_a_permutation_and_depends_on_ids():
    cfg = O.make_config()
    base = np.sort(canonical_deck(cfg))
    seen = set()
    for gid in range(4):
        for ep in range(3):
            d = O.shuffled_deck(cfg, 1234, gid, ep)
            assert np.array_equal(np.sort(d), base)
            seen.add(d.tobytes())
    assert len(seen) == 12
    assert np.array_equal(O.shuffled_deck(cfg, 1234, 7, 2), O.shuffled_deck(cfg, 1234, 7, 2))


def test_fresh_game_popcount_and_sections():
    """App. A.6 popcount KAT: 5 + 40 + 8 + 3 + 250 = 306 ones, no discards/last action."""
    cfg = O.make_config()
    env = O.OracleEnv(cfg, 3, seed=99)
    out = env.observe()
    obs, legal = out["obs"], out["legal"]
    assert obs.shape == (3, 658) and set(np.unique(obs)) <= {0, 1}
    for g in range(3):
        o = obs[g]
        assert o.sum() == 306
        assert o[0:125].sum() == 5 and o[125:127].sum() == 0
        assert o[127:167].sum() == 40 and o[167:192].sum() == 0
        assert o[192:200].sum() == 8 and o[200:203].sum() == 3
        assert o[203:308].sum() == 0
        k = o[308:658].reshape(10, 35)
        assert (k[:, :25] == 1).all() and (k[:, 25:] == 0).all()
        # legal: no discards at max info, 5 plays, hints exactly for what the partner holds
        assert legal[g, 0:5].sum() == 0 and legal[g, 5:10].sum() == 5
        partner = [int(np.argmax(o[25 * i:25 * i + 25])) for i in range(5)]
        for c in range(5):
            assert legal[g, 10 + c] == int(any(card // 5 == c for card in partner))
        for r in range(5):
            assert legal[g, 15 + r] == int(any(card % 5 == r for card in partner))
    assert (out["agent_step_type"] == 0).all() and (out["agent_reward"] == 0).all()


def test_scripted_full_game_prefix():
    """App. A.7 scenario on the canonical deck: P0 = R1 R1 R1 R2 R2, P1 = R3 R3 R4 R4 R5."""
    cfg = O.make_config()
    env = O.OracleEnv(cfg, 1, decks=canonical_deck(cfg)[None])
    o = env.observe()["obs"][0]
    assert [int(np.argmax(o[25 * i:25 * i + 25])) for i in range(5)] == [2, 2, 3, 3, 4]  # P0 sees P1
    # P0 plays slot 0 (uid 5 = hand_size + 0): R1 on an empty stack succeeds
    out = env.step([5])
    o = out["obs"][0]
    assert out["reward"][0] == 1 and out["terminal"][0] == 0 and out["score"][0] == 1
    # observer is now P1: sees P0 = R1 R1 R2 R2 + new card Y1 (index 5) in slot 4
    assert [int(np.argmax(o[25 * i:25 * i + 25])) for i in range(5)] == [0, 0, 1, 1, 5]
    assert o[127:167].sum() == 39                       # deck thermometer
    assert o[167] == 1 and o[167:192].sum() == 1        # fireworks R at rank 1
    last = np.flatnonzero(o[253:308]) + 253
    # actor (rel. 1), type play, position 0, card R1, scored
    assert list(last) == [253 + 1, 255 + 0, 276 + 0, 281 + 0, 306]
    assert out["agent_step_type"][0] == 0               # P1 has not moved yet
    # P1 hints colour R to P0 (uid 10 = 2*5 + 0*5 + 0): R cards are slots 0..3
    out = env.step([10])
    o = out["obs"][0]
    assert out["reward"][0] == 0
    assert o[192:200].sum() == 7
    last = np.flatnonzero(o[253:308]) + 253
    # actor rel 1, type reveal colour, target = (1+1)%2 = 0, colour R, outcome slots 0-3
    assert list(last) == [254, 257, 259, 261, 271, 272, 273, 274]
    k = o[308:658].reshape(10, 35)
    for slot in range(4):  # own cards 0..3: only R plausible (5 bits), colour hinted R
        assert list(np.flatnonzero(k[slot, :25])) == [0, 1, 2, 3, 4]
        assert list(np.flatnonzero(k[slot, 25:])) == [0]
    assert list(np.flatnonzero(k[4, :25])) == list(range(5, 25)) and k[4, 25:].sum() == 0
    assert out["agent_step_type"][0] == 1 and out["agent_reward"][0] == 1   # P0: MID, its play scored 1
    # P0 may now discard (info 7 < 8)
    assert out["legal"][0, 0:5].sum() == 5
    # P0 plays slot 3 (an R2, uid 8): fireworks R -> 2
    out = env.step([8])
    assert out["reward"][0] == 1 and out["score"][0] == 2
    # three misplays end the game: P1 plays R5, P0 plays R1 (stack at 2), P1 plays R4
    out = env.step([5 + 4])
    assert out["reward"][0] == 0 and out["obs"][0][200:203].sum() == 2
    last = np.flatnonzero(out["obs"][0][253:308]) + 253
    assert 306 not in last and 307 not in last and (281 + 4) in last  # not scored, card R5
    assert out["obs"][0][203:253].sum() == 1                          # one discard: R5 thermometer
    assert out["obs"][0][203 + 9] == 1
    out = env.step([5 + 0])
    assert out["obs"][0][200:203].sum() == 1
    out = env.step([5 + 3])
    assert out["terminal"][0] == 1 and out["reward"][0] == -2 and out["score"][0] == 0
    assert out["agent_step_type"][0] == 2                              # seat 0 sees LAST


def test_completing_a_stack_returns_an_info_token():
    """Very-small game (1 colour): hint, then five successful plays; the fifth refunds the token."""
    cfg = O.make_config("Hanabi-Very-Small", 2)
    deck = np.array([[1, 3, 0, 2, 4, 0, 0, 1, 2, 3]], np.uint8)
    env = O.OracleEnv(cfg, 1, decks=deck)
    assert env.num_actions == 2 + 2 + 1 + 5
    out = env.step([5 + 0])  # P0 hints rank 1 (index 0) to P1 -> info 2
    assert out["obs"][0].sum() > 0 and out["reward"][0] == 0
    rewards = []
    for _ in range(5):
        out = env.step([2 + 0])  # play slot 0
        rewards.append(out["reward"][0])
    assert rewards == [1, 1, 1, 1, 1]
    assert out["terminal"][0] == 1 and out["score"][0] == 5
    st = env.export_state()[0]
    assert (st[0] >> 19) & 3 == 2            # status: fireworks completed
    assert (st[0] >> 6) & 15 == 3            # info back to max
    assert (st[2] >> 18) & 3 == 3            # last action: scored + info_token


def _unpack(cfg, row):
    P = cfg.players
    d = dict(deck_size=row[0] & 63, info=(row[0] >> 6) & 15, life=(row[0] >> 10) & 7, cur=(row[0] >> 13) & 7,
             turns=(row[0] >> 16) & 7, status=(row[0] >> 19) & 3, moves=(row[0] >> 21) & 255)
    d["fireworks"] = [(row[1] >> (3 * c)) & 7 for c in range(cfg.colors)]
    d["hand_n"] = [(row[1] >> (15 + 3 * p)) & 7 for p in range(P)]
    disc = int(row[8]) | int(row[9]) << 32
    copies = [3 if r == 0 else (1 if r == cfg.ranks - 1 else 2) for r in range(cfg.ranks)]
    d["discards"], pos = [], 0                # words 8-9: the encoder's discard section (thermometer per card identity)
    for i in range(cfg.colors * cfg.ranks):
        n = copies[i % cfg.ranks]
        d["discards"].append(bin((disc >> pos) & ((1 << n) - 1)).count("1"))
        pos += n
    d["hands"] = [[(int(row[10 + p]) >> (5 * i)) & 31 for i in range(d["hand_n"][p])] for p in range(P)]
    return d


@pytest.mark.parametrize("game,players", [("Hanabi-Full", 2), ("Hanabi-Full", 5), ("Hanabi-Small", 3),
                                          ("Hanabi-Very-Small", 2), ("Hanabi-Full", 4)])
def test_random_play_invariants(game, players):
    """App. A.7: card conservation, token bounds, end-game turn count, encoder structure."""
    cfg = O.make_config(game, players)
    n = 64
    env = O.OracleEnv(cfg, n, seed=5)
    D, bits = env.deck_size, cfg.colors * cfg.ranks
    copies = np.bincount(canonical_deck(cfg), minlength=bits)
    out = env.observe()
    done = np.zeros(n, bool)
    moves_after_empty = np.full(n, -1)
    for t in range(120):
        assert (out["legal"].sum(1)[~done] > 0).all()
        act = O.random_legal_actions(out["legal"], seed=4321, draw=t)
        prev_done = done.copy()
        out = env.step(act)
        rows = env.export_state()
        for g in range(n):
            if prev_done[g]:
                continue
            s = _unpack(cfg, rows[g])
            counts = np.zeros(bits, int)
            for h in s["hands"]:
                for card in h:
                    counts[card] += 1
            counts += np.array(s["discards"])
            for c, f in enumerate(s["fireworks"]):
                counts[c * cfg.ranks:c * cfg.ranks + f] += 1
            deck = np.frombuffer(rows[g, 10 + 3 * players:].tobytes(), np.uint8)[D - s["deck_size"]:D]
            counts += np.bincount(deck, minlength=bits)
            assert np.array_equal(counts, copies)
            assert 0 <= s["info"] <= cfg.max_info and 0 <= s["life"] <= cfg.max_life
            if s["deck_size"] == 0:
                moves_after_empty[g] += 1
            if out["terminal"][g]:
                done[g] = True
                if s["status"] == 3:
                    assert moves_after_empty[g] == players  # exactly P moves after the last draw
        o = out["obs"]
        assert set(np.unique(o)) <= {0, 1}
        if done.all():
            break
    assert done.all()
    assert env.illegal_count() == 0


def test_illegal_move_is_ignored_and_counted():
    cfg = O.make_config()
    env = O.OracleEnv(cfg, 2, seed=1)
    before = env.export_state()
    first = env.observe()
    out = env.step([0, 99])  # discard at 8 info tokens; uid out of range
    assert env.illegal_count() == 2
    assert np.array_equal(env.export_state(), before)
    assert np.array_equal(out["obs"], first["obs"]) and np.array_equal(out["legal"], first["legal"])


def test_auto_reset_and_seat_bookkeeping():
    """Lock-step 2-seat self-play with auto-reset: per-seat rewards/step types match a replay in Python."""
    cfg = O.make_config("Hanabi-Small", 2, flags=O.FLAG_AUTO_RESET | O.FLAG_RESET_START_NEXT)
    n = 32
    env = O.OracleEnv(cfg, n, seed=3)
    out = env.observe()
    pending = np.zeros((2, n), bool)
    term_since = np.zeros((2, n), bool)
    acc = np.zeros((2, n))
    episodes = 0
    for t in range(300):
        seat = t % 2
        cur = (env.export_state()[:, 0] >> 13) & 7
        assert (cur == seat).all()  # RESET_START_NEXT keeps every game on the same acting seat
        exp_type = np.where(~pending[seat], 0, np.where(term_since[seat], 2, 1))
        assert np.array_equal(out["agent_step_type"], exp_type)
        assert np.array_equal(out["agent_reward"], np.where(pending[seat], acc[seat], 0))
        act = O.random_legal_actions(out["legal"], seed=8, draw=t)
        out = env.step(act)
        pending[seat] = True
        term_since[seat] = False
        acc[seat] = 0
        live = pending & ~term_since
        acc += live * out["reward"][None]
        term_since |= live & (out["terminal"][None] == 1)
        episodes += int(out["terminal"].sum())
        # a fresh game after auto-reset: full deck minus hands, no last action
        fresh = out["terminal"] == 1
        if fresh.any():
            st = env.export_state()[fresh]
            assert ((st[:, 0] & 63) == env.deck_size - 4).all() and (st[:, 2] == 0).all()
    assert episodes > 50
