"""Hand-worked 3-, 4- and 5-player scenarios on explicit decks: what the 2-player KATs of
test_oracle_env.py cannot see — observer-relative actor / target bits of the last-action section for P >= 3, reveal
bitmasks with hand size 4, knowledge shifting after a play, missing-card flags once the deck has run out, and "exactly P
more moves after the last draw". Every expected bit position below is derived by hand from the section formulae of A.6,
not read off the implementation. Each scenario runs on the CPU oracle (always) and on the HIP kernel (`-m gpu`, int8 and
bit-packed output), which must also agree with the oracle bit for bit on everything it emits.

Upstream HLE is absent from the reference: these pin the oracle to the written spec ("parity unpinned")."""
import numpy as np
import pytest

from oracle import oracle_py as O


def canonical_deck(cfg):
    deck = []
    for c in range(cfg.colors):
        for r in range(cfg.ranks):
            deck += [c * cfg.ranks + r] * (3 if r == 0 else (1 if r == cfg.ranks - 1 else 2))
    return np.array(deck, np.uint8)


class OracleEngine:
    def __init__(self, game, players, decks):
        self.cfg = O.make_config(game, players)
        self.env = O.OracleEnv(self.cfg, decks.shape[0], decks=decks)

    def observe(self):
        return self.env.observe()

    def step(self, act):   # the same move in every game (all games hold the same deck)
        return self.env.step(np.ascontiguousarray(np.broadcast_to(np.asarray(act, np.int32), (self.env.n,))))

    def state(self):
        return self.env.export_state()


class HipEngine:
    """The HIP kernel through the C-ABI, checked against the oracle on every call."""

    def __init__(self, game, players, decks, packed):
        import hanabi_hip

        self.ref = OracleEngine(game, players, decks)
        self.cfg = self.ref.cfg
        self.env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players), n_games=decks.shape[0], decks=decks, packed=packed)

    def _out(self, want, stepped):
        e = self.env
        got = dict(obs=e.obs.cpu().numpy(), legal=e.legal.cpu().numpy(), agent_reward=e.agent_reward.cpu().numpy(),
                   agent_step_type=e.agent_step_type.cpu().numpy())
        if stepped:
            got.update(reward=e.reward.cpu().numpy(), terminal=e.terminal.cpu().numpy(), score=e.score.cpu().numpy())
        for k, v in got.items():
            assert np.array_equal(v, want[k]), f"HIP != oracle in {k}"
        assert np.array_equal(self.state(), self.ref.state())
        return got

    def observe(self):
        self.env.observe()
        return self._out(self.ref.observe(), False)

    def step(self, act):
        import torch

        a = np.ascontiguousarray(np.broadcast_to(np.asarray(act, np.int32), (self.env.n,)))
        self.env.step(torch.as_tensor(a).cuda())
        return self._out(self.ref.step(a), True)

    def state(self):
        return self.env.export_state().cpu().numpy().view(np.uint32)


def ones(o, lo, hi):
    return [int(i) for i in np.flatnonzero(o[lo:hi]) + lo]


def cards_in(o, lo, n, bits=25):
    return [int(np.argmax(o[lo + bits * i:lo + bits * (i + 1)])) if o[lo + bits * i:lo + bits * (i + 1)].any() else None for i in range(n)]


# ---- 3 players, full game: obs 956 = hands 250 | flags 3 | deck 35 | fireworks 25 | info 8 | life 3 | discards 50 |
#      last action 57 (actor 3, type 4, target 3, colour 5, rank 5, outcome 5, position 5, card 25, scored/info 2) | knowledge 525
H3, FLAGS3, DECK3, FW3, INFO3, LIFE3, DISC3, LAST3, KN3, END3 = 0, 250, 253, 288, 313, 321, 324, 374, 431, 956


def scenario_three_players(eng):
    """Canonical deck: P0 = R1 R1 R1 R2 R2, P1 = R3 R3 R4 R4 R5, P2 = Y1 Y1 Y1 Y2 Y2 (dealt player by player, A.3)."""
    out = eng.observe()
    o = out["obs"][0]
    assert len(o) == END3 and out["legal"].shape[1] == 30
    assert cards_in(o, 0, 5) == [2, 2, 3, 3, 4] and cards_in(o, 125, 5) == [5, 5, 5, 6, 6]   # P0 sees P1, then P2
    assert o[DECK3:FW3].sum() == 35 and not o[LAST3:KN3].any()
    # 1. P0 reveals rank 1 to the player at offset 2 (P2): uid = 2*5 + 2*5 + (2-1)*5 + 0 = 25; touches slots 0, 1, 2
    out = eng.step([25])
    o = out["obs"][0]                                             # observer: P1
    # actor P0 is at offset (0 - 1) mod 3 = 2; target = (2 + 2) mod 3 = 1 (P2 sits right after P1); rank index 0; outcome 0-2
    assert ones(o, LAST3, KN3) == [374 + 2, 377 + 3, 381 + 1, 389 + 0, 394, 395, 396]
    assert o[INFO3:LIFE3].sum() == 7 and out["reward"][0] == 0
    kn = o[KN3:END3].reshape(15, 35)                               # slots: P1's own 0-4, P2's 5-9, P0's 10-14
    for i in (5, 6, 7):                                            # P2's hinted cards: rank 1 of any colour, rank hinted
        assert list(np.flatnonzero(kn[i, :25])) == [0, 5, 10, 15, 20] and list(np.flatnonzero(kn[i, 25:])) == [5 + 0]
    for i in (8, 9):                                               # its other cards: anything but rank 1, nothing hinted
        assert list(np.flatnonzero(kn[i, :25])) == [c * 5 + r for c in range(5) for r in range(1, 5)] and not kn[i, 25:].any()
    assert (kn[:5, :25] == 1).all() and (kn[10:, :25] == 1).all() and not kn[:5, 25:].any() and not kn[10:, 25:].any()
    assert out["agent_step_type"][0] == 0                          # P1 has no move pending: FIRST
    # 2. P1 discards slot 0 (R3; legal now that a token is spent), draws deck[15] = Y3
    assert out["legal"][0, 0:5].sum() == 5
    out = eng.step([0])
    o = out["obs"][0]                                             # observer: P2
    assert ones(o, LAST3, KN3) == [374 + 2, 377 + 1, 399 + 0, 404 + 2]   # actor P1 at offset 2, discard, position 0, card R3
    assert o[INFO3:LIFE3].sum() == 8 and o[DECK3:FW3].sum() == 34
    assert ones(o, DISC3, LAST3) == [DISC3 + 3 + 2]                # R3's thermometer starts after R1 (3 bits) and R2 (2 bits)
    assert cards_in(o, 0, 5) == [0, 0, 0, 1, 1] and cards_in(o, 125, 5) == [2, 3, 3, 4, 7]   # P2 sees P0, then P1 (new card last)
    kn = o[KN3:END3].reshape(15, 35)                               # now P2's own knowledge comes first
    assert all(list(np.flatnonzero(kn[i, 25:])) == [5] for i in (0, 1, 2)) and not kn[3:5, 25:].any()
    assert (kn[10 + 4, :25] == 1).all()                            # P1's fresh card: everything plausible
    assert out["legal"][0, 0:5].sum() == 0                         # 8 tokens again: no discards
    # 3. P2 plays slot 0 (Y1): success, draws deck[16] = Y3; its remaining cards shift left
    out = eng.step([5])
    o = out["obs"][0]                                             # observer: P0
    assert out["reward"][0] == 1 and out["score"][0] == 1 and out["terminal"][0] == 0
    assert ones(o, LAST3, KN3) == [374 + 2, 377 + 0, 399 + 0, 404 + 5, 429]    # actor P2 at offset 2, play, pos 0, card Y1, scored
    assert ones(o, FW3, INFO3) == [FW3 + 1 * 5 + 0]                # yellow stack at rank 1
    assert cards_in(o, 125, 5) == [5, 5, 6, 6, 7]                  # P2 (offset 2): Y1 Y1 Y2 Y2 Y3
    kn = o[KN3:END3].reshape(15, 35)                               # P0's own 0-4, P1's 5-9, P2's 10-14
    assert all(list(np.flatnonzero(kn[10 + i, 25:])) == [5] for i in (0, 1))          # the two hinted Y1 moved to slots 0, 1
    assert not kn[10 + 2:10 + 5, 25:].any() and (kn[10 + 4, :25] == 1).all() and kn[10 + 2, 0] == 0
    assert out["agent_step_type"][0] == 1 and out["agent_reward"][0] == 1        # P0: MID; P2's play counts for it
    assert not o[FLAGS3:DECK3].any()                               # nobody is short of cards


# ---- 5 players, full game: obs 1280 = hands 400 | flags 5 | deck 30 | fireworks 25 | info 8 | life 3 | discards 50 |
#      last action 59 (actor 5, type 4, target 5, colour 5, rank 5, outcome 4, position 4, card 25, scored/info 2) | knowledge 700
FLAGS5, DECK5, FW5, INFO5, LIFE5, DISC5, LAST5, KN5, END5 = 400, 405, 435, 460, 468, 471, 521, 580, 1280


def scenario_five_players(eng):
    """Canonical deck, hands of 4: P0 = R1 R1 R1 R2, P1 = R2 R3 R3 R4, P2 = R4 R5 Y1 Y1, P3 = Y1 Y2 Y2 Y3, P4 = Y3 Y4 Y4 Y5."""
    P = 5
    out = eng.observe()
    o = out["obs"][0]
    assert len(o) == END5 and out["legal"].shape[1] == 48
    assert [cards_in(o, 100 * k, 4) for k in range(4)] == [[1, 2, 2, 3], [3, 4, 5, 5], [5, 6, 6, 7], [7, 8, 8, 9]]
    assert o[DECK5:FW5].sum() == 30
    # P0 reveals colour Y to the player at offset 3 (P3): uid = 2*4 + (3-1)*5 + 1 = 19; all FOUR of its cards are yellow
    out = eng.step([19])
    o = out["obs"][0]                                             # observer: P1
    # actor P0 at offset (0 - 1) mod 5 = 4; target = (4 + 3) mod 5 = 2 (P3 from P1); colour index 1; outcome bits 0-3
    assert ones(o, LAST5, KN5) == [521 + 4, 526 + 2, 530 + 2, 535 + 1, 545, 546, 547, 548]
    kn = o[KN5:END5].reshape(20, 35)                               # P1's own 0-3, P2's 4-7, P3's 8-11, ...
    for i in range(8, 12):
        assert list(np.flatnonzero(kn[i, :25])) == [5, 6, 7, 8, 9] and list(np.flatnonzero(kn[i, 25:])) == [1]
    # Run the deck down: odd moves discard slot 0 (legal: a token was just spent), even moves hint the rank of the next
    # player's oldest card. Nothing is ever played, so the score stays 0 and the game must end by running out of turns.
    moves, after_empty, deck_left = 1, None, 30
    while True:
        st = eng.state()[0]
        assert (st[0] & 63) == deck_left
        cur = (st[0] >> 13) & 7
        assert cur == moves % P
        if moves % 2 == 1:
            uid = 0
        else:
            nxt = (cur + 1) % P
            uid = 2 * 4 + 4 * 5 + 0 * 5 + ((int(st[10 + nxt]) & 31) % 5)
        assert out["legal"][0, uid] == 1
        out = eng.step([uid])
        o = out["obs"][0]
        moves += 1
        if uid == 0 and deck_left > 0:
            deck_left -= 1
        assert o[DECK5:FW5].sum() == deck_left and o[LIFE5:DISC5].sum() == 3 and out["reward"][0] == 0
        if after_empty is not None:
            after_empty += 1
            if uid == 0:
                # the discarder could not draw: it is one card short. The next observer sees it at offset P - 1.
                assert o[FLAGS5 + P - 1] == 1 and cards_in(o, 300, 4)[3] is None
                assert not o[KN5 + (4 * (P - 1) + 3) * 35:KN5 + (4 * (P - 1) + 4) * 35].any()   # its 4th knowledge slot is empty
        elif deck_left == 0:
            after_empty = 0
            assert not o[FLAGS5:DECK5].any()                       # the last card was just drawn: every hand is still full
        if out["terminal"][0]:
            break
        assert moves < 200
    assert after_empty == P                                        # exactly P moves after the draw that emptied the deck
    st = eng.state()[0]
    assert (st[0] >> 19) & 3 == 3 and out["score"][0] == 0 and out["agent_step_type"][0] == 2
    # seats that discarded during the last round (moves are discard, hint, discard, ...): their flags, observer-relative
    short = [(s - (moves % P)) % P for s in range(P) if ((st[1] >> (15 + 3 * s)) & 7) < 4]
    assert ones(o, FLAGS5, DECK5) == sorted(FLAGS5 + r for r in short) and 1 <= len(short) <= 3


# ---- 4 players, full game: obs 1041 = hands 300 | flags 4 | deck 34 | fireworks 25 | info 8 | life 3 | discards 50 |
#      last action 57 (actor 4, type 4, target 4, colour 5, rank 5, outcome 4, position 4, card 25, scored/info 2) | knowledge 560
FLAGS4, DECK4, FW4, INFO4, LIFE4, DISC4, LAST4, KN4, END4 = 300, 304, 338, 363, 371, 374, 424, 481, 1041


def scenario_four_players(eng):
    """Canonical deck, hands of 4: P0 = R1 R1 R1 R2, P1 = R2 R3 R3 R4, P2 = R4 R5 Y1 Y1, P3 = Y1 Y2 Y2 Y3; deck goes on Y3 Y4 Y4 Y5.
    One successful play, then three misplays: the game ends on the third lost life with every point forfeited (A.5, A.7)."""
    out = eng.observe()
    o = out["obs"][0]
    assert len(o) == END4 and out["legal"].shape[1] == 38          # 4 discards + 4 plays + 3 x 5 colour + 3 x 5 rank hints
    assert [cards_in(o, 100 * k, 4) for k in range(3)] == [[1, 2, 2, 3], [3, 4, 5, 5], [5, 6, 6, 7]]
    assert o[DECK4:FW4].sum() == 34 and o[LIFE4:DISC4].sum() == 3 and not o[LAST4:KN4].any()
    # 1. P0 plays slot 0 (R1): success, draws Y3
    out = eng.step([4 + 0])
    o = out["obs"][0]                                             # observer: P1; P0 sits at offset (0 - 1) mod 4 = 3
    assert out["reward"][0] == 1 and out["score"][0] == 1 and out["terminal"][0] == 0
    assert ones(o, LAST4, KN4) == [LAST4 + 3, 428 + 0, 450 + 0, 454 + 0, 479]      # actor, play, position 0, card R1, scored
    assert ones(o, FW4, INFO4) == [FW4 + 0] and o[DECK4:FW4].sum() == 33
    assert cards_in(o, 200, 4) == [0, 0, 1, 7]                      # P0 (offset 3): R1 R1 R2 + the fresh Y3 in the last slot
    # 2. P1 plays slot 1 (R3) on a red stack at 1: misplay, a life is lost, R3 goes to the discards, draws Y4
    out = eng.step([4 + 1])
    o = out["obs"][0]                                             # observer: P2
    assert out["reward"][0] == 0 and out["score"][0] == 1 and out["terminal"][0] == 0
    assert ones(o, LAST4, KN4) == [LAST4 + 3, 428 + 0, 450 + 1, 454 + 2]           # no scored / info bit
    assert o[LIFE4:DISC4].sum() == 2 and ones(o, DISC4, LAST4) == [DISC4 + 5]      # R3's thermometer follows R1 (3) and R2 (2)
    assert ones(o, FW4, INFO4) == [FW4 + 0] and o[INFO4:LIFE4].sum() == 8
    assert cards_in(o, 200, 4) == [1, 2, 3, 8]                      # P1 (offset 3): R2 R3 R4 Y4
    # 3. P2 plays slot 1 (R5): second misplay
    out = eng.step([4 + 1])
    o = out["obs"][0]                                             # observer: P3
    assert ones(o, LAST4, KN4) == [LAST4 + 3, 428 + 0, 450 + 1, 454 + 4]
    assert o[LIFE4:DISC4].sum() == 1 and ones(o, DISC4, LAST4) == [DISC4 + 5, DISC4 + 9]
    assert out["terminal"][0] == 0 and out["agent_step_type"][0] == 0   # P3 has not moved yet: FIRST
    # 4. P3 plays slot 3 (Y3) on an empty yellow stack: third misplay, the last life: game over, the point is forfeited
    out = eng.step([4 + 3])
    o = out["obs"][0]                                             # observer: P0
    assert out["terminal"][0] == 1 and out["score"][0] == 0 and out["reward"][0] == -1
    assert o[LIFE4:DISC4].sum() == 0 and ones(o, DISC4, LAST4) == [DISC4 + 5, DISC4 + 9, DISC4 + 10 + 5]
    assert ones(o, LAST4, KN4) == [LAST4 + 3, 428 + 0, 450 + 3, 454 + 7]
    assert out["agent_step_type"][0] == 2                          # LAST for the seat that would move next
    assert out["agent_reward"][0] == 0                             # P0: +1 for its own play ... -1 at the end, summed since its move
    st = eng.state()[0]
    assert ((st[0] >> 10) & 7) == 0 and ((st[0] >> 19) & 3) != 0  # no lives left, game marked over


def _decks(cfg_game, players, n=3):
    return np.tile(canonical_deck(O.make_config(cfg_game, players)), (n, 1))


def test_three_player_scenario_on_the_oracle():
    scenario_three_players(OracleEngine("Hanabi-Full", 3, _decks("Hanabi-Full", 3)))


def test_four_player_scenario_on_the_oracle():
    scenario_four_players(OracleEngine("Hanabi-Full", 4, _decks("Hanabi-Full", 4)))


def test_five_player_scenario_on_the_oracle():
    scenario_five_players(OracleEngine("Hanabi-Full", 5, _decks("Hanabi-Full", 5)))


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True])
def test_three_player_scenario_on_the_hip_kernel(packed):
    scenario_three_players(HipEngine("Hanabi-Full", 3, _decks("Hanabi-Full", 3), packed))


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True])
def test_five_player_scenario_on_the_hip_kernel(packed):
    scenario_five_players(HipEngine("Hanabi-Full", 5, _decks("Hanabi-Full", 5), packed))


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True])
def test_four_player_scenario_on_the_hip_kernel(packed):
    scenario_four_players(HipEngine("Hanabi-Full", 4, _decks("Hanabi-Full", 4), packed))
