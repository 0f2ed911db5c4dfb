"""Rule-based partners: hanabi_agents/rule_based of the reference on the GPU.

The reference's rules run on `hanabi_learning_environment` observation objects, which are absent here, and the
reference has no tests for them: parity against the running reference is UNPINNED. What pins the behaviour:
  * hand-worked situations below (CPU oracle, explicit decks), derived from ruleset.py by reading;
  * bit-exact agreement of the HIP kernel with the independent C restatement (oracle/rule_oracle.c) over long
    self-play of all four predefined rule lists, 2-5 players, three game sizes;
  * self-play scores of the four agents in the range the literature reports for them.
"""
import numpy as np
import pytest

from oracle import oracle_py as O

FULL = [3, 2, 2, 2, 1]


def deck_with_prefix(prefix, colors=5, ranks=5):
    """A legal deck starting with `prefix` (card = colour * ranks + rank), the rest in canonical order."""
    left = {c * ranks + r: FULL[r] for c in range(colors) for r in range(ranks)}
    for card in prefix:
        left[card] -= 1
        assert left[card] >= 0
    return list(prefix) + [card for card in sorted(left) for _ in range(left[card])]


def rules_of(agent_rules):
    return [(r.kind, r.arg, r.threshold) for r in agent_rules]


R1, R2, Y1, Y2, G3, G5, W4, W5, B1, B5 = 0, 1, 5, 6, 12, 14, 18, 19, 20, 24


def _env(prefix):
    cfg = O.make_config("Hanabi-Full", 2)
    return O.OracleEnv(cfg, 1, seed=1, decks=np.array([deck_with_prefix(prefix)], np.uint8))


def test_opening_moves_of_the_four_agents():
    """P0 holds [R1 R1 Y2 G3 W4] and knows nothing; P1 holds [B1 R2 Y1 G5 B5]; 8 information tokens."""
    from hanabi_agents.rule_based import Ruleset, predefined_rules as PR

    env = _env([R1, R1, Y2, G3, W4, B1, R2, Y1, G5, B5, W5])
    one = lambda rule: env.rule_act(rules_of([rule]), 7, 1)
    none = lambda rule: one(rule)[1][0] == 1                      # fired == n_rules: the rule returned None
    # nothing is known and the tokens are full: no safe play, no discard (ruleset.py:207,221), nothing dispensable
    for rule in (Ruleset.play_safe_card, Ruleset.play_if_certain, Ruleset.osawa_discard, Ruleset.discard_oldest_first,
                 Ruleset.discard_randomly, Ruleset.hail_mary, Ruleset.tell_dispensable_factory(8),
                 Ruleset.tell_most_information, Ruleset.play_probably_safe_factory(0.6, True)):
        assert none(rule), rule
    # B1 is the first playable card of the partner: rank before colour (ruleset.py:424-432) -> reveal rank 1 = uid 15
    assert one(Ruleset.tell_playable_card_outer)[0][0] == 15 and one(Ruleset.tell_anyone_useful_card)[0][0] == 15
    # tell_unknown: first card without a colour hint -> its colour, blue = uid 10 + 4
    assert one(Ruleset.tell_unknown)[0][0] == 14
    # playability of every own slot: 15 rank-1 cards minus the visible B1, Y1 = 13 of the 45 unseen cards = 0.2889
    assert one(Ruleset.play_probably_safe_factory(0.25))[0][0] == 5           # play slot 0 (first maximum)
    assert none(Ruleset.play_probably_safe_factory(0.29))
    for rules, uid, which in ((PR.flawed_rules, 5, 1), (PR.iggi_rules, 15, 2), (PR.outer_rules, 15, 2), (PR.piers_rules, 15, 3)):
        act, fired = env.rule_act(rules_of(rules), 7, 1)
        assert (act[0], fired[0]) == (uid, which)


def test_follow_up_situations():
    from hanabi_agents.rule_based import Ruleset

    env = _env([R1, R1, Y2, G3, W4, B1, R2, Y1, G5, B5, W5])
    one = lambda rule: env.rule_act(rules_of([rule]), 7, 1)
    env.step([15])                                                 # P0 reveals rank 1: touches B1 (slot 0) and Y1 (slot 2)
    # P1: slot 0 is a 1 of unknown colour and every pile is empty: all plausible identities playable (ruleset.py:362-369)
    assert one(Ruleset.play_safe_card)[0][0] == 5 and one(Ruleset.play_if_certain)[1][0] == 1
    assert one(Ruleset.osawa_discard)[1][0] == 1                   # nothing known to be dead, 7 tokens
    assert one(Ruleset.discard_oldest_first)[0][0] == 0
    env.step([5])                                                  # B1 played: blue pile 1, P1 = [R2 Y1 G5 B5 W5]
    # P0: Y1 is playable and already rank-hinted -> reveal its colour (yellow = uid 11) (ruleset.py:434-441)
    assert one(Ruleset.tell_playable_card_outer)[0][0] == 11
    assert one(Ruleset.tell_dispensable_factory(8))[1][0] == 1     # no card below its pile, no complete pile
    env.step([11])                                                 # yellow: touches Y1 only -> fully known
    assert one(Ruleset.play_if_certain)[0][0] == 5 + 1             # P1 is certain about slot 1
    env.step([6])                                                  # Y1 played; P1 = [R2 G5 B5 W5 + next card]
    # P0 has 6 tokens, knows nothing about its own cards; hand over the turn by discarding, then look at P1's view
    env.step([0])                                                  # P0 discards R1 -> 7 tokens
    # a 1 below every pile? piles are R0 Y1 G0 W0 B1 -> min 0: nothing dispensable by rank
    assert one(Ruleset.osawa_discard)[1][0] == 1


def test_osawa_and_dispensable_on_dead_cards():
    """Blue pile at 1 and a second B1 in the partner's hand: it is dispensable; once its holder knows colour and
    rank, osawa_discard throws it away (ruleset.py:238-247, 489-507)."""
    from hanabi_agents.rule_based import Ruleset

    env = _env([R1, R2, Y2, G3, W4, B1, B1, Y1, G5, B5, W5])
    one = lambda rule: env.rule_act(rules_of([rule]), 3, 9)
    env.step([15])                                                 # P0: rank 1 to P1 (B1, B1, Y1)
    env.step([5])                                                  # P1 plays B1 -> blue 1; P1 = [B1 Y1 G5 B5 W5]
    # P0, 7 tokens: P1's B1 (slot 0) has rank 0 < pile 1, rank hinted, colour not -> reveal colour blue (uid 14)
    assert one(Ruleset.tell_dispensable_factory(8))[0][0] == 14
    assert one(Ruleset.tell_dispensable_factory(3))[1][0] == 1     # Piers only does this below 3 tokens
    env.step([14])                                                 # blue touches B1 and B5
    # P1: slot 0 known B1 with blue pile at 1: colour+rank hinted and rank < pile -> discard slot 0
    assert one(Ruleset.osawa_discard)[0][0] == 0
    # slot 3 (B5): blue known, rank in {2..5}: not dead. slot 1 (Y1): rank 1, colours {R,Y,G,W}: all playable
    assert one(Ruleset.play_safe_card)[0][0] == 5 + 1


def test_hail_mary_and_probabilities_small_game():
    """Very small game (1 colour): deck runs out quickly; with an empty deck and a spare life Piers plays its
    likeliest card (ruleset.py:653-655) — there is only one life in this game, so the rule must stay silent."""
    from hanabi_agents.rule_based import Ruleset

    cfg = O.make_config("Hanabi-Very-Small", 2)
    env = O.OracleEnv(cfg, 16, seed=5)
    hm = rules_of([Ruleset.hail_mary])
    for t in range(12):
        act, fired = env.rule_act(hm, 1, t)
        assert (fired == 1).all()                                  # max_life = 1: never "more than one life"
        env.step(act)


@pytest.mark.gpu
@pytest.mark.parametrize("game,players", [("Hanabi-Full", 2), ("Hanabi-Full", 3), ("Hanabi-Full", 5), ("Hanabi-Small", 2),
                                          ("Hanabi-Small", 4), ("Hanabi-Very-Small", 2)])
def test_hip_rule_agents_match_oracle(game, players):
    """All rule kinds, walked by the HIP kernel and by rule_oracle.c on identical games: same move and same firing
    rule for every game at every step, through several episodes (auto-reset)."""
    import hanabi_hip
    from hanabi_agents.rule_based import RulebasedAgent, Ruleset, predefined_rules as PR

    n, steps = 192, 260
    flags = hanabi_hip.FLAG_AUTO_RESET
    env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config(game, players, flags), n_games=n, seed=21)
    orc = O.OracleEnv(O.make_config(game, players, flags), n, seed=21)
    everything = [Ruleset.hail_mary, Ruleset.play_if_certain, Ruleset.play_safe_card, Ruleset.play_probably_safe_factory(0.8, True),
                  Ruleset.discard_probably_useless_factory(0.9), Ruleset.tell_anyone_useless_card, Ruleset.tell_dispensable_factory(5),
                  Ruleset.tell_playable_card, Ruleset.tell_most_information, Ruleset.tell_unknown, Ruleset.osawa_discard,
                  Ruleset.discard_randomly]
    lists = [PR.flawed_rules, PR.iggi_rules, PR.outer_rules, PR.piers_rules, everything, [Ruleset.tell_randomly], []]
    agents = [RulebasedAgent(r, seed=100 + i) for i, r in enumerate(lists)]
    fired_any = np.zeros(16, bool)
    for t in range(steps):
        k = (t // players) % len(agents)                            # one rule list per round of the table
        ag = agents[k]
        act = ag.explore((env, (env.obs, env.legal)))
        want_act, want_fired = orc.rule_act(rules_of(ag.rules), ag.seed, ag._draws)
        got_act, got_fired = act.cpu().numpy(), ag._fired.cpu().numpy()
        assert np.array_equal(got_fired, want_fired), f"step {t}: rule index differs"
        assert np.array_equal(got_act, want_act), f"step {t}: move differs"
        legal = env.legal.cpu().numpy()
        assert legal[np.arange(n), got_act].all(), "a rule produced an illegal move"
        for q in np.unique(got_fired):
            if q < len(ag.rules):
                fired_any[ag.rules[q].kind] = True
        env.step(act)
        orc.step(got_act)
    assert env.illegal_count() == 0
    assert np.array_equal(env.export_state().cpu().numpy().astype(np.uint32), orc.export_state())
    assert sum(a.totalCalls for a in agents) == n * steps
    if game == "Hanabi-Full" and players == 2:
        # every kind except tell_most_information (never fires) and legal_random-as-a-rule was exercised
        assert fired_any[[1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14]].all(), fired_any


@pytest.mark.gpu
def test_self_play_scores_are_plausible():
    """2-player Hanabi-Full self-play. Published self-play means (Walton-Rivers et al. 2017 / Canaan et al. 2019, the
    agents these lists re-implement): IGGI ~16.9, Outer ~14.5, Piers ~17.3, Flawed ~2.6 (Flawed bombs out often).
    The reference's ports differ in details, so only the ordering and coarse ranges are asserted."""
    import hanabi_hip
    from hanabi_agents.rule_based import RulebasedAgent, predefined_rules as PR
    from hanabi_hip.selfplay import SelfPlaySession

    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT
    score = {}
    for name, rules in (("flawed", PR.flawed_rules), ("iggi", PR.iggi_rules), ("outer", PR.outer_rules), ("piers", PR.piers_rules)):
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Full", 2, flags), n_games=2048, seed=2)
        agent = RulebasedAgent(rules)
        sess = SelfPlaySession(env, [agent, agent])
        sess.run(400, train=False)
        assert sess.episodes > 2048 and env.illegal_count() == 0
        score[name] = sess.mean_score()
        hist = agent.histogram
        assert sum(hist) == 2048 * 400 and hist[-1] == 0 if name != "outer" else True
    print("self-play means:", score)
    assert score["flawed"] < 8 < score["outer"] and score["iggi"] > 12 and score["piers"] > 12


@pytest.mark.gpu
def test_dqn_agent_trains_beside_a_rule_based_partner(tmp_path):
    """Seat 0 learns, seat 1 is Piers: the session feeds each what it needs (vectorised obs vs state rows), only the
    learner's replay fills, and the mixed session checkpoints / resumes."""
    import torch

    import hanabi_hip
    from hanabi_agents.rlax_dqn import ActionSpec, DQNAgent, ObservationSpec, RlaxRainbowParams
    from hanabi_agents.rule_based import RulebasedAgent, predefined_rules as PR
    from hanabi_hip.selfplay import SelfPlaySession

    n = 128
    flags = hanabi_hip.FLAG_AUTO_RESET | hanabi_hip.FLAG_RESET_START_NEXT

    def make():
        env = hanabi_hip.HanabiEnv(config=hanabi_hip.make_config("Hanabi-Small", 2, flags), n_games=n, seed=4)
        params = RlaxRainbowParams(train_batch_size=32, experience_buffer_size=n * 16, layers=[32], mask_terminal=True)
        learner = DQNAgent(ObservationSpec((n, env.obs_len)), ActionSpec(env.num_actions), params, device="cuda")
        partner = RulebasedAgent(PR.piers_rules)
        return env, learner, partner, SelfPlaySession(env, [learner, partner], train_seats=[0])

    env, learner, partner, sess = make()
    sess.run(30)
    assert env.illegal_count() == 0 and learner.experience.size == n * 14 and sess.grad_steps > 0
    assert partner.totalCalls == n * 15 and not partner.requires_vectorized_observation()
    sess.save_checkpoint(tmp_path / "mixed.ckpt")
    sess.run(10)
    want = (env.export_state().cpu().numpy(), learner.online.layers[0].w.detach().cpu().numpy(), partner.histogram)
    env2, learner2, partner2, sess2 = make()
    sess2.load_checkpoint(tmp_path / "mixed.ckpt")
    sess2.run(10)
    assert np.array_equal(env2.export_state().cpu().numpy(), want[0])
    assert np.array_equal(learner2.online.layers[0].w.detach().cpu().numpy(), want[1]) and partner2.histogram == want[2]
    assert torch.isfinite(learner2.last_loss).item()
