"""Ruleset — the rule vocabulary of the reference (hanabi_agents/rule_based/ruleset.py:202-655) as descriptors.

In the reference every rule is a Python function `rule(observation) -> HanabiMove | None` over the rich
observation object of ONE game. Here a rule is a small record (kind, arg, threshold) that `hb_rule_act`
(include/hanabi_hip.h) evaluates for all games in one launch; the names, the factory signatures and their defaults
are the reference's, so rule lists read the same (`predefined_rules.py`). Semantics per rule: the table in
include/hanabi_hip.h and csrc/rule_agent.hip; what was read how: oracle/rule_oracle.c header.
"""
from typing import NamedTuple

from hanabi_hip import _capi as K


class Rule(NamedTuple):
    kind: int
    arg: int = 0
    threshold: float = 0.0
    name: str = ""

    def __repr__(self):
        return f"<Rule {self.name}>"

    def __call__(self, observation):  # the reference calls rule(observation) per game; there is no per-game path here
        raise TypeError("rules are evaluated on the GPU for all games at once: pass them to RulebasedAgent")


class Ruleset:
    legal_random = Rule(K.RULE_LEGAL_RANDOM, name="legal_random")                          # ruleset.py:598
    discard_oldest_first = Rule(K.RULE_DISCARD_OLDEST_FIRST, name="discard_oldest_first")  # :206
    osawa_discard = Rule(K.RULE_OSAWA_DISCARD, name="osawa_discard")                       # :220
    tell_unknown = Rule(K.RULE_TELL_UNKNOWN, name="tell_unknown")                          # :285
    tell_randomly = Rule(K.RULE_TELL_RANDOMLY, name="tell_randomly")                       # :314
    play_safe_card = Rule(K.RULE_PLAY_SAFE_CARD, name="play_safe_card")                    # :350
    play_if_certain = Rule(K.RULE_PLAY_IF_CERTAIN, name="play_if_certain")                 # :383
    tell_playable_card_outer = Rule(K.RULE_TELL_PLAYABLE_CARD_OUTER, name="tell_playable_card_outer")  # :413
    tell_anyone_useful_card = Rule(K.RULE_TELL_PLAYABLE_CARD_OUTER, name="tell_anyone_useful_card")    # :518 (same rule)
    tell_anyone_useless_card = Rule(K.RULE_TELL_ANYONE_USELESS_CARD, name="tell_anyone_useless_card")  # :522
    tell_most_information = Rule(K.RULE_TELL_MOST_INFORMATION, name="tell_most_information")           # :539 (never fires)
    tell_playable_card = Rule(K.RULE_TELL_PLAYABLE_CARD, name="tell_playable_card")        # :570
    discard_randomly = Rule(K.RULE_DISCARD_RANDOMLY, name="discard_randomly")              # :607
    hail_mary = Rule(K.RULE_HAIL_MARY, name="hail_mary")                                   # :653

    @staticmethod
    def tell_dispensable_factory(min_information_tokens=8):                                # :454
        return Rule(K.RULE_TELL_DISPENSABLE, arg=int(min_information_tokens),
                    name=f"tell_dispensable({min_information_tokens})")

    @staticmethod
    def play_probably_safe_factory(treshold=0.95, require_extra_lives=False):               # :617 (sic: treshold)
        return Rule(K.RULE_PLAY_PROBABLY_SAFE, arg=int(bool(require_extra_lives)), threshold=float(treshold),
                    name=f"play_probably_safe({treshold}, {require_extra_lives})")

    @staticmethod
    def discard_probably_useless_factory(treshold=0.75):                                    # :638
        return Rule(K.RULE_DISCARD_PROBABLY_USELESS, threshold=float(treshold), name=f"discard_probably_useless({treshold})")
