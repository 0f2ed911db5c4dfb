"""One replayed step: the record type exchanged between replay and learner.

Field names and order follow hanabi_agents/rlax_dqn/transition.py:12-14:
  observation_tm1  observation the action was taken from          [B, obs_len]
  action_tm1       move uid                                        [B, 1]
  reward_t         reward collected until the seat's next turn     [B, 1]
  observation_t    the seat's next observation                     [B, obs_len]
  legal_moves_t    0/1 legal-move mask belonging to observation_t  [B, n_actions]
  terminal_t       episode ended between the two observations      [B, 1]
"""
import collections

Transition = collections.namedtuple(
    "Transition",
    "observation_tm1 action_tm1 reward_t observation_t legal_moves_t terminal_t",
)
