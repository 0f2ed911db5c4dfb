"""hanabi_agents — MI355X-native counterpart of the reference package of the same name.
Only the rlax_dqn agent family (the hot path) is provided."""
