"""Rule-based partners (reference: hanabi_agents/rule_based/__init__.py:6-7), evaluated on the GPU."""
from .rule_based import RulebasedAgent
from .ruleset import Rule, Ruleset
from . import predefined_rules
