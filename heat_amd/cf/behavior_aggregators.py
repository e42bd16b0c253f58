"""Import-compatible alias (the reference keeps these names in cf/behavior_aggregators.py); implementation: heat_amd.cf.frontend."""
from .frontend import AggregatorWeights  # noqa: F401
