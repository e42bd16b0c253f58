"""Import-compatible alias (the reference keeps these names in cf/cf_config.py); implementation: heat_amd.cf.frontend."""
from .frontend import CFConfig  # noqa: F401
