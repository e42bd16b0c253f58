"""Import-compatible alias (the reference keeps these names in cf/utils.py); implementation: heat_amd.cf.frontend."""
from .frontend import load_config  # noqa: F401
