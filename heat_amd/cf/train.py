"""Import-compatible alias (the reference keeps these names in cf/train.py); implementation: heat_amd.cf.frontend."""
from .frontend import Engine  # noqa: F401
