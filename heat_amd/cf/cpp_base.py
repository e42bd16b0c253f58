"""Import-compatible alias (the reference keeps these names in cf/cpp_base.py); implementation: heat_amd.cf.frontend."""
from .frontend import CPPBase  # noqa: F401
