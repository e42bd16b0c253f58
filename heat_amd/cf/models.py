"""Import-compatible alias (the reference keeps these names in cf/models.py); implementation: heat_amd.cf.frontend."""
from .frontend import Model, MatrixFactorization  # noqa: F401
