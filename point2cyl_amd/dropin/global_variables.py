"""Drop-in for global_variables.py: the only constant the hot path reads; anything else from the shadowed reference module."""
g_zero_tol = 1.0e-6
from point2cyl_amd._shadow import reexport as _reexport  # noqa: E402

_reexport("global_variables", __file__, globals())
