"""Drop-in for the reference's IGR/sampler.py (the off-surface sampler of the implicit losses)."""
from point2cyl_amd.implicit import NormalPerPoint  # noqa: F401
