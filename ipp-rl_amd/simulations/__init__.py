"""``Simulation`` base class, re-exported at the reference's module path (the matplotlib viewer is not mirrored)."""
from .._interfaces import Simulation  # noqa: F401
