"""``Sensor`` base class, re-exported at the reference's module path."""
from .._interfaces import Sensor  # noqa: F401
