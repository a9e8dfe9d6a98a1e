"""``SensorModel`` base class, re-exported at the reference's module path."""
from ..._interfaces import SensorModel  # noqa: F401
