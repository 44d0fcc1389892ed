"""Stand-in package (build host only): see units.py."""
from . import units, constants, table  # noqa: F401
