"""`pyamg` stand-in (absent from this image). TEST INFRASTRUCTURE ONLY."""
from . import classical  # noqa: F401
