"""Stand-in for `torch_geometric` (absent from this image). TEST INFRASTRUCTURE ONLY.

Only the four names the reference imports are provided.
"""
from . import nn, utils, data  # noqa: F401
