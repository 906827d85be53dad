from . import split  # noqa: F401
