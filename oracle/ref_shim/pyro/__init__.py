"""Stub of ``pyro`` for importing the reference in this container (test infrastructure only)."""
from . import distributions  # noqa: F401


def factor(*args, **kwargs):  # only used by the (out of scope) pyro-VI hook
    raise NotImplementedError("pyro stub")
