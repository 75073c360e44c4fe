"""Minimal stand-in for the `gym` package (not installed in this image).

ORACLE HARNESS ONLY.  The reference env subclasses gym.Env and builds
gym.spaces objects in its constructor; nothing else of gym is used on the
path we capture golden vectors from.  This is not reference code.
"""
from . import spaces  # noqa: F401


class Env:
    metadata = {}

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, *a, **k):
        raise NotImplementedError
