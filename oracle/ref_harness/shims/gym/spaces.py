"""Holder classes standing in for gym.spaces (oracle harness only)."""


class Discrete:
    def __init__(self, n, *a, **k):
        self.n = n


class Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None, *a, **k):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class Dict(dict):
    def __init__(self, d=None, *a, **k):
        super().__init__(d or {})
