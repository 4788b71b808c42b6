"""Observation / action spaces (reference: accel_rl/spaces/discrete.py,
accel_rl/spaces/uintbox.py).  Host-side objects; `sample()` draws from the global
numpy RNG with exactly the reference's calls so stream positions stay aligned."""
import numpy as np


class Discrete(object):
    """{0, ..., n-1}; dtype chosen by n (reference: spaces/discrete.py:12-20)."""

    def __init__(self, n):
        self._n = int(n)
        self._dtype = "uint8" if n <= 2 ** 8 else ("uint16" if n <= 2 ** 16 else "uint32")

    n = property(lambda self: self._n)
    dtype = property(lambda self: self._dtype)
    flat_dim = property(lambda self: self._n)
    default_value = property(lambda self: 0)

    def sample(self):
        return np.random.randint(self._n, dtype=self._dtype)

    def sample_n(self, n):
        return np.random.randint(low=0, high=self._n, size=n, dtype=self._dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == () and x.dtype.kind in "iu" and 0 <= x < self._n

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self._n

    def __hash__(self):
        return hash(self._n)

    def __repr__(self):
        return "Discrete(%d)" % self._n


class UintBox(object):
    """Unsigned-integer box with one (low, high) for all coordinates
    (reference: spaces/uintbox.py:16-43)."""

    def __init__(self, shape, low=0, high=None, bits=8):
        if bits not in (8, 16, 32, 64):
            raise ValueError("bits must be 8/16/32/64")
        self.dtype = "uint%d" % bits
        top = 2 ** bits - 1
        self.low = np.asarray(low, dtype=self.dtype)
        self.high = np.asarray(top if high is None else high, dtype=self.dtype)
        if not (0 <= self.low < self.high <= top):
            raise ValueError("need 0 <= low < high <= %d" % top)
        self._shape = tuple(shape)

    shape = property(lambda self: self._shape)
    flat_dim = property(lambda self: int(np.prod(self._shape)))
    bounds = property(lambda self: (self.low, self.high))

    def sample(self):
        return np.random.randint(low=self.low, high=self.high, size=self._shape, dtype=self.dtype)

    def sample_n(self, n):
        return np.random.randint(low=self.low, high=self.high, size=(n,) + self._shape,
                                 dtype=self.dtype)

    def contains(self, x):
        return x.shape == self._shape and (x >= self.low).all() and (x <= self.high).all()

    def __eq__(self, other):
        return (isinstance(other, UintBox) and self.dtype == other.dtype and
                self.low == other.low and self.high == other.high and self._shape == other._shape)

    def __hash__(self):
        return hash((int(self.low), int(self.high), self._shape))

    def __repr__(self):
        return "Uint%sBox%s" % (self.dtype[4:], self._shape)


class EnvSpec(object):
    """(observation_space, action_space) pair (reference: rllab/envs/env_spec.py:5-25)."""

    def __init__(self, observation_space, action_space):
        self.observation_space = observation_space
        self.action_space = action_space
