"""Observation / action space descriptions for the gym-facing classes.

With gymnasium importable these ARE gymnasium's spaces (`make_*` return gymnasium.spaces objects, so an RL library that introspects
spaces sees what it expects).  gymnasium is not installed in the build image, so without it the same constructors return the small
stand-ins below: same attribute names (`shape`, `dtype`, `low`, `high`, `n`, `nvec`, `spaces`), `sample()` and `contains()` -- enough for
code that reads shapes and dtypes, and for this repository's own tests.  The reference defines its spaces at
/root/reference/gym_agario/AgarioEnv.py:55-62 (action) and :232-264 (observation).
"""
import numpy as np

try:  # optional
    from gymnasium import spaces as _gs
except Exception:  # pragma: no cover - gymnasium is not installed in the build image
    _gs = None

HAVE_GYMNASIUM = _gs is not None


class _Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(int(s) for s in shape), np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low.astype(np.float64)), self.low, -1e6).astype(np.float64)
        hi = np.where(np.isfinite(self.high.astype(np.float64)), self.high, 1e6).astype(np.float64)
        x = np.random.uniform(lo, hi)
        return (np.floor(x) if np.issubdtype(self.dtype, np.integer) else x).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.flat[0] if self.low.size else None, self.high.flat[0] if self.high.size else None, self.shape, self.dtype)


class _Discrete:
    def __init__(self, n):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return int(x) == x and 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class _MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape, self.dtype = self.nvec.shape, np.dtype(np.int64)

    def sample(self):
        return (np.random.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))

    def __repr__(self):
        return "MultiDiscrete(shape=%s)" % (self.shape,)


class _Tuple:
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __len__(self):
        return len(self.spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))

    def __repr__(self):
        return "Tuple(%s)" % ", ".join(repr(s) for s in self.spaces)


def Box(low, high, shape, dtype=np.float32):
    return _gs.Box(low=low, high=high, shape=tuple(shape), dtype=dtype) if _gs is not None else _Box(low, high, shape, dtype)


def Discrete(n):
    return _gs.Discrete(n) if _gs is not None else _Discrete(n)


def MultiDiscrete(nvec):
    return _gs.MultiDiscrete(nvec) if _gs is not None else _MultiDiscrete(nvec)


def Tuple(spaces):
    return _gs.Tuple(tuple(spaces)) if _gs is not None else _Tuple(spaces)


# observation bounds per observation kind: (low, high, dtype) -- AgarioEnv.py:232-264 for grid / screen / gobigger; "ram" is this library's own
OBS_BOUNDS = {"grid": (-1, np.iinfo(np.int32).max, np.int32), "screen": (0, 255, np.uint8), "ram": (-np.inf, np.inf, np.float32),
              "gobigger": (0, 255, np.float32)}


def single_action_space(num_agents=1, multi_agent=False):
    """((dx, dy) in [-1, 1]^2, kind in {0 none, 1 feed, 2 split}) -- AgarioEnv.py:55-62; with several agents one row per agent"""
    if not multi_agent:
        return Tuple((Box(-1, 1, (2,)), Discrete(3)))
    return Tuple((Box(-1, 1, (num_agents, 2)), MultiDiscrete([3] * num_agents)))


def batched_action_space(num_envs, num_agents=1, multi_agent=False):
    """the action space of `num_envs` arenas at once, as gymnasium.vector batches a Tuple(Box, Discrete): (Box [N, 2], MultiDiscrete [N])"""
    if not multi_agent:
        return Tuple((Box(-1, 1, (num_envs, 2)), MultiDiscrete([3] * num_envs)))
    return Tuple((Box(-1, 1, (num_envs, num_agents, 2)), MultiDiscrete(np.full((num_envs, num_agents), 3))))


def observation_space(kind, shape):
    lo, hi, dt = OBS_BOUNDS[kind]
    return Box(lo, hi, shape, dt)
