"""`gymnasium` when it is installed, otherwise the few names of it the aviaries use.

The reference's envs derive from `gymnasium.Env` and describe their spaces with
`gymnasium.spaces.Box` (`envs/BaseAviary.py:13,19`, `envs/BaseRLAviary.py:4,156,277`).
gymnasium is an optional dependency here: when it is importable the real classes are
re-exported (so SB3 / `gymnasium.make` see genuine `Env`s); when it is not, a minimal
duck-typed stand-in with the same attributes is provided so the simulator still runs.
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    from gymnasium import spaces  # noqa: F401
    Env = _gym.Env
    HAVE_GYMNASIUM = True

    def register(id, entry_point):  # noqa: A002 - gymnasium's own keyword
        from gymnasium.envs.registration import register as _register, registry
        if id not in registry:
            _register(id=id, entry_point=entry_point)

except ImportError:
    HAVE_GYMNASIUM = False

    class Env:
        """Subset of `gymnasium.Env`: metadata, seeding via `reset(seed=...)`, context manager."""
        metadata = {"render_modes": []}
        render_mode = None
        spec = None
        action_space = None
        observation_space = None
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.default_rng()
            return self._np_random

        @property
        def unwrapped(self):
            return self

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.default_rng(seed)
            return None

        def step(self, action):
            raise NotImplementedError

        def render(self):
            raise NotImplementedError

        def close(self):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self.close()
            return False

    class _Box:
        """Subset of `gymnasium.spaces.Box`: bounds, shape, dtype, sample, contains."""

        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(int(s) for s in shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            u = self._rng.uniform(size=self.shape)
            x = lo + (hi - lo) * u
            unb = ~np.isfinite(self.low) | ~np.isfinite(self.high)
            if unb.any():
                x = np.where(unb, self._rng.normal(size=self.shape), x)
            return x.astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        __contains__ = contains

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class _Spaces:
        Box = _Box

    spaces = _Spaces()
    _REGISTRY = {}

    def register(id, entry_point):  # noqa: A002
        _REGISTRY[id] = entry_point

    def make(id, **kwargs):  # noqa: A002
        mod, name = _REGISTRY[id].split(":")
        import importlib
        return getattr(importlib.import_module(mod), name)(**kwargs)
