"""Observation/action spaces.  Uses gymnasium.spaces when gymnasium is importable, otherwise a minimal
stand-in with the same attributes (shape, low, high, dtype, contains, sample) so the package works in
images without gymnasium (the build image and the GPU boxes of this project have none)."""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    from gymnasium import spaces as _spaces

    HAVE_GYMNASIUM = True
    Box = _spaces.Box
    Dict = _spaces.Dict
    EnvBase = _gym.Env
except Exception:  # gymnasium absent
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
            self._rng = np.random.default_rng(seed)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and x.dtype == self.dtype and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            if np.issubdtype(self.dtype, np.integer):
                return self._rng.integers(self.low, self.high, endpoint=True, size=self.shape).astype(self.dtype)
            return self._rng.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(dict):
        @property
        def spaces(self):
            return self

        def contains(self, x):
            return isinstance(x, dict) and x.keys() == self.keys() and all(self[k].contains(x[k]) for k in self)

        def sample(self):
            return {k: s.sample() for k, s in self.items()}

    class EnvBase:
        metadata = {}
        render_mode = None

        def reset(self, *, seed=None, options=None):
            return None

        def close(self):
            pass
