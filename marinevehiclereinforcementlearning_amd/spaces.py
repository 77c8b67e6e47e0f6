"""Observation/action spaces: gymnasium.spaces.Box / gym.spaces.Box when importable, else a minimal stand-in with
the attributes SB3 and the reference's helper scripts read (low, high, shape, dtype, sample, contains)."""
import numpy as np

try:  # pragma: no cover - neither package ships in the build image
    from gymnasium.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001
    try:
        from gym.spaces import Box  # type: ignore
    except Exception:  # noqa: BLE001
        class Box:  # minimal duck type of gym.spaces.Box
            def __init__(self, low, high, shape=None, dtype=np.float32):
                if shape is None:
                    shape = np.shape(low)
                self.shape = tuple(shape)
                self.dtype = np.dtype(dtype)
                self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
                self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
                self._rng = np.random.default_rng()

            def seed(self, seed=None):
                self._rng = np.random.default_rng(seed)
                return [seed]

            def sample(self):
                return self._rng.uniform(self.low, self.high).astype(self.dtype)

            def contains(self, x):
                x = np.asarray(x)
                return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

            def __repr__(self):
                return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def unit_box(dim):
    """Box(-1, 1, (dim,), float32): every env of the reference uses it for both spaces
    (6DoF.py:455-465, 3DoF.py:385-395, verySimpleAuv.py:128-145)."""
    return Box(low=-np.ones(dim, dtype=np.float32), high=np.ones(dim, dtype=np.float32), shape=(dim,), dtype=np.float32)
