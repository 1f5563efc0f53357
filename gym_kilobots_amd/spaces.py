"""`spaces.Box` as the reference uses it (gym.spaces.Box(low, high, dtype)).  gym itself is an
optional dependency: if it is importable its Box is used, otherwise this minimal equivalent."""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is not None and np.isscalar(low):
                low = np.full(shape, low, dtype=dtype)
                high = np.full(shape, high, dtype=dtype)
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            if self.low.shape != self.high.shape:
                raise ValueError('low and high must have the same shape')
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)
            self._rng = np.random.RandomState()

        def seed(self, seed=None):
            self._rng = np.random.RandomState(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return 'Box(%s, %s, %s)' % (self.low.min(), self.high.max(), self.shape)

        def __eq__(self, other):
            return isinstance(other, Box) and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)
