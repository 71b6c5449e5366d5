"""gym.spaces.Box / gym.Env when gym is installed; otherwise minimal stand-ins: a Box with the attributes PPO code
reads (low, high, shape, dtype, sample, contains) and an empty Env base. The reference derives its env from gym.Env
(trex_env.py:25) and builds spaces.Box(low=..., high=..., dtype=np.float32) at trex_env.py:93-96; TrexBulletEnv and
TrexVecEnv derive from `Env` below, so that isinstance(env, gym.Env) holds wherever gym exists."""
import numpy as np

try:  # pragma: no cover - gym is not in this image
    from gym import Env  # type: ignore
except Exception:  # noqa: BLE001

    class Env:
        """Stand-in for gym.Env (gym absent): the old-gym surface the reference uses is implemented by the subclasses."""
        metadata = {"render.modes": []}
        reward_range = (-float("inf"), float("inf"))
        spec = None

try:  # pragma: no cover - gym is not in this image
    from gym.spaces import Box  # type: ignore
except Exception:  # noqa: BLE001

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            if shape is not None:
                low = np.broadcast_to(low, shape).copy()
                high = np.broadcast_to(high, shape).copy()
            assert low.shape == high.shape
            self.low, self.high = low, high
            self.shape = low.shape
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box(%s, %s)" % (self.shape, self.dtype)
