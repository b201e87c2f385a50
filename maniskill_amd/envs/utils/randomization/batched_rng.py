"""One numpy RandomState per env so an N-env run draws what N single-env runs would draw
(counterpart of mani_skill/envs/utils/randomization/batched_rng.py:15-70)."""
from typing import List

import numpy as np

from maniskill_amd.utils import common


class BatchedRNG:
    def __init__(self, rngs: List[np.random.RandomState]):
        self.rngs = list(rngs)
        self.batch_size = len(self.rngs)

    @classmethod
    def from_seeds(cls, seeds, backend: str = "numpy:random_state"):
        if backend != "numpy:random_state":
            raise ValueError(f"Unknown batched RNG backend: {backend}")
        return cls([np.random.RandomState(int(s)) for s in seeds])

    @classmethod
    def from_rngs(cls, rngs):
        return cls(rngs)

    @staticmethod
    def expand_seeds(seed, n: int) -> np.ndarray:
        """the seeding rule of the env (mani_skill/envs/sapien_env.py:881-917): one seed per env; a scalar or a single
        seed is used for env 0 and the other n - 1 are drawn from `RandomState(that seed)`"""
        seeds = np.atleast_1d(common.to_numpy(seed)).astype(np.int64)
        if len(seeds) == 1 and n > 1:
            seeds = np.concatenate((seeds, np.random.RandomState(int(seeds[0])).randint(2**31, size=(n - 1,))))
        return seeds

    def reseed(self, idx, seeds):
        """new streams for the envs in `idx`"""
        self[idx] = BatchedRNG.from_seeds(seeds)

    def __getitem__(self, idx):
        idx = common.to_numpy(idx)
        if np.iterable(idx):
            return BatchedRNG([self.rngs[int(i)] for i in idx])
        return self.rngs[int(idx)]

    def __setitem__(self, idx, value):
        idx = common.to_numpy(idx)
        if np.iterable(idx):
            vals = value.rngs if isinstance(value, BatchedRNG) else value
            for i, v in zip(idx, vals):
                self.rngs[int(i)] = v
        else:
            self.rngs[int(idx)] = value

    def __len__(self):
        return len(self.rngs)

    def __getattr__(self, item):
        # only reached for names not defined on BatchedRNG: forward to every per-env RandomState
        attr = getattr(self.rngs[0], item)
        if callable(attr):
            def method(*args, **kwargs):
                return np.array([getattr(r, item)(*args, **kwargs) for r in self.rngs])

            return method
        return np.array([getattr(r, item) for r in self.rngs])
