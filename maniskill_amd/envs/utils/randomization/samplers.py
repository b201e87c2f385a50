"""rejection sampler for non-overlapping placements (API of
mani_skill/envs/utils/randomization/samplers.py UniformPlacementSampler)"""
import torch


class UniformPlacementSampler:
    def __init__(self, bounds, batch_size: int, device=None):
        self._bounds = torch.tensor(bounds, dtype=torch.float32, device=device)
        self._ranges = self._bounds[1] - self._bounds[0]
        self.fixtures_radii = None
        self.fixture_positions = None
        self.batch_size = batch_size
        self.device = device

    def sample(self, radius, max_trials, append=True, verbose=False):
        if self.fixture_positions is None:
            sampled = torch.rand((self.batch_size, 2), device=self.device) * self._ranges + self._bounds[0]
        else:
            pass_mask = torch.zeros(self.batch_size, dtype=torch.bool, device=self.device)
            sampled = torch.zeros((self.batch_size, 2), device=self.device)
            for _ in range(max_trials):
                pos = torch.rand((self.batch_size, 2), device=self.device) * self._ranges + self._bounds[0]
                dist = torch.linalg.norm(pos - self.fixture_positions, axis=-1)
                ok = torch.all(dist > self.fixtures_radii + radius, axis=0)
                take = ok & ~pass_mask
                sampled[take] = pos[take]
                pass_mask = pass_mask | ok
                if bool(pass_mask.all()):
                    break
            sampled[~pass_mask] = pos[~pass_mask]
        if append:
            if self.fixture_positions is None:
                self.fixture_positions = sampled[None, ...]
            else:
                self.fixture_positions = torch.concat([self.fixture_positions, sampled[None, ...]])
            r = torch.tensor(radius, device=self.device).reshape(1, 1)
            self.fixtures_radii = r if self.fixtures_radii is None else torch.concat([self.fixtures_radii, r])
        return sampled
