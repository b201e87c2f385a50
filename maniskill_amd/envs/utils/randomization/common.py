"""uniform sampler (counterpart of mani_skill/envs/utils/randomization/common.py:9-20)"""
import torch

from maniskill_amd.utils import common


def uniform(low, high, size, device=None):
    if not isinstance(low, float):
        low = common.to_tensor(low, device=device)
    if not isinstance(high, float):
        high = common.to_tensor(high, device=device)
    return torch.rand(size=size, device=device) * (high - low) + low
