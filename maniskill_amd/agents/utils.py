"""joint lookup + action-space flattening (counterpart of mani_skill/agents/utils.py:10-56)"""
from typing import Dict, Sequence

import numpy as np
import torch
from gymnasium import spaces


def get_active_joint_indices(articulation, joint_names: Sequence[str]) -> torch.Tensor:
    names = [j.name for j in articulation.get_active_joints()]
    return torch.tensor([names.index(n) for n in joint_names], dtype=torch.int32)


def get_joints_by_names(articulation, joint_names: Sequence[str]):
    m = {j.name: j for j in articulation.get_active_joints()}
    return [m[n] for n in joint_names]


def flatten_action_spaces(action_spaces: Dict[str, spaces.Space]):
    """concatenate 1-D Box spaces in dict order -> (Box, {uid: (start, end)})"""
    lows, highs, mapping, offset = [], [], {}, 0
    for uid, sp in action_spaces.items():
        assert isinstance(sp, spaces.Box) and len(sp.shape) == 1, (uid, sp)
        lows.append(sp.low)
        highs.append(sp.high)
        mapping[uid] = (offset, offset + sp.shape[0])
        offset += sp.shape[0]
    return spaces.Box(np.concatenate(lows), np.concatenate(highs), shape=[offset], dtype=np.float32), mapping
