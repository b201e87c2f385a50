"""Shared machinery of the joint-space PD controllers (pd_joint_pos / pd_joint_vel / pd_joint_pos_vel).

Two things every one of them does are factored out here instead of being spelled out per controller:

* `apply_joint_gains` -- one table of per-joint drive parameters (scalars broadcast over the joints), written to the
  joints in one pass;
* `TargetTrack` -- the pair (start, target) of joint-position tensors a position controller carries between control
  steps, with the partial-reset rule (only the envs selected by the scene's reset mask are re-captured) in one place;
* `TARGET_RULES` -- how an action becomes a position target, keyed by (use_delta, use_target).

Behavioural contract: mani_skill/agents/controllers/pd_joint_pos.py:35-98, pd_joint_vel.py:20-33.
"""
from typing import Callable, Dict, Sequence, Tuple

import numpy as np
import torch


def apply_joint_gains(joints: Sequence, *, stiffness, damping, force_limit, friction, drive_mode) -> None:
    """`joint.set_drive_properties(k, d, force_limit, mode)` + `joint.set_friction(f)` for every joint; every argument
    is a scalar or one value per joint (the drive mode a string or one string per joint)"""
    n = len(joints)
    table = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in (stiffness, damping, force_limit, friction)], axis=1)
    modes = [drive_mode] * n if isinstance(drive_mode, str) else list(drive_mode)
    for joint, (k, d, f, fr), mode in zip(joints, table, modes):
        joint.set_drive_properties(k, d, force_limit=f, mode=mode)
        joint.set_friction(fr)


class TargetTrack:
    """start / target joint positions of a position controller, [N, n_joints] each"""

    def __init__(self):
        self.start: torch.Tensor = None
        self.target: torch.Tensor = None

    def capture(self, qpos: torch.Tensor, rows: torch.Tensor = None) -> None:
        """take the current joint positions as start and target -- for every env (`rows` None, or nothing captured
        yet) or for the envs in `rows` only (partial reset)"""
        if self.start is None or rows is None:
            self.start, self.target = qpos.clone(), qpos.clone()
        else:
            self.start[rows] = qpos[rows]
            self.target[rows] = qpos[rows]


# action -> position target. Arguments: action [N, k] (k = n_joints, or 1 for mimic joints), the qpos the control step
# starts from, the previous target.
TARGET_RULES: Dict[Tuple[bool, bool], Callable] = {
    (False, False): lambda action, start, prev: torch.broadcast_to(action, start.shape).clone(),  # absolute
    (False, True): lambda action, start, prev: torch.broadcast_to(action, start.shape).clone(),
    (True, False): lambda action, start, prev: start + action,  # delta on the current position
    (True, True): lambda action, start, prev: prev + action,  # delta on the previous target
}


def fused_joint_columns(active_joint_indices, n_action: int, low, high, normalized: bool, flags: int):
    """rows `(dof, action column, low, high, flags)` of the native action map (include/mssim.h set_action_map) for a
    controller whose joint i reads action column i (or column 0 when one column drives every joint: mimic)"""
    dofs = active_joint_indices.tolist()
    rows = []
    for i, dof in enumerate(dofs):
        col = i if n_action == len(dofs) else 0
        rows.append((dof, col, float(low[col]) if normalized else 0.0, float(high[col]) if normalized else 0.0, flags | (2 if normalized else 0)))
    return rows
