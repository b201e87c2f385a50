"""Batched articulation joints (API counterpart of mani_skill/utils/structs/articulation_joint.py:20-260): limits, drive
properties and targets of one joint of the articulation in every env."""
from typing import Optional, Sequence, Union

import numpy as np
import torch

from maniskill_amd.utils import common


class ArticulationJoint:
    def __init__(self, articulation, name: str, jtype: str, index: int, active_index: Optional[int], limits, child_link=None, parent_link=None):
        self.articulation = articulation
        self.scene = articulation.scene
        self.name = name
        self.type_str = jtype
        self.index_int = index
        self.active_index_int = active_index
        self._limits = limits  # (lo, hi) or None
        self.child_link = child_link
        self.parent_link = parent_link
        self._drive = [0.0, 0.0, float("inf"), "force"]
        self._friction = 0.0

    @property
    def device(self):
        return self.scene.device

    @property
    def type(self):
        return [self.type_str] * self.scene.num_envs

    @property
    def index(self):
        return torch.full((self.scene.num_envs,), self.index_int, dtype=torch.int, device=self.device)

    @property
    def active_index(self):
        if self.active_index_int is None:
            return None
        return torch.tensor([self.active_index_int], dtype=torch.int, device=self.device)

    @property
    def dof(self):
        return torch.full((self.scene.num_envs,), 0 if self.active_index_int is None else 1, dtype=torch.int, device=self.device)

    @property
    def limits(self) -> torch.Tensor:
        lo, hi = self._limits if self._limits is not None else (0.0, 0.0)
        return torch.tensor([[lo, hi]], dtype=torch.float32, device=self.device).repeat(self.scene.num_envs, 1)

    def get_limits(self):
        return self.limits

    @property
    def qpos(self):
        return self.articulation.qpos[:, self.active_index_int]

    @property
    def qvel(self):
        return self.articulation.qvel[:, self.active_index_int]

    # drive -----------------------------------------------------------------------
    def set_drive_properties(self, stiffness: float, damping: float, force_limit: float = 3.4028234663852886e38, mode: str = "force"):
        self._drive = [float(stiffness), float(damping), float(force_limit), mode]
        self.articulation._set_drive(self)

    def set_drive_property(self, stiffness, damping, force_limit=3.4028234663852886e38, mode="force"):
        self.set_drive_properties(stiffness, damping, force_limit, mode)

    def set_friction(self, friction: float):
        self._friction = float(friction)
        if friction != 0:
            import warnings

            warnings.warn("joint friction is not modelled by this simulation core yet; value recorded only")

    @property
    def stiffness(self):
        return torch.full((self.scene.num_envs,), self._drive[0], device=self.device)

    @property
    def damping(self):
        return torch.full((self.scene.num_envs,), self._drive[1], device=self.device)

    @property
    def force_limit(self):
        return torch.full((self.scene.num_envs,), self._drive[2], device=self.device)

    @property
    def friction(self):
        return torch.full((self.scene.num_envs,), self._friction, device=self.device)

    @property
    def drive_mode(self):
        return [self._drive[3]] * self.scene.num_envs

    @property
    def drive_target(self):
        return self.articulation.px.cuda_articulation_target_qpos.torch()[:, self.active_index_int]

    @property
    def drive_velocity_target(self):
        return self.articulation.px.cuda_articulation_target_qvel.torch()[:, self.active_index_int]

    def set_drive_target(self, target):
        self.articulation.set_joint_drive_targets(common.to_tensor(target, device=self.device).reshape(-1, 1), [self], self.active_index.long())

    def set_drive_velocity_target(self, target):
        self.articulation.set_joint_drive_velocity_targets(common.to_tensor(target, device=self.device).reshape(-1, 1), [self], self.active_index.long())

    def get_name(self):
        return self.name

    def __repr__(self):
        return f"<ArticulationJoint {self.name}>"
