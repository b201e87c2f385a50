"""PD joint velocity controller (counterpart of mani_skill/agents/controllers/pd_joint_vel.py)."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig


class PDJointVelController(BaseController):
    config: "PDJointVelControllerConfig"

    def _initialize_action_space(self):
        n = len(self.joints)
        low = np.float32(np.broadcast_to(self.config.lower, n))
        high = np.float32(np.broadcast_to(self.config.upper, n))
        self.single_action_space = spaces.Box(low, high, dtype=np.float32)

    def set_drive_property(self):
        n = len(self.joints)
        d = np.broadcast_to(self.config.damping, n)
        f = np.broadcast_to(self.config.force_limit, n)
        fr = np.broadcast_to(self.config.friction, n)
        for i, joint in enumerate(self.joints):
            joint.set_drive_properties(0, d[i], force_limit=f[i], mode=self.config.drive_mode)
            joint.set_friction(fr[i])

    def set_action(self, action):
        action = self._preprocess_action(action)
        self.articulation.set_joint_drive_velocity_targets(action, self.joints, self.active_joint_indices)

    def fused_action_spec(self):
        """[(dof, local action column, low, high, flags)] for the native action map: flag 8 = velocity target"""
        out = []
        for i, dof in enumerate(self.active_joint_indices.tolist()):
            lo = float(self.action_space_low[i]) if self._normalize_action else 0.0
            hi = float(self.action_space_high[i]) if self._normalize_action else 0.0
            out.append((dof, i, lo, hi, 8 | (2 if self._normalize_action else 0)))
        return out


@dataclass
class PDJointVelControllerConfig(ControllerConfig):
    lower: Union[float, Sequence[float]]
    upper: Union[float, Sequence[float]]
    damping: Union[float, Sequence[float]]
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    normalize_action: bool = True
    drive_mode: str = "force"
    controller_cls = PDJointVelController
