"""PD joint velocity controller: the action is the drives' velocity target (zero stiffness).
Behavioural counterpart of mani_skill/agents/controllers/pd_joint_vel.py, built on joint_drive.py."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig
from .joint_drive import apply_joint_gains, fused_joint_columns


class PDJointVelController(BaseController):
    config: "PDJointVelControllerConfig"

    def _initialize_action_space(self):
        bounds = [np.broadcast_to(b, len(self.joints)).astype(np.float32) for b in (self.config.lower, self.config.upper)]
        self.single_action_space = spaces.Box(*bounds, dtype=np.float32)

    def set_drive_property(self):
        c = self.config
        apply_joint_gains(self.joints, stiffness=0.0, damping=c.damping, force_limit=c.force_limit, friction=c.friction, drive_mode=c.drive_mode)

    def set_action(self, action):
        self.articulation.set_joint_drive_velocity_targets(self._preprocess_action(action), self.joints, self.active_joint_indices)

    def fused_action_spec(self):
        """rows for the native action map: flag 8 = the value is the joint's velocity drive target"""
        return fused_joint_columns(self.active_joint_indices, len(self.joints), getattr(self, "action_space_low", None), getattr(self, "action_space_high", None), self._normalize_action, 8)


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
