"""PD joint position + velocity controller (counterpart of
mani_skill/agents/controllers/pd_joint_pos_vel.py): action = [target qpos | target qvel]."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
import torch
from gymnasium import spaces

from .pd_joint_pos import PDJointPosController, PDJointPosControllerConfig


class PDJointPosVelController(PDJointPosController):
    """the action is [position part | velocity part]: the first half goes through the position controller's target
    rule, the second half is the drives' velocity target"""

    config: "PDJointPosVelControllerConfig"
    _target_qvel = None

    def _initialize_action_space(self):
        n = len(self.joints)
        pos_lo, pos_hi = self._get_joint_limits().T
        vel_lo, vel_hi = (np.broadcast_to(v, n) for v in (self.config.vel_lower, self.config.vel_upper))
        self.single_action_space = spaces.Box(np.concatenate([pos_lo, vel_lo]).astype(np.float32), np.concatenate([pos_hi, vel_hi]).astype(np.float32), dtype=np.float32)

    def reset(self):
        super().reset()
        fresh = self._target_qvel is None or self.scene._reset_mask_all
        if fresh:
            self._target_qvel = torch.zeros_like(self._target_qpos)
        else:
            self._target_qvel[self.scene._reset_idx] = 0

    def set_action(self, action):
        n = len(self.joints)
        action = self._preprocess_action(action)
        pos_part, vel_part = action[:, :n], action[:, n:]
        self._step, self._start_qpos = 0, self.qpos
        if not self.config.use_delta:
            self._target_qpos = torch.broadcast_to(pos_part, self._start_qpos.shape).clone()
        else:
            base = self._target_qpos if self.config.use_target else self._start_qpos
            self._target_qpos = base + pos_part
        self._target_qvel = vel_part
        self.set_drive_targets(self._target_qpos)
        self.articulation.set_joint_drive_velocity_targets(vel_part, self.joints, self.active_joint_indices)


@dataclass
class PDJointPosVelControllerConfig(PDJointPosControllerConfig):
    controller_cls = PDJointPosVelController
    vel_lower: Union[float, Sequence[float]] = -1.0
    vel_upper: Union[float, Sequence[float]] = 1.0
