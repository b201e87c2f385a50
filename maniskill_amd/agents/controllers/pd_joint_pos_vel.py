"""PD joint position + velocity controller (counterpart of
mani_skill/agents/controllers/pd_joint_pos_vel.py): action = [target qpos | target qvel]."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
import torch
from gymnasium import spaces

from .pd_joint_pos import PDJointPosController, PDJointPosControllerConfig


class PDJointPosVelController(PDJointPosController):
    config: "PDJointPosVelControllerConfig"
    _target_qvel = None

    def _initialize_action_space(self):
        lim = self._get_joint_limits()
        n = len(self.joints)
        vlo = np.broadcast_to(self.config.vel_lower, n)
        vhi = np.broadcast_to(self.config.vel_upper, n)
        self.single_action_space = spaces.Box(np.float32(np.hstack([lim[:, 0], vlo])), np.float32(np.hstack([lim[:, 1], vhi])), dtype=np.float32)

    def reset(self):
        super().reset()
        if self._target_qvel is None or self.scene._reset_mask_all:
            self._target_qvel = torch.zeros_like(self._target_qpos)
        else:
            self._target_qvel[self.scene._reset_idx] = 0

    def set_action(self, action):
        action = self._preprocess_action(action)
        n = len(self.joints)
        self._step = 0
        self._start_qpos = self.qpos
        if self.config.use_delta:
            self._target_qpos = (self._target_qpos if self.config.use_target else self._start_qpos) + action[:, :n]
        else:
            self._target_qpos = torch.broadcast_to(action[:, :n], self._start_qpos.shape).clone()
        self._target_qvel = action[:, n:]
        self.set_drive_targets(self._target_qpos)
        self.articulation.set_joint_drive_velocity_targets(self._target_qvel, self.joints, self.active_joint_indices)


@dataclass
class PDJointPosVelControllerConfig(PDJointPosControllerConfig):
    controller_cls = PDJointPosVelController
    vel_lower: Union[float, Sequence[float]] = -1.0
    vel_upper: Union[float, Sequence[float]] = 1.0
