"""Velocity control of a planar mobile base whose pose is three joints of the articulation -- x, y, yaw, in that order
(the Fetch: root_x_axis_joint, root_y_axis_joint, root_z_rotation_joint). Behavioural counterpart of
mani_skill/agents/controllers/pd_base_vel.py:11-74: the action is given in the base's own frame and turned into the three
joints' velocity-drive targets by the current yaw.

* `PDBaseVelController`: action = (forward, left, turn [, further joints]) velocities;
* `PDBaseForwardVelController`: action = (forward, turn): a differential-drive base that cannot move sideways.

Neither is an affine action -> target map (the yaw enters), so there is no native action map for them
(`fused_action_spec` is None: the env applies the action through this class and then launches the step).
"""
from dataclasses import dataclass

import numpy as np
import torch
from gymnasium import spaces

from .pd_joint_vel import PDJointVelController, PDJointVelControllerConfig


def _base_frame_to_world(forward: torch.Tensor, left: torch.Tensor, yaw: torch.Tensor):
    """planar velocity (forward, left) of a base turned by `yaw` -> (x, y) in the frame the base joints move in"""
    c, s = torch.cos(yaw), torch.sin(yaw)
    return c * forward - s * left, s * forward + c * left


class PDBaseVelController(PDJointVelController):
    config: "PDBaseVelControllerConfig"

    def _initialize_action_space(self):
        if len(self.joints) < 3:
            raise AssertionError(f"a planar base needs its x, y and yaw joints, got {len(self.joints)}")
        super()._initialize_action_space()

    def _joint_velocities(self, action: torch.Tensor) -> torch.Tensor:
        action = action.float()
        vx, vy = _base_frame_to_world(action[:, 0], action[:, 1], self.qpos[:, 2])
        return torch.cat([vx[:, None], vy[:, None], action[:, 2:]], dim=1)

    def set_action(self, action):
        targets = self._joint_velocities(self._preprocess_action(action))
        self.articulation.set_joint_drive_velocity_targets(targets, self.joints, self.active_joint_indices)

    def fused_action_spec(self):
        return None


class PDBaseForwardVelController(PDBaseVelController):
    config: "PDBaseForwardVelControllerConfig"

    def _initialize_action_space(self):
        if len(self.joints) < 3:
            raise AssertionError(f"a planar base needs its x, y and yaw joints, got {len(self.joints)}")
        bounds = [np.broadcast_to(b, 2).astype(np.float32) for b in (self.config.lower, self.config.upper)]
        self.single_action_space = spaces.Box(*bounds, dtype=np.float32)

    def _joint_velocities(self, action: torch.Tensor) -> torch.Tensor:
        action = action.float()
        vx, vy = _base_frame_to_world(action[:, 0], torch.zeros_like(action[:, 0]), self.qpos[:, 2])
        return torch.cat([vx[:, None], vy[:, None], action[:, 1:]], dim=1)


@dataclass
class PDBaseVelControllerConfig(PDJointVelControllerConfig):
    controller_cls = PDBaseVelController


@dataclass
class PDBaseForwardVelControllerConfig(PDJointVelControllerConfig):
    controller_cls = PDBaseForwardVelController
