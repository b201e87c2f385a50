"""Velocity control of a planar mobile base whose pose is three joints of the articulation -- x, y, yaw, in that order
(the Fetch: root_x_axis_joint, root_y_axis_joint, root_z_rotation_joint). Behavioural counterpart of
mani_skill/agents/controllers/pd_base_vel.py:11-74: the action is given in the base's own frame and turned into the three
joints' velocity-drive targets by the current yaw.

* `PDBaseVelController`: action = (forward, left, turn [, further joints]) velocities;
* `PDBaseForwardVelController`: action = (forward, turn): a differential-drive base that cannot move sideways.

The yaw enters, so this is not the affine map of the joint controllers; the native action map has two flags for the
forward-only controller (include/mssim.h set_action_map: the x / y joint take the forward column times cos / sin of the
yaw joint's position). The three-column controller has no native map (`fused_action_spec` None: the env applies the
action through this class and then launches the step).
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

    def fused_action_spec(self):
        """rows (dof, action column, low, high, flags) of the native action map: flag 8 = velocity target, 2 = clip + scale,
        16 / 32 = times cos / sin of the yaw joint (its index in bits 8..12)"""
        if len(self.joints) != 3:
            return None
        jx, jy, jyaw = self.active_joint_indices.tolist()
        norm = self._normalize_action
        lo = self.action_space_low.tolist() if norm else [0.0, 0.0]
        hi = self.action_space_high.tolist() if norm else [0.0, 0.0]
        base = 8 | (2 if norm else 0)
        return [
            (jx, 0, lo[0], hi[0], base | 16 | (jyaw << 8)),
            (jy, 0, lo[0], hi[0], base | 32 | (jyaw << 8)),
            (jyaw, 1, lo[1], hi[1], base),
        ]


@dataclass
class PDBaseVelControllerConfig(PDJointVelControllerConfig):
    controller_cls = PDBaseVelController


@dataclass
class PDBaseForwardVelControllerConfig(PDJointVelControllerConfig):
    controller_cls = PDBaseForwardVelController
