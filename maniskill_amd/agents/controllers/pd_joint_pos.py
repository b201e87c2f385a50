"""PD joint position (+ delta, target-delta, mimic) controller -- counterpart of
mani_skill/agents/controllers/pd_joint_pos.py:14-131."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
import torch
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig


class PDJointPosController(BaseController):
    config: "PDJointPosControllerConfig"
    _start_qpos = None
    _target_qpos = None

    def _get_joint_limits(self):
        qlimits = self.articulation.get_qlimits()[0, self.active_joint_indices.long()].cpu().numpy()
        if self.config.lower is not None:
            qlimits[:, 0] = self.config.lower
        if self.config.upper is not None:
            qlimits[:, 1] = self.config.upper
        return qlimits

    def _initialize_action_space(self):
        lim = self._get_joint_limits()
        self.single_action_space = spaces.Box(lim[:, 0], lim[:, 1], dtype=np.float32)

    def set_drive_property(self):
        n = len(self.joints)
        k = np.broadcast_to(self.config.stiffness, n)
        d = np.broadcast_to(self.config.damping, n)
        f = np.broadcast_to(self.config.force_limit, n)
        fr = np.broadcast_to(self.config.friction, n)
        for i, joint in enumerate(self.joints):
            mode = self.config.drive_mode if isinstance(self.config.drive_mode, str) else self.config.drive_mode[i]
            joint.set_drive_properties(k[i], d[i], force_limit=f[i], mode=mode)
            joint.set_friction(fr[i])

    def reset(self):
        super().reset()
        self._step = 0
        if self._start_qpos is None or self.scene._reset_mask_all:
            self._start_qpos = self.qpos.clone()
            self._target_qpos = self.qpos.clone()
        else:
            m = self.scene._reset_idx
            self._start_qpos[m] = self.qpos[m].clone()
            self._target_qpos[m] = self.qpos[m].clone()

    def set_drive_targets(self, targets):
        self.articulation.set_joint_drive_targets(targets, self.joints, self.active_joint_indices)

    def set_action(self, action):
        action = self._preprocess_action(action)
        self._step = 0
        # `qpos` is a view of the sim buffer here (the reference gets a gathered copy)
        self._start_qpos = self.qpos.clone() if self.config.interpolate else self.qpos
        if self.config.use_delta:
            if self.config.use_target:
                self._target_qpos = self._target_qpos + action
            else:
                self._target_qpos = self._start_qpos + action
        else:
            self._target_qpos = torch.broadcast_to(action, self._start_qpos.shape).clone()
        if self.config.interpolate:
            self._step_size = (self._target_qpos - self._start_qpos) / self._sim_steps
        else:
            self.set_drive_targets(self._target_qpos)

    def before_simulation_step(self):
        self._step += 1
        if self.config.interpolate:
            self.set_drive_targets(self._start_qpos + self._step_size * self._step)
            self.articulation.px.gpu_apply_articulation_target_position()

    def fused_action_spec(self):
        """[(dof, local action column, low, high, flags)] for the fused native action kernel, or None
        when this controller keeps state across steps (use_target) or updates targets per substep"""
        if self.config.use_target or self.config.interpolate:
            return None
        nact = self.single_action_space.shape[0]
        out = []
        for i, dof in enumerate(self.active_joint_indices.tolist()):
            col = i if nact == len(self.joints) else 0  # mimic controllers broadcast one column
            lo = float(self.action_space_low[col]) if self._normalize_action else 0.0
            hi = float(self.action_space_high[col]) if self._normalize_action else 0.0
            flags = (1 if self.config.use_delta else 0) | (2 if self._normalize_action else 0)
            out.append((dof, col, lo, hi, flags))
        return out

    def get_state(self) -> dict:
        return {"target_qpos": self._target_qpos} if self.config.use_target else {}

    def set_state(self, state: dict):
        if self.config.use_target:
            self._target_qpos = state["target_qpos"]


@dataclass
class PDJointPosControllerConfig(ControllerConfig):
    lower: Union[None, float, Sequence[float]]
    upper: Union[None, float, Sequence[float]]
    stiffness: Union[float, Sequence[float]]
    damping: Union[float, Sequence[float]]
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    use_delta: bool = False
    use_target: bool = False
    interpolate: bool = False
    normalize_action: bool = True
    drive_mode: Union[Sequence[str], str] = "force"
    controller_cls = PDJointPosController


class PDJointPosMimicController(PDJointPosController):
    def _get_joint_limits(self):
        lim = super()._get_joint_limits()
        assert np.allclose(lim[0:-1] - lim[1:], 0), "Mimic joints should have the same limit"
        return lim[0:1]


class PDJointPosMimicControllerConfig(PDJointPosControllerConfig):
    controller_cls = PDJointPosMimicController
