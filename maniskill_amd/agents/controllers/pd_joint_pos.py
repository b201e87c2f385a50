"""PD joint position controllers: absolute, delta, target-delta, mimic, optionally interpolated over the substeps.
Behavioural counterpart of mani_skill/agents/controllers/pd_joint_pos.py:14-131, built on joint_drive.py."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig
from .joint_drive import TARGET_RULES, TargetTrack, apply_joint_gains, fused_joint_columns


class PDJointPosController(BaseController):
    config: "PDJointPosControllerConfig"

    # ---- spaces / gains ---------------------------------------------------------------------------------------------
    def _get_joint_limits(self) -> np.ndarray:
        """[n_joints, 2]: the articulation's limits, overridden by the config's `lower` / `upper` where given"""
        lim = self.articulation.get_qlimits()[0, self.active_joint_indices.long()].cpu().numpy()
        for col, override in enumerate((self.config.lower, self.config.upper)):
            if override is not None:
                lim[:, col] = override
        return lim

    def _initialize_action_space(self):
        low, high = self._get_joint_limits().T
        self.single_action_space = spaces.Box(low, high, dtype=np.float32)

    def set_drive_property(self):
        c = self.config
        apply_joint_gains(self.joints, stiffness=c.stiffness, damping=c.damping, force_limit=c.force_limit, friction=c.friction, drive_mode=c.drive_mode)

    # ---- targets ----------------------------------------------------------------------------------------------------
    @property
    def _track(self) -> TargetTrack:
        t = self.__dict__.get("_track_obj")
        if t is None:
            t = self.__dict__["_track_obj"] = TargetTrack()
        return t

    # (the names the subclasses -- pd_ee_pose, pd_joint_pos_vel -- and the tests read and write)
    @property
    def _start_qpos(self):
        return self._track.start

    @_start_qpos.setter
    def _start_qpos(self, v):
        self._track.start = v

    @property
    def _target_qpos(self):
        return self._track.target

    @_target_qpos.setter
    def _target_qpos(self, v):
        self._track.target = v

    def reset(self):
        super().reset()
        self._step = 0
        self._track.capture(self.qpos, None if self.scene._reset_mask_all else self.scene._reset_idx)

    def set_drive_targets(self, targets):
        self.articulation.set_joint_drive_targets(targets, self.joints, self.active_joint_indices)

    def set_action(self, action):
        action = self._preprocess_action(action)
        c, track = self.config, self._track
        self._step = 0
        # (`qpos` is a view of the simulation buffer: a copy is only needed when the start is used again in later substeps)
        track.start = self.qpos.clone() if c.interpolate else self.qpos
        track.target = TARGET_RULES[(bool(c.use_delta), bool(c.use_target))](action, track.start, track.target)
        if c.interpolate:
            self._step_size = (track.target - track.start) / self._sim_steps
        else:
            self.set_drive_targets(track.target)

    def before_simulation_step(self):
        self._step += 1
        if self.config.interpolate:
            self.set_drive_targets(self._track.start + self._step_size * self._step)
            self.articulation.px.gpu_apply_articulation_target_position()

    # ---- native action map / state ------------------------------------------------------------------------------------
    def fused_action_spec(self):
        """rows for the fused native action kernel, or None when this controller keeps state across steps (use_target)
        or updates targets per substep (interpolate)"""
        if self.config.use_target or self.config.interpolate:
            return None
        return fused_joint_columns(self.active_joint_indices, self.single_action_space.shape[0], getattr(self, "action_space_low", None), getattr(self, "action_space_high", None),
                                   self._normalize_action, 1 if self.config.use_delta else 0)

    def get_state(self) -> dict:
        return {"target_qpos": self._track.target} if self.config.use_target else {}

    def set_state(self, state: dict):
        if self.config.use_target:
            self._track.target = state["target_qpos"]


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
    """one action column drives every joint of the group (the Panda's two fingers)"""

    def _get_joint_limits(self):
        lim = super()._get_joint_limits()
        assert np.allclose(lim, lim[:1]), "Mimic joints should have the same limit"
        return lim[:1]


class PDJointPosMimicControllerConfig(PDJointPosControllerConfig):
    controller_cls = PDJointPosMimicController
