"""End-effector space controllers (counterpart of mani_skill/agents/controllers/pd_ee_pose.py:23-262):
the action is a (delta) position / pose of the end-effector link in the robot's root frame, turned into
joint position targets by `Kinematics.compute_ik`."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
import torch
from gymnasium import spaces

from maniskill_amd.utils import gym_utils
from maniskill_amd.utils.geometry.rotation_conversions import (
    euler_angles_to_matrix,
    matrix_to_quaternion,
    quaternion_apply,
    quaternion_multiply,
)
from maniskill_amd.utils.structs.pose import Pose

from .base_controller import ControllerConfig
from .pd_joint_pos import PDJointPosController
from .utils.kinematics import Kinematics


class PDEEPosController(PDJointPosController):
    config: "PDEEPosControllerConfig"
    _target_pose = None

    def _check_gpu_sim_works(self):
        assert self.config.frame == "root_translation", "only translation in the root frame is supported for EE control in the GPU sim"

    def _initialize_joints(self):
        self.initial_qpos = None
        super()._initialize_joints()
        self._check_gpu_sim_works()
        self.kinematics = Kinematics(self.config.urdf_path, self.config.ee_link, self.articulation, self.active_joint_indices)
        self.ee_link = self.kinematics.end_link

    def _initialize_action_space(self):
        lo, hi = (np.broadcast_to(b, 3).astype(np.float32) for b in (self.config.pos_lower, self.config.pos_upper))
        self.single_action_space = spaces.Box(lo, hi, dtype=np.float32)

    def fused_action_spec(self):
        """the delta controllers (`pd_ee_delta_pos`, `pd_ee_delta_pose`) are one pseudo-inverse step of the
        commanded translation (and rotation vector): the native action map has an end-effector block for
        exactly that (include/mssim.h `set_ee_action_map`). Returns {"ee": (link index, rows, low, high,
        rot_scale, flags), "dofs": [...]}; every other configuration keeps IK in torch between the action and
        the joint targets."""
        cfg = self.config
        pose = type(cfg) is PDEEPoseControllerConfig
        if not (pose or type(cfg) is PDEEPosControllerConfig) or not cfg.use_delta or cfg.use_target or cfg.interpolate:
            return None
        if cfg.frame != ("root_translation:root_aligned_body_rotation" if pose else "root_translation"):
            return None
        lo, hi = np.broadcast_to(cfg.pos_lower, 3), np.broadcast_to(cfg.pos_upper, 3)
        if self._normalize_action and not (np.all(lo == lo[0]) and np.all(hi == hi[0])):
            return None  # the native block takes one [low, high] for the three axes
        rot_scale = 0.0
        if pose:
            rl = np.broadcast_to(cfg.rot_lower, 3)
            if not np.all(rl == rl[0]):
                return None
            rot_scale = float(rl[0])
        if set(self.kinematics.active_ancestor_joint_idxs) != set(int(i) for i in self.active_joint_indices.tolist()):
            return None  # the step moves every joint on the link's path: they must all be this controller's
        flags = 2 if self._normalize_action else 0
        return dict(ee=(self.kinematics.end_link_idx, 6 if pose else 3, float(lo[0]), float(hi[0]), rot_scale, flags),
                    dofs=[int(i) for i in self.active_joint_indices.tolist()])

    # ---- where the end effector is --------------------------------------------------------------------
    @property
    def ee_pose(self):
        return self.ee_link.pose

    @property
    def ee_pos(self):
        return self.ee_pose.p

    @property
    def ee_pose_at_base(self):
        """pose of the end-effector link in the robot's root frame"""
        return self.articulation.pose.inv() * self.ee_pose

    def reset(self):
        super().reset()
        here = self.ee_pose_at_base.raw_pose
        if self._target_pose is not None and not self.scene._reset_mask_all:
            rows = self.scene._reset_idx
            self._target_pose.raw_pose[rows] = here[rows]
        else:
            self._target_pose = Pose.create(here.clone())

    # ---- action -> target pose ------------------------------------------------------------------------
    # translation-only control: how a commanded offset combines with the previous pose, per frame name
    _OFFSET_RULES = {
        "root_translation": lambda prev, offset: offset * prev,   # offset expressed in the root frame
        "body_translation": lambda prev, offset: prev * offset,   # offset expressed in the end effector's own frame
    }

    def compute_target_pose(self, prev_ee_pose_at_base, action):
        frame = self.config.frame
        commanded = Pose.create_from_pq(p=action)
        if not self.config.use_delta:
            assert frame == "root_translation", frame
            return commanded
        if frame not in self._OFFSET_RULES:
            raise NotImplementedError(frame)
        return self._OFFSET_RULES[frame](prev_ee_pose_at_base, commanded)

    def set_action(self, action):
        cfg = self.config
        action = self._preprocess_action(action)
        self._step, self._start_qpos = 0, self.qpos.clone()
        reference = self._target_pose if cfg.use_target else self.ee_pose_at_base
        self._target_pose = self.compute_target_pose(reference, action)
        solved = self.kinematics.compute_ik(
            self._target_pose,
            self.articulation.get_qpos(),
            pos_only=type(cfg) == PDEEPosControllerConfig,
            action=action,
            use_delta_ik_solver=cfg.use_delta and not cfg.use_target,
        )
        self._target_qpos = self._start_qpos if solved is None else solved
        if not cfg.interpolate:
            self.set_drive_targets(self._target_qpos)
        else:
            self._step_size = (self._target_qpos - self._start_qpos) / self._sim_steps

    # ---- state (only target-tracking modes carry one) ---------------------------------------------------
    def get_state(self) -> dict:
        if not self.config.use_target:
            return {}
        return dict(target_pose=self._target_pose.raw_pose)

    def set_state(self, state: dict):
        if not self.config.use_target:
            return
        raw = state["target_pose"]
        self._target_pose = Pose.create_from_pq(raw[:, :3], raw[:, 3:])


@dataclass
class PDEEPosControllerConfig(ControllerConfig):
    pos_lower: Union[float, Sequence[float]] = None
    pos_upper: Union[float, Sequence[float]] = None
    stiffness: Union[float, Sequence[float]] = None
    damping: Union[float, Sequence[float]] = None
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    ee_link: str = None
    urdf_path: str = None
    frame: str = "root_translation"
    use_delta: bool = True
    use_target: bool = False
    interpolate: bool = False
    normalize_action: bool = True
    drive_mode: str = "force"
    controller_cls = PDEEPosController


class PDEEPoseController(PDEEPosController):
    """6-dof action: translation (3) and a rotation given as XYZ Euler angles (3). The frame name
    "<translation frame>:<rotation frame>" says in which frame each part of a delta is applied."""

    config: "PDEEPoseControllerConfig"

    def _check_gpu_sim_works(self):
        assert self.config.frame == "root_translation:root_aligned_body_rotation", (
            "only root-frame translation with root-aligned rotation is supported for EE control in the GPU sim"
        )

    def _initialize_action_space(self):
        cfg = self.config
        bounds = [np.concatenate([np.broadcast_to(p, 3), np.broadcast_to(r, 3)]).astype(np.float32)
                  for p, r in ((cfg.pos_lower, cfg.rot_lower), (cfg.pos_upper, cfg.rot_upper))]
        self.single_action_space = spaces.Box(bounds[0], bounds[1], dtype=np.float32)

    def _clip_and_scale_action(self, action):
        """translation: per-axis clip + affine map; rotation: the 3-vector is limited to unit length as a whole, then
        scaled by the rotation bound (reference behaviour: pd_ee_pose.py:197-210)"""
        lin = gym_utils.clip_and_scale_action(action[:, :3], self.action_space_low[:3], self.action_space_high[:3])
        ang = action[:, 3:]
        length = torch.linalg.norm(ang, dim=1, keepdim=True)
        ang = torch.where(length > 1, ang / length.clamp_min(1e-12), ang)
        return torch.cat([lin, ang * self.config.rot_lower], dim=1)

    @staticmethod
    def _euler_quat(angles):
        return matrix_to_quaternion(euler_angles_to_matrix(angles, "XYZ"))

    def compute_target_pose(self, prev_ee_pose_at_base, action):
        frame = self.config.frame
        lin, rot = action[:, :3], self._euler_quat(action[:, 3:6])
        if not self.config.use_delta:
            assert frame == "root_translation:root_aligned_body_rotation", frame
            return Pose.create_from_pq(lin, rot)
        p0, q0 = prev_ee_pose_at_base.p, prev_ee_pose_at_base.q
        # rotation: pre-multiplied = about the root's axes, post-multiplied = about the end effector's own axes
        q1 = quaternion_multiply(rot, q0) if "root_aligned_body_rotation" in frame else quaternion_multiply(q0, rot)
        # translation: along the root's axes, or along the end effector's (previous) axes
        p1 = p0 + (lin if "root_translation" in frame else quaternion_apply(q0, lin))
        return Pose.create_from_pq(p1, q1)


@dataclass
class PDEEPoseControllerConfig(PDEEPosControllerConfig):
    rot_lower: Union[float, Sequence[float]] = None
    rot_upper: Union[float, Sequence[float]] = None
    frame: str = "root_translation:root_aligned_body_rotation"
    controller_cls = PDEEPoseController
