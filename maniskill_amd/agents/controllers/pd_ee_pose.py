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
        low = np.float32(np.broadcast_to(self.config.pos_lower, 3))
        high = np.float32(np.broadcast_to(self.config.pos_upper, 3))
        self.single_action_space = spaces.Box(low, high, dtype=np.float32)

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

    @property
    def ee_pos(self):
        return self.ee_link.pose.p

    @property
    def ee_pose(self):
        return self.ee_link.pose

    @property
    def ee_pose_at_base(self):
        return self.articulation.pose.inv() * self.ee_pose

    def reset(self):
        super().reset()
        cur = self.ee_pose_at_base
        if self._target_pose is None or self.scene._reset_mask_all:
            self._target_pose = Pose.create(cur.raw_pose.clone())
        else:
            m = self.scene._reset_idx
            self._target_pose.raw_pose[m] = cur.raw_pose[m]

    def compute_target_pose(self, prev_ee_pose_at_base, action):
        if self.config.use_delta:
            delta_pose = Pose.create_from_pq(p=action)
            if self.config.frame == "root_translation":
                return delta_pose * prev_ee_pose_at_base
            if self.config.frame == "body_translation":
                return prev_ee_pose_at_base * delta_pose
            raise NotImplementedError(self.config.frame)
        assert self.config.frame == "root_translation", self.config.frame
        return Pose.create_from_pq(p=action)

    def set_action(self, action):
        action = self._preprocess_action(action)
        self._step = 0
        self._start_qpos = self.qpos.clone()
        prev = self._target_pose if self.config.use_target else self.ee_pose_at_base
        self._target_pose = self.compute_target_pose(prev, action)
        pos_only = type(self.config) == PDEEPosControllerConfig
        self._target_qpos = self.kinematics.compute_ik(
            self._target_pose, self.articulation.get_qpos(), pos_only=pos_only, action=action,
            use_delta_ik_solver=self.config.use_delta and not self.config.use_target,
        )
        if self._target_qpos is None:
            self._target_qpos = self._start_qpos
        if self.config.interpolate:
            self._step_size = (self._target_qpos - self._start_qpos) / self._sim_steps
        else:
            self.set_drive_targets(self._target_qpos)

    def get_state(self) -> dict:
        return {"target_pose": self._target_pose.raw_pose} if self.config.use_target else {}

    def set_state(self, state: dict):
        if self.config.use_target:
            t = state["target_pose"]
            self._target_pose = Pose.create_from_pq(t[:, :3], t[:, 3:])


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
    config: "PDEEPoseControllerConfig"

    def _check_gpu_sim_works(self):
        assert self.config.frame == "root_translation:root_aligned_body_rotation", (
            "only root-frame translation with root-aligned rotation is supported for EE control in the GPU sim"
        )

    def _initialize_action_space(self):
        low = np.float32(np.hstack([np.broadcast_to(self.config.pos_lower, 3), np.broadcast_to(self.config.rot_lower, 3)]))
        high = np.float32(np.hstack([np.broadcast_to(self.config.pos_upper, 3), np.broadcast_to(self.config.rot_upper, 3)]))
        self.single_action_space = spaces.Box(low, high, dtype=np.float32)

    def _clip_and_scale_action(self, action):
        # translation per axis, rotation clipped by its norm (pd_ee_pose.py:197-210)
        pos = gym_utils.clip_and_scale_action(action[:, :3], self.action_space_low[:3], self.action_space_high[:3])
        rot = action[:, 3:]
        norm = torch.linalg.norm(rot, dim=1, keepdim=True)
        rot = torch.where(norm > 1, rot / norm.clamp_min(1e-12), rot) * self.config.rot_lower
        return torch.hstack([pos, rot])

    def compute_target_pose(self, prev_ee_pose_at_base, action):
        if self.config.use_delta:
            delta_pos, delta_rot = action[:, 0:3], action[:, 3:6]
            delta_quat = matrix_to_quaternion(euler_angles_to_matrix(delta_rot, "XYZ"))
            if "root_aligned_body_rotation" in self.config.frame:
                q = quaternion_multiply(delta_quat, prev_ee_pose_at_base.q)
            else:
                q = quaternion_multiply(prev_ee_pose_at_base.q, delta_quat)
            if "root_translation" in self.config.frame:
                p = prev_ee_pose_at_base.p + delta_pos
            else:
                p = prev_ee_pose_at_base.p + quaternion_apply(prev_ee_pose_at_base.q, delta_pos)
            return Pose.create_from_pq(p, q)
        assert self.config.frame == "root_translation:root_aligned_body_rotation", self.config.frame
        target_pos, target_rot = action[:, 0:3], action[:, 3:6]
        return Pose.create_from_pq(target_pos, matrix_to_quaternion(euler_angles_to_matrix(target_rot, "XYZ")))


@dataclass
class PDEEPoseControllerConfig(PDEEPosControllerConfig):
    rot_lower: Union[float, Sequence[float]] = None
    rot_upper: Union[float, Sequence[float]] = None
    frame: str = "root_translation:root_aligned_body_rotation"
    controller_cls = PDEEPoseController
