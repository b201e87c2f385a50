"""Fetch mobile manipulator (counterpart of mani_skill/agents/robots/fetch/fetch.py:20-423): a 7-joint arm with a
parallel gripper on a torso lift, a pan / tilt head, and a planar base modelled as three root joints (x, y, yaw) of the
same articulation -- 15 velocity components, which is what one env's 16 lanes hold.

Same uid, URDF, finger material, rest keyframe, joint groups and control-mode names as the reference. The control modes
are generated: every mode is {arm controller of that name, gripper, body, base}, where only the arm part differs.
"""
from copy import deepcopy

import numpy as np
import sapien
import torch

from maniskill_amd import PACKAGE_ASSET_DIR
from maniskill_amd.agents.base_agent import BaseAgent, Keyframe
from maniskill_amd.agents.controllers import (
    PDBaseForwardVelControllerConfig,
    PDEEPosControllerConfig,
    PDEEPoseControllerConfig,
    PDJointPosControllerConfig,
    PDJointPosMimicControllerConfig,
    PDJointPosVelControllerConfig,
    PDJointVelControllerConfig,
    deepcopy_dict,
)
from maniskill_amd.agents.registration import register_agent
from maniskill_amd.utils import common
from maniskill_amd.utils.structs.pose import Pose

# group-2 collision bits: shapes that share a set bit do not collide (the ground of a scene sets them so that the base,
# whose height is fixed by the root joints, does not scrub on it -- fetch.py:20-23, envs/tasks/empty_env.py:41)
FETCH_WHEELS_COLLISION_BIT = 30
FETCH_BASE_COLLISION_BIT = 31

_FINGER = dict(material="gripper", patch_radius=0.1, min_patch_radius=0.1)


@register_agent()
class Fetch(BaseAgent):
    uid = "fetch"
    urdf_path = f"{PACKAGE_ASSET_DIR}/robots/fetch/fetch.urdf"
    urdf_config = dict(
        _materials=dict(gripper=dict(static_friction=2.0, dynamic_friction=2.0, restitution=0.0)),
        link=dict(r_gripper_finger_link=dict(_FINGER), l_gripper_finger_link=dict(_FINGER)),
    )
    keyframes = dict(
        rest=Keyframe(pose=sapien.Pose(), qpos=np.array([0, 0, 0, 0.386, 0, -0.370, 0.562, -1.032, 0.695, 0.955, -0.1, 2.077, 0, 0.015, 0.015]))
    )

    base_joint_names = ["root_x_axis_joint", "root_y_axis_joint", "root_z_rotation_joint"]
    body_joint_names = ["head_pan_joint", "head_tilt_joint", "torso_lift_joint"]
    arm_joint_names = ["shoulder_pan_joint", "shoulder_lift_joint", "upperarm_roll_joint", "elbow_flex_joint", "forearm_roll_joint", "wrist_flex_joint", "wrist_roll_joint"]
    gripper_joint_names = ["l_gripper_finger_joint", "r_gripper_finger_joint"]
    ee_link_name = "gripper_link"

    # stiffness, damping, force limit of the arm / gripper / body drives
    arm_gains = (1e3, 1e2, 100)
    gripper_gains = (1e3, 1e2, 100)
    body_gains = (1e3, 1e2, 100)

    @property
    def _controller_configs(self):
        J = self.arm_joint_names
        k, d, f = self.arm_gains
        ee = dict(joint_names=J, stiffness=k, damping=d, force_limit=f, ee_link=self.ee_link_name, urdf_path=self.urdf_path)

        def with_target(cfg):
            cfg = deepcopy(cfg)
            cfg.use_target = True
            return cfg

        arm = dict(
            pd_joint_delta_pos=PDJointPosControllerConfig(J, -0.1, 0.1, k, d, f, use_delta=True),
            pd_joint_pos=PDJointPosControllerConfig(J, None, None, k, d, f, normalize_action=False),
            pd_ee_delta_pos=PDEEPosControllerConfig(pos_lower=-0.1, pos_upper=0.1, **ee),
            pd_ee_delta_pose=PDEEPoseControllerConfig(pos_lower=-0.1, pos_upper=0.1, rot_lower=-0.1, rot_upper=0.1, **ee),
            pd_joint_vel=PDJointVelControllerConfig(J, -1.0, 1.0, d, f),
            pd_joint_pos_vel=PDJointPosVelControllerConfig(J, None, None, k, d, f, normalize_action=True),
            pd_joint_delta_pos_vel=PDJointPosVelControllerConfig(J, -0.1, 0.1, k, d, f, use_delta=True),
        )
        for mode in ("pd_joint_delta_pos", "pd_ee_delta_pos", "pd_ee_delta_pose"):
            arm[mode.replace("delta", "target_delta")] = with_target(arm[mode])
        # (a thin object still gets squeezed: the lower target lies inside the closed position, fetch.py:205-214)
        gripper = PDJointPosMimicControllerConfig(self.gripper_joint_names, -0.01, 0.05, *self.gripper_gains)
        body = PDJointPosControllerConfig(self.body_joint_names, -0.1, 0.1, *self.body_gains, use_delta=True)
        stiff_body = PDJointPosControllerConfig(self.body_joint_names, None, None, 1e5, 1e5, 1e5, normalize_action=False)
        base = PDBaseForwardVelControllerConfig(self.base_joint_names, lower=[-1, -3.14], upper=[1, 3.14], damping=1000, force_limit=500)
        modes = {name: dict(arm=cfg, gripper=gripper, body=body, base=base) for name, cfg in arm.items()}
        modes["pd_joint_delta_pos_stiff_body"] = dict(arm=arm["pd_joint_delta_pos"], gripper=gripper, body=stiff_body, base=base)
        return deepcopy_dict(modes)

    def _after_init(self):
        links = self.robot.links_map
        self.finger1_link, self.finger2_link = links["l_gripper_finger_link"], links["r_gripper_finger_link"]
        self.tcp = links[self.ee_link_name]
        self.base_link, self.torso_lift_link, self.head_camera_link = links["base_link"], links["torso_lift_link"], links["head_camera_link"]
        self.l_wheel_link, self.r_wheel_link = links["l_wheel_link"], links["r_wheel_link"]
        for wheel in (self.l_wheel_link, self.r_wheel_link):
            wheel.set_collision_group_bit(group=2, bit_idx=FETCH_WHEELS_COLLISION_BIT, bit=1)
        self.base_link.set_collision_group_bit(group=2, bit_idx=FETCH_BASE_COLLISION_BIT, bit=1)

    def is_grasping(self, object, min_force=0.5, max_angle=85):
        """both fingers press `object` with at least min_force N, each within max_angle of its own closing direction"""
        flags = []
        for finger, sign in ((self.finger1_link, -1.0), (self.finger2_link, 1.0)):
            force = self.scene.get_pairwise_contact_forces(finger, object)
            closing = sign * finger.pose.to_transformation_matrix()[..., :3, 1]
            angle = torch.rad2deg(common.compute_angle_between(closing, force))
            flags.append((torch.linalg.norm(force, dim=1) >= min_force) & (angle <= max_angle))
        return flags[0] & flags[1]

    def is_static(self, threshold: float = 0.2, base_threshold: float = 0.05):
        qvel = self.robot.get_qvel()
        return torch.all(qvel[..., 3:-2] <= threshold, dim=1) & torch.all(qvel[..., :3] <= base_threshold, dim=1)

    @staticmethod
    def build_grasp_pose(approaching, closing, center):
        """pose of the gripper frame whose z axis approaches and whose y axis closes (unit, orthogonal vectors)"""
        approaching, closing = np.asarray(approaching, dtype=float), np.asarray(closing, dtype=float)
        for v in (approaching, closing):
            assert abs(1 - np.linalg.norm(v)) < 1e-3
        assert abs(approaching @ closing) <= 1e-3
        T = np.eye(4)
        T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = np.cross(closing, approaching), closing, approaching, center
        return sapien.Pose(T)

    @property
    def tcp_pose(self) -> Pose:
        a, b = self.finger1_link.pose, self.finger2_link.pose
        return Pose.create_from_pq(p=(a.p + b.p) / 2, q=(a.q + b.q) / 2)
