"""Franka Panda agent (counterpart of mani_skill/agents/robots/panda/panda.py:16-298): same uid,
URDF, gripper material, controller table, grasp / static tests."""
from copy import deepcopy

import numpy as np
import sapien
import torch

from maniskill_amd import PACKAGE_ASSET_DIR
from maniskill_amd.agents.base_agent import BaseAgent, Keyframe
from maniskill_amd.agents.controllers import *  # noqa: F401,F403
from maniskill_amd.agents.controllers import deepcopy_dict
from maniskill_amd.agents.registration import register_agent
from maniskill_amd.utils import common, sapien_utils


@register_agent()
class Panda(BaseAgent):
    uid = "panda"
    urdf_path = f"{PACKAGE_ASSET_DIR}/robots/panda/panda_v2.urdf"
    urdf_config = dict(
        _materials=dict(gripper=dict(static_friction=2.0, dynamic_friction=2.0, restitution=0.0)),
        link=dict(
            panda_leftfinger=dict(material="gripper", patch_radius=0.1, min_patch_radius=0.1),
            panda_rightfinger=dict(material="gripper", patch_radius=0.1, min_patch_radius=0.1),
        ),
    )
    keyframes = dict(
        rest=Keyframe(qpos=np.array([0.0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, -np.pi / 4, 0.04, 0.04]), pose=sapien.Pose())
    )
    arm_joint_names = [f"panda_joint{i}" for i in range(1, 8)]
    gripper_joint_names = ["panda_finger_joint1", "panda_finger_joint2"]
    ee_link_name = "panda_hand_tcp"

    arm_stiffness = 1e3
    arm_damping = 1e2
    arm_force_limit = 100
    gripper_stiffness = 1e3
    gripper_damping = 1e2
    gripper_force_limit = 100

    @property
    def _controller_configs(self):
        J, k, d, f = self.arm_joint_names, self.arm_stiffness, self.arm_damping, self.arm_force_limit
        arm_pd_joint_pos = PDJointPosControllerConfig(J, lower=None, upper=None, stiffness=k, damping=d, force_limit=f, normalize_action=False)
        arm_pd_joint_delta_pos = PDJointPosControllerConfig(J, lower=-0.1, upper=0.1, stiffness=k, damping=d, force_limit=f, use_delta=True)
        arm_pd_joint_target_delta_pos = deepcopy(arm_pd_joint_delta_pos)
        arm_pd_joint_target_delta_pos.use_target = True
        ee = dict(stiffness=k, damping=d, force_limit=f, ee_link=self.ee_link_name, urdf_path=self.urdf_path)
        arm_pd_ee_delta_pos = PDEEPosControllerConfig(joint_names=J, pos_lower=-0.1, pos_upper=0.1, **ee)
        arm_pd_ee_delta_pose = PDEEPoseControllerConfig(joint_names=J, pos_lower=-0.1, pos_upper=0.1, rot_lower=-0.1, rot_upper=0.1, **ee)
        arm_pd_ee_pose = PDEEPoseControllerConfig(joint_names=J, pos_lower=None, pos_upper=None, use_delta=False, normalize_action=False, **ee)
        arm_pd_ee_target_delta_pos = deepcopy(arm_pd_ee_delta_pos)
        arm_pd_ee_target_delta_pos.use_target = True
        arm_pd_ee_target_delta_pose = deepcopy(arm_pd_ee_delta_pose)
        arm_pd_ee_target_delta_pose.use_target = True
        arm_pd_joint_vel = PDJointVelControllerConfig(J, -1.0, 1.0, d, f)
        arm_pd_joint_pos_vel = PDJointPosVelControllerConfig(J, None, None, k, d, f, normalize_action=False)
        arm_pd_joint_delta_pos_vel = PDJointPosVelControllerConfig(J, -0.1, 0.1, k, d, f, use_delta=True)
        # lower = -0.01: "a trick to have force when the object is thin" (panda.py:177-184)
        gripper_pd_joint_pos = PDJointPosMimicControllerConfig(
            self.gripper_joint_names, lower=-0.01, upper=0.04, stiffness=self.gripper_stiffness, damping=self.gripper_damping,
            force_limit=self.gripper_force_limit,
        )
        g = gripper_pd_joint_pos
        controller_configs = dict(
            pd_joint_delta_pos=dict(arm=arm_pd_joint_delta_pos, gripper=g),
            pd_joint_pos=dict(arm=arm_pd_joint_pos, gripper=g),
            pd_ee_delta_pos=dict(arm=arm_pd_ee_delta_pos, gripper=g),
            pd_ee_delta_pose=dict(arm=arm_pd_ee_delta_pose, gripper=g),
            pd_ee_pose=dict(arm=arm_pd_ee_pose, gripper=g),
            pd_joint_target_delta_pos=dict(arm=arm_pd_joint_target_delta_pos, gripper=g),
            pd_ee_target_delta_pos=dict(arm=arm_pd_ee_target_delta_pos, gripper=g),
            pd_ee_target_delta_pose=dict(arm=arm_pd_ee_target_delta_pose, gripper=g),
            pd_joint_vel=dict(arm=arm_pd_joint_vel, gripper=g),
            pd_joint_pos_vel=dict(arm=arm_pd_joint_pos_vel, gripper=g),
            pd_joint_delta_pos_vel=dict(arm=arm_pd_joint_delta_pos_vel, gripper=g),
        )
        return deepcopy_dict(controller_configs)

    def _after_init(self):
        links = self.robot.get_links()
        self.finger1_link = sapien_utils.get_obj_by_name(links, "panda_leftfinger")
        self.finger2_link = sapien_utils.get_obj_by_name(links, "panda_rightfinger")
        self.finger1pad_link = sapien_utils.get_obj_by_name(links, "panda_leftfinger_pad")
        self.finger2pad_link = sapien_utils.get_obj_by_name(links, "panda_rightfinger_pad")
        self.tcp = sapien_utils.get_obj_by_name(links, self.ee_link_name)

    def is_grasping(self, object, min_force=0.5, max_angle=85):
        """both fingers press the object with >= min_force N within max_angle of their closing axis"""
        lf = self.scene.get_pairwise_contact_forces(self.finger1_link, object)
        rf = self.scene.get_pairwise_contact_forces(self.finger2_link, object)
        lforce, rforce = torch.linalg.norm(lf, dim=1), torch.linalg.norm(rf, dim=1)
        ldir = self.finger1_link.pose.to_transformation_matrix()[..., :3, 1]
        rdir = -self.finger2_link.pose.to_transformation_matrix()[..., :3, 1]
        langle = common.compute_angle_between(ldir, lf)
        rangle = common.compute_angle_between(rdir, rf)
        lflag = torch.logical_and(lforce >= min_force, torch.rad2deg(langle) <= max_angle)
        rflag = torch.logical_and(rforce >= min_force, torch.rad2deg(rangle) <= max_angle)
        return torch.logical_and(lflag, rflag)

    def is_static(self, threshold: float = 0.2):
        qvel = self.robot.get_qvel()[..., :-2]
        return torch.max(torch.abs(qvel), 1)[0] <= threshold

    @staticmethod
    def build_grasp_pose(approaching, closing, center):
        assert np.abs(1 - np.linalg.norm(approaching)) < 1e-3
        assert np.abs(1 - np.linalg.norm(closing)) < 1e-3
        assert np.abs(approaching @ closing) <= 1e-3
        ortho = np.cross(closing, approaching)
        T = np.eye(4)
        T[:3, :3] = np.stack([ortho, closing, approaching], axis=1)
        T[:3, 3] = center
        return sapien.Pose(T)
