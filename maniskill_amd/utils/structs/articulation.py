"""Batched articulation + joints over `px.cuda_articulation_*` (counterpart of
mani_skill/utils/structs/articulation.py:23-764 and articulation_joint.py:20-347)."""
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from maniskill_amd.utils import common
from maniskill_amd.utils.structs.articulation_joint import ArticulationJoint  # noqa: F401
from maniskill_amd.utils.structs.link import Link
from maniskill_amd.utils.structs.pose import Pose


class Articulation:
    def __init__(self, scene, name: str, record, initial_pose: Pose):
        self.scene = scene
        self.name = name
        self._record = record  # model.compile.ArticulationRecord
        self.initial_pose = initial_pose
        self.links: List[Link] = []
        self.links_map: Dict[str, Link] = {}
        self.joints: List[ArticulationJoint] = []
        self.joints_map: Dict[str, ArticulationJoint] = {}
        self.active_joints: List[ArticulationJoint] = []
        self.active_joints_map: Dict[str, ArticulationJoint] = {}
        self.merged = False
        self._cached_joint_target_indices = {}
        rb = record.robot
        for i, lname in enumerate(rb.link_order):
            link = Link(scene, self, lname, i)
            self.links.append(link)
            self.links_map[lname] = link
        for i, lname in enumerate(rb.link_order):
            pj = rb.parent_joint.get(lname)
            if pj is None:
                j = ArticulationJoint(self, "", "fixed" if record.fix_root_link else "undefined", i, None, None, child_link=self.links[i])
            else:
                active = pj.type != "fixed"
                aidx = len(self.active_joints) if active else None
                lim = None if not active else (pj.limit if pj.type != "continuous" else (-np.inf, np.inf))
                jt = {"continuous": "revolute_unwrapped"}.get(pj.type, pj.type)
                j = ArticulationJoint(self, pj.name, jt, i, aidx, lim, child_link=self.links[i], parent_link=self.links_map[pj.parent])
                if active:
                    self.active_joints.append(j)
                    self.active_joints_map[pj.name] = j
                    j._drive = [0.0, pj.damping, float("inf"), "force"]
            self.joints.append(j)
            self.joints_map[j.name] = j
            self.links[i].joint = j
        self.root = self.links[0]
        self.max_dof = len(self.active_joints)
        self.dof = torch.full((scene.num_envs,), self.max_dof, dtype=torch.int)
        self.fixed_root_link = torch.full((scene.num_envs,), bool(record.fix_root_link))

    # ------------------------------------------------------------------ basics
    @property
    def device(self):
        return self.scene.device

    @property
    def px(self):
        return self.scene.px

    @property
    def _num_objs(self):
        return self.scene.num_envs

    @property
    def _scene_idxs(self):
        return self.scene._all_env_idx

    @property
    def _data_index(self):
        return self.scene._all_env_idx

    def get_links(self):
        return self.links

    def get_joints(self):
        return self.joints

    def get_active_joints(self):
        return self.active_joints

    def get_name(self):
        return self.name

    def get_dof(self):
        return self.dof

    def find_link_by_name(self, name):
        return self.links_map.get(name)

    def find_joint_by_name(self, name):
        return self.joints_map.get(name)

    def __hash__(self):
        return hash(("Articulation", self.name, id(self.scene)))

    def __repr__(self):
        return f"<Articulation {self.name}>"

    # collision groups are a build-time property of the shapes -----------------------------------
    def _set_link_collision_group_bit(self, link_name, group, bit_idx, bit):
        assert not self.scene._gpu_sim_initialized, "collision groups must be set before the simulation is initialised"
        for s in self._record.link_shapes.get(link_name, []):
            g = list(s.collision_groups)
            g[group] = (g[group] & ~(1 << bit_idx)) | (int(bool(bit)) << bit_idx)
            s.collision_groups = tuple(g)

    def _set_link_collision_group(self, link_name, group, value):
        assert not self.scene._gpu_sim_initialized, "collision groups must be set before the simulation is initialised"
        for s in self._record.link_shapes.get(link_name, []):
            g = list(s.collision_groups)
            g[group] = int(value)
            s.collision_groups = tuple(g)

    def _set_drive(self, joint: ArticulationJoint):
        k, d, f, mode = joint._drive
        self._record.drives[joint.name] = (k, d, f, 1 if mode == "acceleration" else 0)
        if self.scene._gpu_sim_initialized:
            drive = self.px.model.arrays["dof_drive"].copy()
            drive[joint.active_index_int] = [k, d, min(f, 3.0e38), 1.0 if mode == "acceleration" else 0.0]
            self.px.set_drive_properties(drive)

    # ------------------------------------------------------------------ state
    def get_state(self):
        pose = self.root.pose
        return torch.hstack([pose.p, pose.q, self.root.linear_velocity, self.root.angular_velocity, self.qpos, self.qvel])

    def set_state(self, state, env_idx: torch.Tensor = None):
        state = common.to_tensor(state, device=self.device)
        with self.scene._narrow_reset_mask(env_idx):
            self.set_root_pose(Pose.create(state[:, :7]))
            self.set_root_linear_velocity(state[:, 7:10])
            self.set_root_angular_velocity(state[:, 10:13])
            self.set_qpos(state[:, 13 : 13 + self.max_dof])
            self.set_qvel(state[:, 13 + self.max_dof :])

    def _masked_write(self, buf: torch.Tensor, value):
        value = common.to_tensor(value, device=self.device)
        if self.scene._reset_mask_all:
            buf[:, : self.max_dof] = value
        else:
            buf[self.scene._reset_idx, : self.max_dof] = value

    @property
    def qpos(self):
        return self.px.cuda_articulation_qpos.torch()[:, : self.max_dof]

    @qpos.setter
    def qpos(self, v):
        self._masked_write(self.px.cuda_articulation_qpos.torch(), v)

    @property
    def qvel(self):
        return self.px.cuda_articulation_qvel.torch()[:, : self.max_dof]

    @qvel.setter
    def qvel(self, v):
        self._masked_write(self.px.cuda_articulation_qvel.torch(), v)

    @property
    def qacc(self):
        return self.px.cuda_articulation_qacc.torch()[:, : self.max_dof]

    @property
    def qf(self):
        return self.px.cuda_articulation_qf.torch()[:, : self.max_dof]

    @qf.setter
    def qf(self, v):
        self._masked_write(self.px.cuda_articulation_qf.torch(), v)

    @property
    def qlimits(self):
        lims = np.array([j._limits for j in self.active_joints], dtype=np.float32).reshape(self.max_dof, 2)
        return torch.from_numpy(lims).to(self.device)[None].repeat(self.scene.num_envs, 1, 1)

    def get_qpos(self):
        return self.qpos

    def get_qvel(self):
        return self.qvel

    def get_qacc(self):
        return self.qacc

    def get_qf(self):
        return self.qf

    def get_qlimits(self):
        return self.qlimits

    def set_qpos(self, v):
        self.qpos = v

    def set_qvel(self, v):
        self.qvel = v

    def set_qf(self, v):
        self.qf = v

    # root -----------------------------------------------------------------------------
    @property
    def pose(self) -> Pose:
        return self.root.pose

    @pose.setter
    def pose(self, v):
        self.root.pose = v

    @property
    def root_pose(self):
        return self.root.pose

    @root_pose.setter
    def root_pose(self, v):
        self.root.pose = v

    def get_pose(self):
        return self.pose

    def set_pose(self, v):
        self.pose = v

    def get_root_pose(self):
        return self.root_pose

    def set_root_pose(self, v):
        self.root_pose = v

    @property
    def root_linear_velocity(self):
        return self.root.linear_velocity

    @property
    def root_angular_velocity(self):
        return self.root.angular_velocity

    def get_root_linear_velocity(self):
        return self.root.linear_velocity

    def get_root_angular_velocity(self):
        return self.root.angular_velocity

    def set_root_linear_velocity(self, v):
        # fixed base: the value is stored in the buffer (state round trip) but has no dynamics effect
        self.root._masked_write(slice(7, 10), v)

    def set_root_angular_velocity(self, v):
        self.root._masked_write(slice(10, 13), v)

    # drive targets (articulation.py:720-764) ------------------------------------------
    def _joint_cols(self, joint_indices):
        key = id(joint_indices)
        hit = self._cached_joint_target_indices.get(key)
        if hit is None or hit[0] is not joint_indices:
            idx = common.to_tensor(joint_indices, device=self.device).long()
            first, last = int(idx[0]), int(idx[-1])
            contiguous = len(idx) == last - first + 1 and bool((idx == torch.arange(first, last + 1, device=idx.device)).all())
            hit = (joint_indices, idx, slice(first, last + 1) if contiguous else None)
            self._cached_joint_target_indices[key] = hit
        return hit[1], hit[2]

    def set_joint_drive_targets(self, targets, joints: List[ArticulationJoint] = None, joint_indices: torch.Tensor = None):
        targets = common.to_tensor(targets, device=self.device)
        idx, sl = self._joint_cols(joint_indices)
        buf = self.px.cuda_articulation_target_qpos.torch()
        if sl is not None:
            buf[:, sl] = targets
        else:
            buf[:, idx] = targets

    def set_joint_drive_velocity_targets(self, targets, joints: List[ArticulationJoint] = None, joint_indices: torch.Tensor = None):
        targets = common.to_tensor(targets, device=self.device)
        idx, sl = self._joint_cols(joint_indices)
        buf = self.px.cuda_articulation_target_qvel.torch()
        if sl is not None:
            buf[:, sl] = targets
        else:
            buf[:, idx] = targets

    # contacts (articulation.py:371-427) -----------------------------------------------
    def get_net_contact_impulses(self, link_names: Union[List[str], Sequence[str]]):
        rows = [self.links_map[n]._body_row for n in link_names]
        q = self.scene._body_query(tuple(rows))
        self.px.gpu_query_contact_body_impulses(q)
        return q.cuda_impulses.torch().clone().reshape(len(rows), self.scene.num_envs, 3).transpose(1, 0)

    def get_net_contact_forces(self, link_names):
        return self.get_net_contact_impulses(link_names) / self.scene.timestep
