"""ArticulationBuilder: wraps a parsed URDF and registers one batched articulation record
(counterpart of mani_skill/utils/building/articulation_builder.py:24-212; mimic joints become
stiff tendon rows as at :160-199)."""
from typing import Optional

import numpy as np

from maniskill_amd.model import geom
from maniskill_amd.model.compile import ArticulationRecord
from maniskill_amd.utils import common
from maniskill_amd.utils.structs.articulation import Articulation
from maniskill_amd.utils.structs.pose import Pose


class LinkBuilderView:
    """per-link handle exposing the fields loaders tweak before build (collision_groups, name)"""

    def __init__(self, record: ArticulationRecord, name: str):
        self._record, self.name = record, name

    @property
    def collision_groups(self):
        shapes = self._record.link_shapes.get(self.name, [])
        return list(shapes[0].collision_groups) if shapes else [1, 1, 0, 0]

    @collision_groups.setter
    def collision_groups(self, g):
        for s in self._record.link_shapes.get(self.name, []):
            s.collision_groups = tuple(g)


class ArticulationBuilder:
    def __init__(self, scene, record: ArticulationRecord):
        self.scene = scene
        self.record = record
        self.name: Optional[str] = record.name
        self.initial_pose = None
        self.scene_idxs = None
        self.disable_self_collisions = False
        self.link_builders = [LinkBuilderView(record, n) for n in record.robot.link_order]

    def set_name(self, name):
        self.name = name
        return self

    def set_initial_pose(self, pose):
        self.initial_pose = pose
        return self

    def set_scene_idxs(self, scene_idxs=None):
        if scene_idxs is not None and len(scene_idxs) != self.scene.num_envs:
            raise NotImplementedError("per-env articulation subsets (build_separate / merge) are not supported by this core yet")
        self.scene_idxs = scene_idxs
        return self

    def build(self, name=None, fix_root_link=None, build_mimic_joints=True) -> Articulation:
        if name is not None:
            self.set_name(name)
        assert self.name is not None and self.name != "" and self.name not in self.scene.articulations, (
            "built articulations in ManiSkill must have unique names and cannot be None or empty strings"
        )
        if fix_root_link is not None:
            self.record.fix_root_link = bool(fix_root_link)
        self.record.name = self.name
        self.record.build_mimic_joints = build_mimic_joints
        self.record.disable_self_collisions = self.record.disable_self_collisions or self.disable_self_collisions
        init = Pose.create(self.initial_pose if self.initial_pose is not None else Pose.create_from_pq(), device=self.scene.device)
        raw = common.to_numpy(init.raw_pose)
        self.record.initial_pose = geom.pose(raw[0, :3], raw[0, 3:])
        art = Articulation(self.scene, self.name, self.record, init)
        self.scene._register_articulation(art, self.record)
        return art
