"""Articulation links as batched rigid bodies (API counterpart of mani_skill/utils/structs/link.py:27-340)."""
import torch

from maniskill_amd.utils.structs.base import _RigidBase
from maniskill_amd.utils.structs.pose import Pose, vectorize_pose


class Link(_RigidBase):
    """articulation link (link.py:27-340)"""

    def __init__(self, scene, articulation, name: str, index: int, joint=None):
        self.scene = scene
        self.articulation = articulation
        self.name = name
        self.index_int = index
        self._body_row = index
        self.joint = joint
        self.merged = False
        self.disable_gravity_flag = False

    @property
    def index(self) -> torch.Tensor:
        return torch.full((self.scene.num_envs,), self.index_int, dtype=torch.int, device=self.device)

    @property
    def is_root(self) -> torch.Tensor:
        return torch.full((self.scene.num_envs,), self.index_int == 0, dtype=torch.bool, device=self.device)

    def get_index(self):
        return self.index

    def get_joint(self):
        return self.joint

    def get_articulation(self):
        return self.articulation

    def get_name(self):
        return self.name

    @property
    def disable_gravity(self):
        return torch.full((self.scene.num_envs,), self.disable_gravity_flag, dtype=torch.bool, device=self.device)

    @disable_gravity.setter
    def disable_gravity(self, v: bool):
        if self.scene._gpu_sim_initialized:
            raise AssertionError("disable_gravity cannot be changed after gpu_init (structs/decorators.py:1-13)")
        self.disable_gravity_flag = bool(v)
        self.articulation._record.link_gravity[self.name] = not bool(v)

    def set_collision_group_bit(self, group: int, bit_idx: int, bit):
        self.articulation._set_link_collision_group_bit(self.name, group, bit_idx, bit)

    def set_collision_group(self, group: int, value):
        self.articulation._set_link_collision_group(self.name, group, value)

    @property
    def pose(self) -> Pose:
        return Pose.create(self._rows()[:, :7])

    @pose.setter
    def pose(self, arg1) -> None:
        """only meaningful for the root link (articulation root pose, link.py:239-269)"""
        raw = vectorize_pose(arg1, device=self.device)
        if not self.scene._gpu_sim_initialized:
            self.articulation.initial_pose = Pose.create(raw)
            return
        self._masked_write(slice(0, 7), raw)

    def set_pose(self, arg1) -> None:
        self.pose = arg1
