"""State shared by the batched rigid bodies (actors and articulation links): rows of `px.cuda_rigid_body_data`, the
partial-reset write contract, velocities, net contact forces.

API counterpart of mani_skill/utils/structs/base.py (PhysxRigidBodyComponentStruct, :100-450). The reference keeps one
`sapien.Entity` per sub-scene and gathers rows by `_body_data_index`; here one record describes the body in all envs and
its rows are one contiguous slice (body-major rows), so getters are zero-copy views. Setters keep the reference's
partial-reset contract: only rows selected by `scene._reset_mask` are written (base.py:369-374, 442-447).
Velocity columns are lin 7:10 / ang 10:13 for all rows (documented deviation).
"""
from typing import Optional

import torch

from maniskill_amd.utils import common
from maniskill_amd.utils.structs.pose import Pose


class _RigidBase:
    """state shared by Actor and Link"""

    name: str
    scene = None
    _body_row: Optional[int] = None
    # envs the object exists in (ascending), None = all of them. An object built for a subset of the sub-scenes
    # (`set_scene_idxs`, the reference's per-env object sets) is a batched object over THOSE envs: its getters return
    # `len(_own_idx)` rows and its setters take as many (structs/base.py:103-110, actor.py:378-380)
    _own_idx: Optional[torch.Tensor] = None

    @property
    def device(self):
        return self.scene.device

    @property
    def px(self):
        return self.scene.px

    @property
    def _num_objs(self):
        return self.scene.num_envs if self._own_idx is None else len(self._own_idx)

    @property
    def _scene_idxs(self):
        return self.scene._all_env_idx if self._own_idx is None else self._own_idx.to(self.device)

    @property
    def _body_data_index(self) -> torch.Tensor:
        """row indices into `px.cuda_rigid_body_data` (structs/base.py:103-110)"""
        N = self.scene.num_envs
        return self._body_row * N + self._scene_idxs

    @property
    def _body_data(self) -> torch.Tensor:
        return self.px.cuda_rigid_body_data.torch()

    def _rows(self) -> torch.Tensor:
        """the object's rows: a zero-copy view when it exists in every env, a gathered copy for a subset (write through
        `_write_rows`)"""
        N = self.scene.num_envs
        if self._own_idx is None:
            return self._body_data[self._body_row * N : (self._body_row + 1) * N]
        return self._body_data[self._body_data_index]

    def _write_rows(self, cols: slice, value):
        """all of the object's rows, whatever the reset mask says"""
        if self._own_idx is None:
            self._rows()[:, cols] = value
        else:
            self._body_data[self._body_data_index, cols] = value

    def _masked_write(self, cols: slice, value):
        value = common.to_tensor(value, device=self.device)
        if self._own_idx is None:
            rows = self._rows()
            if self.scene._reset_mask_all:
                rows[:, cols] = value
            else:
                rows[self.scene._reset_idx, cols] = value  # (index list, not the boolean mask: no host sync)
            return
        # subset object: its rows among the envs being reset, in ascending env order (the reference's
        # `_reset_mask[self._scene_idxs]`); `value` has one row per selected object (or broadcasts)
        idx = self._body_data_index
        if not self.scene._reset_mask_all:
            idx = idx[self.scene._reset_mask[self._scene_idxs]]
        self._body_data[idx, cols] = value

    # velocities -------------------------------------------------------------
    @property
    def linear_velocity(self) -> torch.Tensor:
        return self._rows()[:, 7:10]

    @property
    def angular_velocity(self) -> torch.Tensor:
        return self._rows()[:, 10:13]

    def get_linear_velocity(self):
        return self.linear_velocity

    def get_angular_velocity(self):
        return self.angular_velocity

    def get_pose(self) -> Pose:
        return self.pose

    # contact forces -----------------------------------------------------------
    def get_net_contact_impulses(self):
        q = self.scene._body_query(self._body_row)
        self.px.gpu_query_contact_body_impulses(q)
        return q.cuda_impulses.torch().clone()

    def get_net_contact_forces(self):
        return self.get_net_contact_impulses() / self.scene.timestep

    def __hash__(self):
        return hash((type(self).__name__, self.name, id(self.scene)))

    def __repr__(self):
        return f"<{type(self).__name__} {self.name}>"
