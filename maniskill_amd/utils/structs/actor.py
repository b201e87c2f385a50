"""Batched rigid actors as views over `px.cuda_rigid_body_data`.

API counterpart of mani_skill/utils/structs/actor.py (shared row logic: structs/base.py, links: structs/link.py). The reference keeps one
`sapien.Entity` per sub-scene and gathers rows by `_body_data_index`; here one record describes
the body in all envs and its rows are one contiguous slice (body-major rows), so getters are
zero-copy views. Setters keep the reference's partial-reset contract: only rows selected by
`scene._reset_mask` are written (actor.py:378-380, link.py:255-257, base.py:369-374, 442-447).
Velocity columns are lin 7:10 / ang 10:13 for all rows (documented deviation).
"""
from typing import List, Optional, Union

import numpy as np
import torch

from maniskill_amd.utils import common
from maniskill_amd.utils.structs.pose import Pose, vectorize_pose
from maniskill_amd.utils.structs.base import _RigidBase


class Actor(_RigidBase):
    """dynamic / kinematic / static actor built by `ActorBuilder` (actor.py:25-400)"""

    def __init__(self, scene, name: str, body_type: str, initial_pose: Pose, has_collision_shapes: bool, mass: float = 0.0):
        self.scene = scene
        self.name = name
        self.px_body_type = body_type
        self.initial_pose = initial_pose
        self.has_collision_shapes = has_collision_shapes
        self._mass = mass
        self.hidden = False
        self.before_hide_pose = None
        self._body_row = None
        self.merged = False
        self._fragment = None
        self._own_idx = None

    @classmethod
    def merge(cls, actors: List["Actor"], name: str = None) -> "Actor":
        """Combine per-env fragments (built with `set_scene_idxs([...])`) into one batched actor with per-env geometry.
        Counterpart of the reference's merged views (utils/structs/actor.py:99-126): the fragments may differ in shape
        types and shape counts; envs that no fragment covers do not contain the object (the merged actor is then a batched
        object over the covered envs only). An env may carry at most one of the fragments."""
        from maniskill_amd.model import geom
        from maniskill_amd.model.compile import ActorRecord

        scene = actors[0].scene
        N = scene.num_envs
        by_env = {}
        for a in actors:
            assert a._fragment is not None, f"{a.name} is not a per-env fragment (build it with set_scene_idxs)"
            for i in a._fragment["scene_idxs"]:
                assert i not in by_env, f"env {i} has more than one fragment"
                by_env[i] = a
        covered = sorted(by_env)
        first = by_env[covered[0]]
        assert all(by_env[i].px_body_type == first.px_body_type for i in covered)
        # (a kinematic body is a row of the state in every env; a static one is geometry only, a dynamic one has mass 0 where it is absent)
        assert first.px_body_type != "kinematic" or covered == list(range(N)), "a kinematic object has to exist in every env"
        if first.px_body_type == "static":  # (scenery has no row to carry a pose of its own: one world pose for the merged actor)
            poses = torch.cat([by_env[i].initial_pose.raw_pose[:1] for i in covered], dim=0)
            assert torch.allclose(poses, poses[:1].expand_as(poses)), "static fragments merge only when they are built at the same pose"
        name = name if name is not None else first.name
        # (an env without the object still has the body's row: it keeps this pose, has mass 0 there and takes part in nothing)
        raw = torch.cat([by_env.get(i, first).initial_pose.raw_pose[:1] for i in range(N)], dim=0)
        p0 = common.to_numpy(raw[0])
        rec = ActorRecord(
            name, first.px_body_type, list(first._fragment["shapes"]), initial_pose=geom.pose(p0[:3], p0[3:]),
            linear_damping=first._fragment["linear_damping"], angular_damping=first._fragment["angular_damping"],
            env_shapes=[list(by_env[i]._fragment["shapes"]) if i in by_env else [] for i in range(N)],
        )
        masses = [sum(s.mass_properties()[0] for s in by_env[i]._fragment["shapes"]) for i in covered]
        merged = cls(scene, name, first.px_body_type, Pose.create(raw), has_collision_shapes=first.has_collision_shapes, mass=0.0)
        merged._mass_per_env = torch.tensor(masses, dtype=torch.float32)
        if covered != list(range(N)):
            merged._own_idx = torch.tensor(covered, dtype=torch.long, device=scene.device)
            merged.initial_pose = Pose.create(raw[merged._own_idx.to(raw.device)])
        merged.merged = True
        for a in actors:
            scene._fragments.pop(a.name, None)
        scene._register_actor(merged, rec)
        scene.remove_from_state_dict_registry(merged)  # callers add merged actors explicitly (peg_insertion_side.py:176-181)
        return merged

    # ---- state dict ----------------------------------------------------------
    def get_state(self):
        pose = self.pose
        if self.px_body_type == "dynamic":
            return torch.hstack([pose.p, pose.q, self.linear_velocity, self.angular_velocity])
        z = torch.zeros((len(pose), 6), device=self.device)
        return torch.hstack([pose.p, pose.q, z])

    def set_state(self, state, env_idx: torch.Tensor = None):
        state = common.to_tensor(state, device=self.device)
        with self.scene._narrow_reset_mask(env_idx):
            self.set_pose(Pose.create(state[:, :7]))
            if self.px_body_type == "dynamic":
                self.set_linear_velocity(state[:, 7:10])
                self.set_angular_velocity(state[:, 10:13])

    # ---- visibility (actor.py:176-218) -----------------------------------------
    def hide_visual(self):
        assert not self.has_collision_shapes, "Cannot hide objects with collision shapes in the GPU sim"
        if self.hidden:
            return
        self.before_hide_pose = self._rows()[:, :7].clone()
        self._write_rows(slice(0, 3), self.before_hide_pose[:, :3] + 99999)
        self.scene._gpu_apply_all()
        self.scene._gpu_fetch_all()
        self.hidden = True

    def show_visual(self):
        assert not self.has_collision_shapes, "Cannot show objects with collision shapes in the GPU sim"
        if not self.hidden:
            return
        self.hidden = False
        self._write_rows(slice(0, 7), self.before_hide_pose)
        self.scene._gpu_apply_all()
        self.scene._gpu_fetch_all()

    def is_static(self, lin_thresh=1e-2, ang_thresh=1e-1):
        return torch.logical_and(
            torch.linalg.norm(self.linear_velocity, dim=1) <= lin_thresh,
            torch.linalg.norm(self.angular_velocity, dim=1) <= ang_thresh,
        )

    def set_collision_group_bit(self, group: int, bit_idx: int, bit):
        self.scene._set_collision_group_bit(self.name, group, bit_idx, bit)

    def set_collision_group(self, group: int, value):
        self.scene._set_collision_group(self.name, group, value)

    def apply_force(self, force):
        """force for the next simulation step only (actor.py:305-316)"""
        force = common.to_tensor(force, device=self.device)
        N = self.scene.num_envs
        if self._own_idx is None:
            buf = self.px.cuda_rigid_body_force.torch()[self._body_row * N : (self._body_row + 1) * N]
            buf[self.scene._reset_idx, :3] = force
        else:
            idx = self._body_data_index
            if not self.scene._reset_mask_all:
                idx = idx[self.scene._reset_mask[self._scene_idxs]]
            self.px.cuda_rigid_body_force.torch()[idx, :3] = force
        self.px.gpu_apply_rigid_dynamic_force()

    @property
    def mass(self):
        if getattr(self, "_mass_per_env", None) is not None:
            return self._mass_per_env.to(self.device)
        return torch.full((self._num_objs,), float(self._mass), device=self.device)

    def get_mass(self):
        return self.mass

    # ---- pose -------------------------------------------------------------------
    @property
    def pose(self) -> Pose:
        if self.px_body_type == "static":
            return self.initial_pose
        if self.hidden:
            return Pose.create(self.before_hide_pose)
        return Pose.create(self._rows()[:, :7])

    @pose.setter
    def pose(self, arg1) -> None:
        if self.px_body_type == "static":
            if self.scene._gpu_sim_initialized:
                raise AssertionError("cannot set the pose of a static actor after the simulation is initialised")
            self.initial_pose = Pose.create(arg1, device=self.device)
            return
        raw = vectorize_pose(arg1, device=self.device)
        if not self.scene._gpu_sim_initialized:
            self.initial_pose = Pose.create(raw)
            return
        if self.hidden:
            if self.scene._reset_mask_all:
                self.before_hide_pose[:] = raw
            elif self._own_idx is None:
                self.before_hide_pose[self.scene._reset_idx] = raw
            else:
                self.before_hide_pose[self.scene._reset_mask[self._scene_idxs]] = raw
            return
        self._masked_write(slice(0, 7), raw)

    def set_pose(self, arg1) -> None:
        self.pose = arg1

    def set_linear_velocity(self, v):
        self._masked_write(slice(7, 10), v)

    def set_angular_velocity(self, v):
        self._masked_write(slice(10, 13), v)

    @_RigidBase.linear_velocity.setter
    def linear_velocity(self, v):
        self.set_linear_velocity(v)

    @_RigidBase.angular_velocity.setter
    def angular_velocity(self, v):
        self.set_angular_velocity(v)


from maniskill_amd.utils.structs.link import Link  # noqa: E402,F401  (historic import path)
