"""ActorBuilder: records collision shapes and registers ONE batched actor record with the scene.

Counterpart of mani_skill/utils/building/actor_builder.py:31-260 (which subclasses
`sapien.ActorBuilder` and builds one entity per sub-scene). Visual shapes are accepted and
dropped: this build has no renderer (state observations only).
"""
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from maniskill_amd.model import geom
from maniskill_amd.model.compile import ShapeRecord
from maniskill_amd.physx import components as pxc
from maniskill_amd.physx.components import PhysxMaterial  # noqa: F401  (historic import path)
from maniskill_amd.utils import common
from maniskill_amd.utils.structs.actor import Actor
from maniskill_amd.utils.structs.pose import Pose


class ActorBuilder:
    def __init__(self, scene=None):
        self.scene = scene
        self.name: Optional[str] = None
        self.initial_pose = None
        self.physx_body_type = "dynamic"
        self.collision_groups = [1, 1, 0, 0]
        self.collision_shapes: List[pxc.PhysxCollisionShape] = []  # in the order of the add_*_collision calls
        self.scene_idxs = None
        self._mass = None
        self._cmass_local_pose = None
        self._inertia = None
        self.linear_damping = 0.0
        self.angular_damping = 0.0

    # -- configuration ------------------------------------------------------------
    def set_scene(self, scene):
        self.scene = scene
        return self

    def set_name(self, name: str):
        self.name = name
        return self

    def set_initial_pose(self, pose):
        self.initial_pose = pose
        return self

    def set_physx_body_type(self, t: str):
        assert t in ("dynamic", "kinematic", "static")
        self.physx_body_type = t
        return self

    def set_scene_idxs(self, scene_idxs=None):
        """Restrict the actor to some envs. Built for a proper subset it is a *fragment*: either it is combined with
        the fragments of other envs by `Actor.merge` into one actor with per-env geometry (shape types, sizes, local
        poses, hulls and shape counts may all differ: one object model per sub-scene, as PegInsertionSide
        (peg_insertion_side.py:114-181) and the PickSingleYCB family do), or it stays what it is: an object that exists
        in those envs only, a batched object over them (getters and setters have one row per such env)."""
        self.scene_idxs = None if scene_idxs is None else [int(i) for i in scene_idxs]
        return self

    def set_collision_groups(self, groups):
        self.collision_groups = list(groups)
        return self

    def set_mass_and_inertia(self, mass, cmass_local_pose, inertia):
        self._mass, self._cmass_local_pose, self._inertia = float(mass), cmass_local_pose, np.asarray(inertia, dtype=np.float64)
        return self

    @property
    def shapes(self) -> List[ShapeRecord]:
        """the recorded shapes as model-compiler records"""
        return [s.to_record() for s in self.collision_shapes]

    def _attach(self, shape: pxc.PhysxCollisionShape, pose, density, patch_radius, min_patch_radius):
        """every add_*_collision call ends here: the shape object gets what the reference sets on it just before
        `component.attach(shape)` (actor_builder.py:152-159)"""
        shape.local_pose = pose
        shape.set_density(density)
        shape.set_patch_radius(patch_radius)
        shape.set_min_patch_radius(min_patch_radius)
        self.collision_shapes.append(shape)
        return self

    def _mat(self, material):
        return material if material is not None else self.scene.default_material

    # -- collision shapes (sapien.ActorBuilder names) ------------------------------------
    def add_plane_collision(self, pose=None, material=None, patch_radius=0, min_patch_radius=0):
        return self._attach(pxc.PhysxCollisionShapePlane(self._mat(material)), pose, 0.0, patch_radius, min_patch_radius)

    def add_box_collision(self, pose=None, half_size=(1, 1, 1), material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        return self._attach(pxc.PhysxCollisionShapeBox(half_size, self._mat(material)), pose, density, patch_radius, min_patch_radius)

    def add_sphere_collision(self, pose=None, radius=1, material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        return self._attach(pxc.PhysxCollisionShapeSphere(radius, self._mat(material)), pose, density, patch_radius, min_patch_radius)

    def add_capsule_collision(self, pose=None, radius=1, half_length=1, material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        return self._attach(pxc.PhysxCollisionShapeCapsule(radius, half_length, self._mat(material)), pose, density, patch_radius, min_patch_radius)

    def add_cylinder_collision(self, pose=None, radius=1, half_length=1, material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        return self._attach(pxc.PhysxCollisionShapeCylinder(radius, half_length, self._mat(material)), pose, density, patch_radius, min_patch_radius)

    def add_convex_collision_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        return self._attach(pxc.PhysxCollisionShapeConvexMesh(filename, scale, self._mat(material)), pose, density, patch_radius, min_patch_radius)

    def build_physx_component(self, link_parent=None):
        """the body as a `physx` component with its shapes attached (reference: actor_builder.py:57-163); `build`
        registers exactly what this component describes"""
        if self.physx_body_type == "static":
            comp = pxc.PhysxRigidStaticComponent()
        elif self.physx_body_type in ("dynamic", "kinematic"):
            comp = pxc.PhysxRigidDynamicComponent()
            comp.kinematic = self.physx_body_type == "kinematic"
            comp.linear_damping, comp.angular_damping = self.linear_damping, self.angular_damping
            if self._mass is not None and not comp.kinematic:
                comp.mass, comp.cmass_local_pose, comp.inertia = self._mass, self._cmass_local_pose, self._inertia
        else:
            raise Exception(f"invalid physx body type [{self.physx_body_type}]")
        for shape in self.collision_shapes:
            shape.set_collision_groups(self.collision_groups)
            comp.attach(shape)
        return comp

    def add_multiple_convex_collisions_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, density=1000, patch_radius=0, min_patch_radius=0, is_trigger=False):
        """every convex part of a decomposition file becomes a shape of the body (actor_builder.py:121-135)"""
        for shape in pxc.PhysxCollisionShapeConvexMesh.load_multiple(filename, scale, self._mat(material)):
            self._attach(shape, pose, density, patch_radius, min_patch_radius)
        return self

    def add_nonconvex_collision_from_file(self, filename, pose=None, scale=(1, 1, 1), material=None, patch_radius=0, min_patch_radius=0, is_trigger=False):
        """triangle-mesh collision for static / kinematic bodies (actor_builder.py:136-150)"""
        return self._attach(pxc.PhysxCollisionShapeTriangleMesh(filename, scale, self._mat(material)), pose, 0.0, patch_radius, min_patch_radius)

    # -- visuals: accepted, ignored -------------------------------------------------------
    def add_box_visual(self, *a, **kw):
        return self

    def add_sphere_visual(self, *a, **kw):
        return self

    def add_capsule_visual(self, *a, **kw):
        return self

    def add_cylinder_visual(self, *a, **kw):
        return self

    def add_visual_from_file(self, *a, **kw):
        return self

    # -- build ---------------------------------------------------------------------------
    def build(self, name=None) -> Actor:
        if name is not None:
            self.set_name(name)
        assert self.name is not None and self.name != "" and self.name not in self.scene.actors, (
            "built actors in ManiSkill must have unique names and cannot be None or empty strings"
        )
        comp = self.build_physx_component()
        comp.name = self.name
        init = Pose.create(self.initial_pose if self.initial_pose is not None else Pose.create_from_pq(), device=self.scene.device)
        if self.scene_idxs is not None and len(self.scene_idxs) != self.scene.num_envs:
            # fragment: lives in a subset of envs until merged
            frag = Actor(self.scene, self.name, self.physx_body_type, init, has_collision_shapes=len(self.collision_shapes) > 0)
            frag._fragment = dict(scene_idxs=list(self.scene_idxs), shapes=self.shapes, linear_damping=self.linear_damping,
                                  angular_damping=self.angular_damping)
            self.scene._fragments[self.name] = frag
            return frag
        raw = common.to_numpy(init.raw_pose)
        if raw.shape[0] != 1:
            # per-env initial poses: the model keeps env 0's pose; the rest are written after gpu_init
            pass
        rec = comp.to_record(self.name, geom.pose(raw[0, :3], raw[0, 3:]))
        mass = 0.0
        if self.physx_body_type == "dynamic":
            mass = rec.mass if rec.mass is not None else sum(sr.mass_properties()[0] for sr in rec.shapes)
        actor = Actor(self.scene, self.name, self.physx_body_type, init, has_collision_shapes=len(rec.shapes) > 0, mass=mass)
        actor._px_component = comp
        self.scene._register_actor(actor, rec)
        return actor

    def build_kinematic(self, name=None) -> Actor:
        self.set_physx_body_type("kinematic")
        return self.build(name)

    def build_static(self, name=None) -> Actor:
        self.set_physx_body_type("static")
        return self.build(name)

    def build_dynamic(self, name=None) -> Actor:
        self.set_physx_body_type("dynamic")
        return self.build(name)
