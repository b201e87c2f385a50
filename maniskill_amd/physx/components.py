"""Builder-time component surface of `sapien.physx`: the objects the reference's builders create and
hand to PhysX -- `ActorBuilder.build_physx_component` (mani_skill/utils/building/actor_builder.py:57-163) makes a
`PhysxRigidDynamicComponent` / `PhysxRigidStaticComponent`, attaches `PhysxCollisionShape*` objects carrying a
`PhysxMaterial`, sets mass properties; `ArticulationBuilder` (articulation_builder.py:65-212) does the same with
`PhysxArticulationLinkComponent`.

Here a component is the builder-time DESCRIPTION of one batched body: attaching shapes and setting fields records what
the model compiler (maniskill_amd/model/compile.py) turns into the constant tables of `mssim_model_desc`
(include/mssim.h). `to_record()` is that hand-over; `ActorBuilder.build` goes through it, so a body assembled by hand
from these classes compiles to the same model as one assembled with the builder's `add_*_collision` calls. After
`gpu_init` the scene fills in `gpu_pose_index` / `gpu_index` (row of env 0 in `cuda_rigid_body_data`; the rows of the
other envs follow at stride 1: body-major layout, include/mssim.h).

Not available (raise on construction): triangle-mesh shapes, multi-convex loading.
"""
from typing import List, Optional, Sequence

import numpy as np

from maniskill_amd.model import geom, mesh
from maniskill_amd.model.compile import ActorRecord, ShapeRecord


class PhysxMaterial:
    """static / dynamic friction and restitution of a collision shape (sapien.physx.PhysxMaterial)"""

    def __init__(self, static_friction: float = 0.3, dynamic_friction: float = 0.3, restitution: float = 0.0):
        self.static_friction, self.dynamic_friction, self.restitution = float(static_friction), float(dynamic_friction), float(restitution)

    def get_static_friction(self):
        return self.static_friction

    def get_dynamic_friction(self):
        return self.dynamic_friction

    def get_restitution(self):
        return self.restitution

    def __repr__(self):
        return f"PhysxMaterial({self.static_friction}, {self.dynamic_friction}, {self.restitution})"


def _pose7(pose) -> np.ndarray:
    if pose is None:
        return geom.pose()
    if isinstance(pose, np.ndarray):
        return geom.pose(pose[:3], pose[3:])
    raw = getattr(pose, "raw_pose", None)
    if raw is not None:  # batched Pose struct
        raw = raw.detach().cpu().numpy()
        assert raw.shape[0] == 1, "shape poses must be unbatched"
        return geom.pose(raw[0, :3], raw[0, 3:])
    return geom.pose(np.asarray(pose.p, dtype=np.float64), np.asarray(pose.q, dtype=np.float64))


class PhysxCollisionShape:
    """common part of every shape: local pose, material, density, collision groups, torsional patch radii"""

    _type = None

    def __init__(self, material: Optional[PhysxMaterial] = None):
        self.physical_material = material if material is not None else PhysxMaterial()
        self.local_pose = None
        self.density = 1000.0
        self.patch_radius = 0.0
        self.min_patch_radius = 0.0
        self.collision_groups = [1, 1, 0, 0]
        self.contact_offset = None  # (scene-wide values are used: types.py:40-41)
        self.rest_offset = None

    # sapien setters / getters -------------------------------------------------------------------
    def set_local_pose(self, pose):
        self.local_pose = pose

    def get_local_pose(self):
        return self.local_pose

    def set_collision_groups(self, groups: Sequence[int]):
        assert len(groups) == 4
        self.collision_groups = [int(g) for g in groups]

    def get_collision_groups(self):
        return list(self.collision_groups)

    def set_density(self, density: float):
        self.density = float(density)

    def set_patch_radius(self, r: float):
        self.patch_radius = float(r)

    def set_min_patch_radius(self, r: float):
        self.min_patch_radius = float(r)

    def set_physical_material(self, material: PhysxMaterial):
        self.physical_material = material

    def get_physical_material(self):
        return self.physical_material

    # hand-over to the model compiler ----------------------------------------------------------------
    def _geometry(self) -> dict:
        return {}

    def to_record(self) -> ShapeRecord:
        m = self.physical_material
        return ShapeRecord(
            self._type,
            _pose7(self.local_pose),
            static_friction=float(m.static_friction),
            dynamic_friction=float(m.dynamic_friction),
            restitution=float(m.restitution),
            patch_radius=self.patch_radius,
            min_patch_radius=self.min_patch_radius,
            density=self.density,
            collision_groups=tuple(self.collision_groups),
            **self._geometry(),
        )


class PhysxCollisionShapePlane(PhysxCollisionShape):
    """half space, normal = +x of the shape frame"""

    _type = "plane"

    def __init__(self, material: Optional[PhysxMaterial] = None):
        super().__init__(material)
        self.density = 0.0


class PhysxCollisionShapeBox(PhysxCollisionShape):
    _type = "box"

    def __init__(self, half_size, material: Optional[PhysxMaterial] = None):
        super().__init__(material)
        self.half_size = np.asarray(half_size, dtype=np.float64).reshape(3)

    def _geometry(self):
        return dict(half_size=self.half_size)


class PhysxCollisionShapeSphere(PhysxCollisionShape):
    _type = "sphere"

    def __init__(self, radius: float, material: Optional[PhysxMaterial] = None):
        super().__init__(material)
        self.radius = float(radius)

    def _geometry(self):
        return dict(radius=self.radius)


class PhysxCollisionShapeCapsule(PhysxCollisionShape):
    """axis = +x of the shape frame"""

    _type = "capsule"

    def __init__(self, radius: float, half_length: float, material: Optional[PhysxMaterial] = None):
        super().__init__(material)
        self.radius, self.half_length = float(radius), float(half_length)

    def _geometry(self):
        return dict(radius=self.radius, half_length=self.half_length)


class PhysxCollisionShapeCylinder(PhysxCollisionShapeCapsule):
    _type = "cylinder"


class PhysxCollisionShapeConvexMesh(PhysxCollisionShape):
    """convex hull of a mesh file (cooked to <= 64 vertices, MSSIM_MAX_HULL_VERTS) or of explicit vertices"""

    _type = "convex"

    def __init__(self, filename: Optional[str] = None, scale=(1.0, 1.0, 1.0), material: Optional[PhysxMaterial] = None, vertices=None):
        super().__init__(material)
        self.filename, self.scale = filename, tuple(float(s) for s in scale)
        self.vertices = np.asarray(vertices, dtype=np.float64) if vertices is not None else mesh.cook_convex_mesh(str(filename), self.scale)

    def _geometry(self):
        return dict(vertices=self.vertices)

    @staticmethod
    def load_multiple(filename, scale=(1.0, 1.0, 1.0), material=None):
        """one convex shape per part of a convex-decomposition file (sapien: PhysxCollisionShapeConvexMesh.load_multiple,
        used by actor_builder.py:121-135): OBJ files with one `o` / `g` group per part"""
        return [PhysxCollisionShapeConvexMesh(filename, scale, material, vertices=v) for v in mesh.cook_convex_parts(str(filename), tuple(float(x) for x in scale))]


class PhysxCollisionShapeTriangleMesh(PhysxCollisionShape):
    """triangle mesh of a static or kinematic body (the reference's nonconvex collision, actor_builder.py:136-150): the
    triangles of an OBJ / STL file, or explicit vertices + triangles"""

    _type = "trimesh"

    def __init__(self, filename: Optional[str] = None, scale=(1.0, 1.0, 1.0), material: Optional[PhysxMaterial] = None, vertices=None, triangles=None):
        super().__init__(material)
        self.filename, self.scale = filename, tuple(float(s) for s in scale)
        self.density = 0.0
        if vertices is None:
            vertices, triangles = mesh.load_triangles(str(filename))
        self.vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3) * np.asarray(self.scale)
        self.triangles = np.asarray(triangles, dtype=np.int64).reshape(-1, 3)

    def _geometry(self):
        return dict(vertices=self.vertices, triangles=self.triangles)


class PhysxBaseComponent:
    def __init__(self):
        self.entity = None
        self.name: Optional[str] = None
        self.collision_shapes: List[PhysxCollisionShape] = []
        self.gpu_pose_index: Optional[int] = None  # row of env 0 in cuda_rigid_body_data (set at gpu_init)

    def attach(self, shape: PhysxCollisionShape):
        self.collision_shapes.append(shape)
        return self

    def get_collision_shapes(self):
        return list(self.collision_shapes)


class PhysxRigidStaticComponent(PhysxBaseComponent):
    body_type = "static"

    def to_record(self, name: str, initial_pose=None) -> ActorRecord:
        return ActorRecord(name, "static", [s.to_record() for s in self.collision_shapes], initial_pose=_pose7(initial_pose))


class PhysxRigidDynamicComponent(PhysxBaseComponent):
    """a dynamic body, or a kinematic one (`kinematic = True`: moved by the user, infinite mass)"""

    def __init__(self):
        super().__init__()
        self.kinematic = False
        self.gpu_index: Optional[int] = None
        self.mass: Optional[float] = None  # None = from the shapes' densities
        self.inertia = None  # principal moments (3) or full 3x3, about the centre of mass
        self.cmass_local_pose = None
        self.linear_damping = 0.0
        self.angular_damping = 0.0
        self.disable_gravity = False
        self._locked = [False] * 6

    @property
    def body_type(self):
        return "kinematic" if self.kinematic else "dynamic"

    def set_locked_motion_axes(self, axes: Sequence[bool]):
        assert len(axes) == 6
        if any(axes):
            raise NotImplementedError("locked motion axes are not supported by this core (free bodies have 6 velocity components)")
        self._locked = [bool(a) for a in axes]

    def get_locked_motion_axes(self):
        return list(self._locked)

    def set_mass(self, m):
        self.mass = float(m)

    def set_kinematic(self, k: bool):
        self.kinematic = bool(k)

    def to_record(self, name: str, initial_pose=None) -> ActorRecord:
        rec = ActorRecord(
            name,
            self.body_type,
            [s.to_record() for s in self.collision_shapes],
            initial_pose=_pose7(initial_pose),
            linear_damping=float(self.linear_damping),
            angular_damping=float(self.angular_damping),
            disable_gravity=bool(self.disable_gravity),
        )
        if self.mass is not None and not self.kinematic:
            rec.mass = float(self.mass)
            rec.com = None if self.cmass_local_pose is None else np.asarray(_pose7(self.cmass_local_pose)[:3], dtype=np.float64)
            I = None if self.inertia is None else np.asarray(self.inertia, dtype=np.float64)
            rec.inertia = np.diag(I) if I is not None and I.ndim == 1 else I
        return rec


class PhysxArticulationLinkComponent(PhysxBaseComponent):
    """one link of the articulation. The builders of this package create articulations from a URDF description
    (model/urdf.py) rather than link by link; after `gpu_init` the scene hands these out as the read-only view the
    reference's struct layer uses: `index`, `is_root`, `parent`, `joint`, `articulation`, `gpu_pose_index`."""

    def __init__(self, parent: Optional["PhysxArticulationLinkComponent"] = None):
        super().__init__()
        self.parent = parent
        self.index: Optional[int] = None
        self.joint = None
        self.articulation = None

    @property
    def is_root(self) -> bool:
        return self.parent is None


class PhysxContactPoint:
    def __init__(self, impulse, normal=None, position=None, separation=None):
        self.impulse, self.normal, self.position, self.separation = impulse, normal, position, separation


class PhysxContact:
    """one entry of `px.get_contacts()`: the two bodies and the contact impulse of the LAST substep between them. The
    core exports impulses per shape pair (include/mssim.h query_pair_impulses), not per point: `points` holds one
    aggregate point per shape pair with the summed impulse on `bodies[0]`."""

    def __init__(self, body0, body1, points):
        self.bodies = [body0, body1]
        self.components = self.bodies
        self.points = points
