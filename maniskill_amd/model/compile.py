"""Scene model compiler: builder records -> flat constant tables (`mssim_model_desc`).

The reference builds one `sapien.Entity` per sub-scene in O(N) Python loops
(mani_skill/utils/building/actor_builder.py:191-260, articulation_builder.py:113-212) and lets
PhysX cook shapes and inertias. Here every env shares ONE compiled model: fixed joints are folded
into their moving ancestor for dynamics (all link frames are kept for pose output), convex
meshes are cooked to <=64-vertex hulls, and the broadphase filter (collision-group words, SRDF
`disable_collisions`, adjacent links, static-static) is evaluated once into a candidate
shape-pair table.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import geom, mesh
from .urdf import RobotDescription

# enums mirrored from include/mssim.h
JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1
SHAPE_PLANE, SHAPE_BOX, SHAPE_SPHERE, SHAPE_CAPSULE, SHAPE_CYLINDER, SHAPE_CONVEX, SHAPE_NONE, SHAPE_TRIMESH = range(8)
BODY_WORLD, BODY_ART, BODY_FREE, BODY_KIN = range(4)
MAX_DOF, MAX_FREE, MAX_HULL_VERTS = 16, 8, 64

_SHAPE_NAMES = {
    "plane": SHAPE_PLANE,
    "box": SHAPE_BOX,
    "sphere": SHAPE_SPHERE,
    "capsule": SHAPE_CAPSULE,
    "cylinder": SHAPE_CYLINDER,
    "convex": SHAPE_CONVEX,
    "none": SHAPE_NONE,  # per-env padding: this env has no shape in the slot (include/mssim.h MSSIM_SHAPE_NONE)
    "trimesh": SHAPE_TRIMESH,  # triangle mesh of a static / kinematic body (vertices + triangles)
}


@dataclass
class ShapeRecord:
    """One collision shape in its owner's (actor / link) frame."""

    type: str  # plane | box | sphere | capsule | cylinder | convex | none (no shape in this env's slot)
    pose: np.ndarray = field(default_factory=geom.pose)
    half_size: Optional[np.ndarray] = None  # box
    radius: float = 0.0
    half_length: float = 0.0
    vertices: Optional[np.ndarray] = None  # convex / trimesh, shape frame
    triangles: Optional[np.ndarray] = None  # trimesh: [T, 3] vertex indices
    static_friction: float = 0.3
    dynamic_friction: float = 0.3
    restitution: float = 0.0
    patch_radius: float = 0.0
    min_patch_radius: float = 0.0
    density: float = 1000.0
    collision_groups: Sequence[int] = (1, 1, 0, 0)

    def param(self):
        if self.type == "box":
            return np.array([*self.half_size, 0.0])
        if self.type == "sphere":
            return np.array([self.radius, 0, 0, 0.0])
        if self.type in ("capsule", "cylinder"):
            return np.array([self.radius, self.half_length, 0, 0.0])
        return np.zeros(4)

    def bound(self):
        """bounding sphere (centre in the shape frame, radius); planes are unbounded (r<0)."""
        if self.type == "plane":
            return np.zeros(3), -1.0
        if self.type == "box":
            return np.zeros(3), float(np.linalg.norm(self.half_size))
        if self.type == "sphere":
            return np.zeros(3), float(self.radius)
        if self.type == "capsule":
            return np.zeros(3), float(self.radius + self.half_length)
        if self.type == "cylinder":
            return np.zeros(3), float(np.hypot(self.radius, self.half_length))
        if self.type == "none":
            return np.zeros(3), 0.0
        if self.type == "trimesh":
            used = np.asarray(self.vertices, dtype=np.float64)[np.unique(np.asarray(self.triangles))]
            return mesh.bounding_sphere(used)
        return mesh.bounding_sphere(self.vertices)

    def mass_properties(self):
        """(mass, com, inertia about com) in the shape frame from density."""
        rho = self.density
        if self.type == "box":
            hx, hy, hz = self.half_size
            m = rho * 8 * hx * hy * hz
            I = np.diag([hy * hy + hz * hz, hx * hx + hz * hz, hx * hx + hy * hy]) * m / 3.0
            return m, np.zeros(3), I
        if self.type == "sphere":
            m = rho * 4.0 / 3.0 * np.pi * self.radius**3
            return m, np.zeros(3), np.eye(3) * 0.4 * m * self.radius**2
        if self.type == "cylinder":
            r, h = self.radius, 2 * self.half_length
            m = rho * np.pi * r * r * h
            return m, np.zeros(3), np.diag([0.5 * m * r * r, m * (3 * r * r + h * h) / 12, m * (3 * r * r + h * h) / 12])
        if self.type == "capsule":
            r, h = self.radius, 2 * self.half_length
            mc = rho * np.pi * r * r * h
            ms = rho * 4.0 / 3.0 * np.pi * r**3
            ix = 0.5 * mc * r * r + 0.4 * ms * r * r
            iy = mc * (3 * r * r + h * h) / 12 + ms * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
            return mc + ms, np.zeros(3), np.diag([ix, iy, iy])
        if self.type == "convex":
            vol, com, I = mesh.hull_volume_com_inertia(self.vertices)
            return rho * vol, com, rho * I
        return 0.0, np.zeros(3), np.zeros((3, 3))


@dataclass
class ActorRecord:
    name: str
    body_type: str  # dynamic | kinematic | static
    shapes: List[ShapeRecord]
    initial_pose: np.ndarray = field(default_factory=geom.pose)
    mass: Optional[float] = None  # explicit override (with com / inertia)
    com: Optional[np.ndarray] = None
    inertia: Optional[np.ndarray] = None
    linear_damping: float = 0.0
    angular_damping: float = 0.0
    disable_gravity: bool = False
    # per-env geometry (same shape types in every env, different sizes / local poses): one list of
    # ShapeRecords per env, e.g. PegInsertionSide's pegs and boxes-with-hole built per sub-scene and
    # merged (reference: Actor.merge, utils/structs/actor.py:99-126)
    env_shapes: Optional[List[List[ShapeRecord]]] = None


@dataclass
class ArticulationRecord:
    name: str
    robot: RobotDescription
    initial_pose: np.ndarray = field(default_factory=geom.pose)
    fix_root_link: bool = True
    link_shapes: Dict[str, List[ShapeRecord]] = field(default_factory=dict)
    disable_self_collisions: bool = False
    link_gravity: Dict[str, bool] = field(default_factory=dict)  # default True
    # per active joint (by joint name): stiffness, damping, force_limit, mode
    drives: Dict[str, Tuple[float, float, float, int]] = field(default_factory=dict)
    joint_friction: Dict[str, float] = field(default_factory=dict)
    build_mimic_joints: bool = True


@dataclass
class CompiledModel:
    arrays: Dict[str, np.ndarray]
    scalars: Dict[str, float]
    link_names: List[str]
    joint_names: List[str]  # all joints (one per link, root first: "" for the root)
    active_joint_names: List[str]
    free_names: List[str]
    kin_names: List[str]
    static_names: List[str]
    shape_owner: List[str]
    n_rows: int

    @property
    def n_dof(self):
        return int(self.scalars["n_dof"])

    @property
    def n_link(self):
        return int(self.scalars["n_link"])

    @property
    def n_free(self):
        return int(self.scalars["n_free"])

    @property
    def n_kin(self):
        return int(self.scalars["n_kin"])

    @property
    def n_pair(self):
        return int(self.scalars["n_pair"])

    def row_of(self, name: str) -> int:
        if name in self.link_names:
            return self.link_names.index(name)
        if name in self.free_names:
            return self.n_link + self.free_names.index(name)
        if name in self.kin_names:
            return self.n_link + self.n_free + self.kin_names.index(name)
        return -1


def shapes_from_urdf_link(link, materials=None, link_cfg=None, max_hull_verts=MAX_HULL_VERTS) -> List[ShapeRecord]:
    """URDF <collision> elements -> ShapeRecords (reference: SAPIEN URDF loader; material /
    patch-radius overrides per mani_skill/utils/sapien_utils.py:113-168 `apply_urdf_config`)."""
    out = []
    cfg = link_cfg or {}
    mat = cfg.get("material", None)
    for c in link.collisions:
        if c.type == "box":
            s = ShapeRecord("box", c.pose.copy(), half_size=0.5 * c.size)
        elif c.type == "sphere":
            s = ShapeRecord("sphere", c.pose.copy(), radius=float(c.size[0]))
        elif c.type in ("cylinder", "capsule"):
            # URDF axis is z; shape convention is +x: rotate x -> z
            rot = geom.pose(q=geom.mat_to_quat(np.array([[0, 0, -1.0], [0, 1, 0], [1, 0, 0]])))
            s = ShapeRecord(c.type, geom.compose(c.pose, rot), radius=float(c.size[0]), half_length=0.5 * float(c.size[1]))
        elif c.type == "mesh":
            v = mesh.cook_convex_mesh(c.filename, tuple(c.scale), max_hull_verts)
            s = ShapeRecord("convex", c.pose.copy(), vertices=v)
        else:
            raise NotImplementedError(c.type)
        if mat is not None:
            s.static_friction = float(mat["static_friction"])
            s.dynamic_friction = float(mat["dynamic_friction"])
            s.restitution = float(mat["restitution"])
        if "patch_radius" in cfg:
            s.patch_radius = float(cfg["patch_radius"])
        if "min_patch_radius" in cfg:
            s.min_patch_radius = float(cfg["min_patch_radius"])
        if "density" in cfg:
            s.density = float(cfg["density"])
        out.append(s)
    return out


class SceneModelBuilder:
    """Accumulates one env's worth of records; `compile()` produces the shared tables."""

    def __init__(self):
        self.articulation: Optional[ArticulationRecord] = None
        self.actors: List[ActorRecord] = []

    def set_articulation(self, rec: ArticulationRecord):
        if self.articulation is not None:
            raise NotImplementedError("this core supports one articulation per env")
        self.articulation = rec

    def add_actor(self, rec: ActorRecord):
        if any(a.name == rec.name for a in self.actors):
            raise ValueError(f"duplicate actor name {rec.name}")
        self.actors.append(rec)

    # ------------------------------------------------------------------ #
    def compile(
        self,
        num_envs: int = 1,
        timestep=0.01,
        gravity=(0, 0, -9.81),
        contact_offset=0.02,
        rest_offset=0.0,
        bounce_threshold=2.0,
        position_iterations=15,
        velocity_iterations=1,
        erp=0.2,
        max_depenetration_velocity=1.0,
        sleep_threshold=0.005,
    ) -> CompiledModel:
        A: Dict[str, np.ndarray] = {}
        shapes: List[dict] = []  # compiled shape dicts
        link_names, joint_names, active_joint_names = [], [], []

        # ---------------- articulation ----------------
        art = self.articulation
        link_body, link_frame = [], []
        dof_parent, dof_type, dof_frame, dof_axis, dof_limit, dof_drive, dof_arm = [], [], [], [], [], [], []
        body_inertial, body_gravity = [], []
        tendon_dof, tendon_param = [], []
        link_to_body: Dict[str, int] = {}
        link_rel: Dict[str, np.ndarray] = {}
        adjacent = set()
        srdf_disabled = set()
        if art is not None:
            if not art.fix_root_link:
                raise NotImplementedError("floating-base articulations are not supported yet")
            rb = art.robot
            body_items: Dict[int, list] = {}
            joint_to_dof: Dict[str, int] = {}
            for lname in rb.link_order:
                link = rb.links[lname]
                pj = rb.parent_joint.get(lname)
                if pj is None:
                    link_to_body[lname] = -1
                    link_rel[lname] = geom.pose()
                    joint_names.append("")
                elif pj.type == "fixed":
                    link_to_body[lname] = link_to_body[pj.parent]
                    link_rel[lname] = geom.compose(link_rel[pj.parent], pj.origin)
                    joint_names.append(pj.name)
                else:
                    j = len(dof_parent)
                    joint_to_dof[pj.name] = j
                    dof_parent.append(link_to_body[pj.parent])
                    dof_type.append(JOINT_PRISMATIC if pj.type == "prismatic" else JOINT_REVOLUTE)
                    dof_frame.append(geom.compose(link_rel[pj.parent], pj.origin))
                    dof_axis.append(pj.axis)
                    lo, hi = pj.limit if pj.type != "continuous" else (-np.inf, np.inf)
                    dof_limit.append([lo, hi])
                    # SAPIEN default drive: stiffness 0, damping = URDF joint damping
                    # (articulation_builder.py:91-101); controllers override (pd_joint_pos.py:35-49)
                    drv = art.drives.get(pj.name, (0.0, pj.damping, np.inf, 0))
                    dof_drive.append([drv[0], drv[1], min(drv[2], 3.0e38), float(drv[3])])
                    dof_arm.append(0.0)
                    link_to_body[lname] = j
                    link_rel[lname] = geom.pose()
                    body_gravity.append(1 if art.link_gravity.get(lname, True) else 0)
                    joint_names.append(pj.name)
                    active_joint_names.append(pj.name)
                    adjacent.add((link_to_body[pj.parent], j))
                b = link_to_body[lname]
                link_names.append(lname)
                link_body.append(b)
                link_frame.append(link_rel[lname])
                if link.has_inertial and link.mass > 0:
                    body_items.setdefault(b, []).append(
                        geom.transform_inertial(link_rel[lname], link.mass, link.com, link.inertia)
                    )
                # fixed links inherit the gravity flag of ... each link separately in the
                # reference; a folded body uses the flag of its joint-bearing link.
                for s in art.link_shapes.get(lname, []):
                    shapes.append(
                        dict(
                            rec=s,
                            kind=BODY_ART,
                            index=b,
                            row=len(link_names) - 1,
                            frame=geom.compose(link_rel[lname], s.pose),
                            owner=lname,
                            link=lname,
                        )
                    )
            n_dof = len(dof_parent)
            if n_dof > MAX_DOF:
                raise NotImplementedError(f"n_dof={n_dof} exceeds MSSIM_MAX_DOF={MAX_DOF}")
            for j in range(n_dof):
                m, c, I = geom.combine_inertials(body_items.get(j, []))
                if m <= 0:  # massless moving body: tiny default like SAPIEN's loader
                    m, c, I = 1e-6, np.zeros(3), np.eye(3) * 1e-9
                body_inertial.append([m, *c, *geom.inertia_mat_to_vec(I)])
            for a, b in rb.disabled_pairs:
                srdf_disabled.add((a, b))
                srdf_disabled.add((b, a))
            if art.build_mimic_joints:
                for j in rb.joints:
                    if j.mimic is not None and j.name in joint_to_dof and j.mimic[0] in joint_to_dof:
                        # q_j = multiplier * q_src + offset ; fixed tendon stiffness 1e5
                        # (articulation_builder.py:160-199)
                        tendon_dof.append([joint_to_dof[j.mimic[0]], joint_to_dof[j.name]])
                        tendon_param.append([-j.mimic[1], 1.0, j.mimic[2], 1e5, 0.0])
        n_dof = len(dof_parent)
        n_link = len(link_names)

        # ---------------- actors ----------------
        free_names, kin_names, static_names = [], [], []
        free_inertial, free_damping, free_gravity = [], [], []
        init_free, init_kin = [], []
        free_env_slot, env_free_inertial = [], []
        for a in self.actors:
            if a.env_shapes is not None:
                assert len(a.env_shapes) == num_envs, f"{a.name}: per-env shapes need one entry per env"
                # envs may carry different shape types and different numbers of shapes: every env's list is padded to
                # the longest with "none" records; a slot's type in the shared tables is the first real one found
                width = max(len(es) for es in a.env_shapes)
                a.env_shapes = [list(es) + [ShapeRecord("none") for _ in range(width - len(es))] for es in a.env_shapes]
                assert all(s.type != "plane" for es in a.env_shapes for s in es), f"{a.name}: planes cannot be per-env shapes"
                a.shapes = [next((es[k] for es in a.env_shapes if es[k].type != "none"), a.env_shapes[0][k]) for k in range(width)]
            if a.body_type == "dynamic":
                if a.mass is not None:
                    m, c, I = a.mass, np.zeros(3) if a.com is None else a.com, a.inertia
                else:
                    items = []
                    for s in a.shapes:
                        ms, cs, Is = s.mass_properties()
                        items.append(geom.transform_inertial(s.pose, ms, cs, Is))
                    m, c, I = geom.combine_inertials(items)
                    if m <= 0:
                        m, c, I = 1.0, np.zeros(3), np.eye(3)  # PhysX default for shapeless bodies
                idx = len(free_names)
                free_names.append(a.name)
                if a.env_shapes is not None and a.mass is None:
                    per = []
                    for es in a.env_shapes:
                        items = [geom.transform_inertial(s.pose, *s.mass_properties()) for s in es if s.type != "none"]
                        me, ce, Ie = geom.combine_inertials(items) if items else (0.0, np.zeros(3), np.eye(3))
                        if me <= 0:
                            # no shape in this env: the body does not exist there (mass 0 = never awake, include/mssim.h)
                            me, ce, Ie = 0.0, np.zeros(3), np.eye(3)
                        per.append([me, *ce, *geom.inertia_mat_to_vec(Ie)])
                    free_env_slot.append(len(env_free_inertial))
                    env_free_inertial.append(np.asarray(per, dtype=np.float64).T)  # [10, N]
                else:
                    free_env_slot.append(-1)
                free_inertial.append([m, *c, *geom.inertia_mat_to_vec(I)])
                free_damping.append([a.linear_damping, a.angular_damping])
                free_gravity.append(0 if a.disable_gravity else 1)
                init_free.append(a.initial_pose)
                kind, row = BODY_FREE, n_link + idx
            elif a.body_type == "kinematic":
                idx = len(kin_names)
                kin_names.append(a.name)
                init_kin.append(a.initial_pose)
                kind, row = BODY_KIN, None  # fixed up below
            elif a.body_type == "static":
                idx = len(static_names)
                static_names.append(a.name)
                kind, row = BODY_WORLD, -1
            else:
                raise ValueError(a.body_type)
            for si, s in enumerate(a.shapes):
                frame = s.pose if kind != BODY_WORLD else geom.compose(a.initial_pose, s.pose)
                env = None
                if a.env_shapes is not None:
                    env = [es[si] for es in a.env_shapes]
                shapes.append(dict(rec=s, kind=kind, index=idx if kind != BODY_WORLD else 0, row=row, frame=frame, owner=a.name, link=None,
                                   env=env, world_pose=a.initial_pose if kind == BODY_WORLD else None))
        n_free, n_kin = len(free_names), len(kin_names)
        if n_free > MAX_FREE:
            raise NotImplementedError(f"n_free={n_free} exceeds MSSIM_MAX_FREE={MAX_FREE}")
        for s in shapes:
            if s["kind"] == BODY_KIN:
                s["row"] = n_link + n_free + s["index"]

        # ---------------- shape tables ----------------
        hull_verts = []
        tri_soup, tri_nodes = [], []
        st, sk, si, srow, sframe, sparam, smat, shull, sbound = [], [], [], [], [], [], [], [], []
        shape_env_slot, env_frame, env_param, env_bound = [], [], [], []
        env_hulls = {}  # id(vertex array) -> (first vertex, count): per-env hulls that share a mesh are stored once

        def hull_of(rec):
            """(hull vertices about their mean, that mean in the record's shape frame). The narrowphase takes a convex
            shape's frame origin as a point INSIDE it (the generic query's interior point, the side of a mesh triangle the
            shape is on); an asset's own origin may lie anywhere -- at the base of an object, outside a part of a convex
            decomposition --, so the compiled shape frame is moved to the mean of the hull's vertices."""
            v = np.asarray(rec.vertices, dtype=np.float64)
            if len(v) > MAX_HULL_VERTS:
                v = mesh.simplify_hull(v, MAX_HULL_VERTS)
            ctr = v.mean(axis=0)
            return v - ctr, ctr

        def recentred(frame, ctr):
            return geom.compose(frame, geom.pose(ctr))

        env_meshes = {}  # id(vertex array) -> (first triangle, triangle count, root node): a mesh shared by envs / slots is stored once

        def add_mesh(rec):
            key = (id(rec.vertices), id(rec.triangles))
            if key not in env_meshes:
                soup = mesh.triangle_soup(rec.vertices, rec.triangles)
                nodes = mesh.build_bvh16(soup)
                refs = nodes[:, 96:].view(np.int32)
                n_nodes0, n_tri0 = sum(len(x) for x in tri_nodes), sum(len(x) for x in tri_soup)
                leaf = (refs < 0) & (refs != -2**31)
                refs[refs >= 0] += n_nodes0                # node references: global node index
                refs[leaf] = ~((~refs[leaf]) + n_tri0)     # leaf references: global triangle index
                tri_soup.append(soup)
                tri_nodes.append(nodes)  # (kept as float32 arrays: the references are int32 bit patterns)
                env_meshes[key] = (n_tri0, len(soup), n_nodes0)
            return env_meshes[key]

        for s in shapes:
            if s.get("env") is not None:
                fr, pr, bd = [], [], []
                for r_e in s["env"]:
                    f_e = r_e.pose if s["world_pose"] is None else geom.compose(s["world_pose"], r_e.pose)
                    c_e, rad_e = r_e.bound()
                    fr.append(f_e)
                    if r_e.type == "convex":
                        # a different hull per env (include/mssim.h env_shape_param): first vertex and vertex count
                        key = id(r_e.vertices)
                        if key not in env_hulls:
                            v, ctr = hull_of(r_e)
                            env_hulls[key] = (len(hull_verts), len(v), ctr)
                            hull_verts.extend(v.tolist())
                        ctr = env_hulls[key][2]
                        fr[-1] = recentred(f_e, ctr)
                        c_e = c_e - ctr
                        f_e = fr[-1]
                        pr.append([float(env_hulls[key][0]), float(env_hulls[key][1]), 0.0, float(SHAPE_CONVEX + 1)])
                    elif r_e.type == "trimesh":
                        # this env's mesh: first triangle, triangle count, root node of its BVH (a different mesh per env is fine)
                        assert s["kind"] in (BODY_WORLD, BODY_KIN), f"{s['owner']}: triangle-mesh collision needs a static or kinematic body"
                        pr.append([*map(float, add_mesh(r_e)), float(SHAPE_TRIMESH + 1)])
                    else:
                        pr.append([*r_e.param()[:3], float(_SHAPE_NAMES[r_e.type] + 1)])  # [3]: this env's shape type + 1
                    bd.append([*geom.transform_point(f_e, c_e), rad_e])
                shape_env_slot.append(len(env_frame))
                env_frame.append(np.asarray(fr, dtype=np.float64).T)  # [7, N]
                env_param.append(np.asarray(pr, dtype=np.float64).T)  # [4, N]
                env_bound.append(np.asarray(bd, dtype=np.float64).T)  # [4, N]: centre in the BODY frame, radius
            else:
                shape_env_slot.append(-1)
            r: ShapeRecord = s["rec"]
            st.append(_SHAPE_NAMES[r.type])
            sk.append(s["kind"])
            si.append(s["index"])
            srow.append(s["row"])
            sframe.append(s["frame"])
            sparam.append(r.param() if r.type != "trimesh" else np.zeros(4))  # (trimesh: first triangle / count, filled in below)
            # [3]: torsional patch radius. PhysX scales `patch_radius` with the penetration and never goes below
            # `min_patch_radius`; the reference sets both (0.1, panda.py:24-31), where the minimum rules: the larger one
            smat.append([r.static_friction, r.dynamic_friction, r.restitution, max(r.patch_radius, r.min_patch_radius)])
            if r.type == "convex":
                if s.get("env") is not None:
                    shull.append(list(env_hulls[id(r.vertices)][:2]))  # (the representative's hull; every env reads its own)
                    ctr = env_hulls[id(r.vertices)][2]
                else:
                    v, ctr = hull_of(r)
                    shull.append([len(hull_verts), len(v)])
                    hull_verts.extend(v.tolist())
                sframe[-1] = recentred(sframe[-1], ctr)
            elif r.type == "trimesh":
                # triangle mesh (static / kinematic bodies): its triangles join the soup, its 16-wide BVH the node table;
                # shape_hull = (root node, triangle count) -- include/mssim.h MSSIM_SHAPE_TRIMESH
                assert s["kind"] in (BODY_WORLD, BODY_KIN), f"{s['owner']}: triangle-mesh collision needs a static or kinematic body"
                n_tri0, n_tri, n_nodes0 = add_mesh(r)
                shull.append([n_nodes0, n_tri])
                sparam[-1] = np.array([float(n_tri0), float(n_tri), 0.0, 0.0])  # its triangles: tri_soup[first .. first + count)
            else:
                shull.append([0, 0])
            c, rad = r.bound()
            if r.type == "convex":
                c = c - ctr  # (centre of the bounding sphere in the recentred shape frame)
            sbound.append([*c, rad])

        # ---------------- candidate pairs ----------------
        def moving(s):
            return s["kind"] == BODY_FREE or (s["kind"] == BODY_ART and s["index"] >= 0)

        pairs = []
        for ia in range(len(shapes)):
            for ib in range(ia + 1, len(shapes)):
                a, b = shapes[ia], shapes[ib]
                if not (moving(a) or moving(b)):
                    continue
                if a["kind"] == b["kind"] and a["index"] == b["index"] and a["kind"] != BODY_WORLD:
                    continue
                ga, gb = a["rec"].collision_groups, b["rec"].collision_groups
                if ga[2] & gb[2]:
                    continue
                if not ((ga[0] & gb[1]) or (ga[1] & gb[0])):
                    continue
                if a["kind"] == BODY_ART and b["kind"] == BODY_ART:
                    if art.disable_self_collisions:
                        continue
                    ba, bb = a["index"], b["index"]
                    if (ba, bb) in adjacent or (bb, ba) in adjacent:
                        continue
                    if (a["link"], b["link"]) in srdf_disabled:
                        continue
                    # links joined directly by a (fixed) joint never collide
                    pja = art.robot.parent_joint.get(a["link"])
                    pjb = art.robot.parent_joint.get(b["link"])
                    if (pja is not None and pja.parent == b["link"]) or (pjb is not None and pjb.parent == a["link"]):
                        continue
                if st[ia] in (SHAPE_PLANE, SHAPE_TRIMESH) and st[ib] in (SHAPE_PLANE, SHAPE_TRIMESH):
                    continue
                # canonical order: lower shape type first (plane < box < ... < convex)
                if st[ia] <= st[ib]:
                    pairs.append([ia, ib])
                else:
                    pairs.append([ib, ia])

        def arr(x, dtype, shape):
            a = np.asarray(x, dtype=dtype)
            if a.size == 0:
                a = np.zeros(shape, dtype=dtype)
            return np.ascontiguousarray(a.reshape(shape))

        f32, i32 = np.float32, np.int32
        A["dof_parent"] = arr(dof_parent, i32, (n_dof,))
        A["dof_type"] = arr(dof_type, i32, (n_dof,))
        A["dof_frame"] = arr(dof_frame, f32, (n_dof, 7))
        A["dof_axis"] = arr(dof_axis, f32, (n_dof, 3))
        lim = np.asarray(dof_limit, dtype=np.float64).reshape(n_dof, 2)
        A["dof_limit"] = arr(np.clip(lim, -3.0e38, 3.0e38), f32, (n_dof, 2))
        A["dof_drive"] = arr(dof_drive, f32, (n_dof, 4))
        A["dof_armature"] = arr(dof_arm, f32, (n_dof,))
        A["body_inertial"] = arr(body_inertial, f32, (n_dof, 10))
        A["body_gravity"] = arr(body_gravity, i32, (n_dof,))
        A["tendon_dof"] = arr(tendon_dof, i32, (len(tendon_dof), 2))
        A["tendon_param"] = arr(tendon_param, f32, (len(tendon_param), 5))
        A["link_body"] = arr(link_body, i32, (n_link,))
        A["link_frame"] = arr(link_frame, f32, (n_link, 7))
        A["free_inertial"] = arr(free_inertial, f32, (n_free, 10))
        A["free_damping"] = arr(free_damping, f32, (n_free, 2))
        A["free_gravity"] = arr(free_gravity, i32, (n_free,))
        ns = len(shapes)
        A["shape_type"] = arr(st, i32, (ns,))
        A["shape_body_kind"] = arr(sk, i32, (ns,))
        A["shape_body_index"] = arr(si, i32, (ns,))
        A["shape_row"] = arr(srow, i32, (ns,))
        A["shape_frame"] = arr(sframe, f32, (ns, 7))
        A["shape_param"] = arr(sparam, f32, (ns, 4))
        A["shape_material"] = arr(smat, f32, (ns, 4))
        A["shape_hull"] = arr(shull, i32, (ns, 2))
        A["shape_bound"] = arr(sbound, f32, (ns, 4))
        A["hull_verts"] = arr(hull_verts, f32, (len(hull_verts), 3))
        A["tri_soup"] = np.ascontiguousarray(np.concatenate(tri_soup)) if tri_soup else np.zeros((0, 12), dtype=f32)
        A["tri_bvh"] = np.ascontiguousarray(np.concatenate(tri_nodes)) if tri_nodes else np.zeros((0, 112), dtype=f32)
        A["pair_shape"] = arr(pairs, i32, (len(pairs), 2))
        # per-env overrides ([items][N], env fastest -- the layout the kernels read directly)
        n_es, n_ef = len(env_frame), len(env_free_inertial)
        A["shape_env_slot"] = arr(shape_env_slot, i32, (ns,))
        A["env_shape_frame"] = arr(np.concatenate(env_frame) if n_es else [], f32, (7 * n_es, num_envs))
        A["env_shape_param"] = arr(np.concatenate(env_param) if n_es else [], f32, (4 * n_es, num_envs))
        A["env_shape_bound"] = arr(np.concatenate(env_bound) if n_es else [], f32, (4 * n_es, num_envs))
        A["free_env_slot"] = arr(free_env_slot, i32, (n_free,))
        A["env_free_inertial"] = arr(np.concatenate(env_free_inertial) if n_ef else [], f32, (10 * n_ef, num_envs))
        # initial state (not part of the C model; used by the python system at gpu_init)
        A["init_root_pose"] = arr(art.initial_pose if art is not None else geom.pose(), f32, (7,))
        A["init_free_pose"] = arr(init_free, f32, (n_free, 7))
        A["init_kin_pose"] = arr(init_kin, f32, (n_kin, 7))

        scalars = dict(
            n_dof=n_dof,
            n_tendon=len(tendon_dof),
            n_link=n_link,
            n_free=n_free,
            n_kin=n_kin,
            n_shape=ns,
            n_hull_verts=len(hull_verts),
            n_tri=int(sum(len(x) for x in tri_soup)),
            n_tri_node=int(sum(len(x) for x in tri_nodes)),
            n_pair=len(pairs),
            n_env_shape=n_es,
            n_env_free=n_ef,
            num_envs=int(num_envs),
            gravity=tuple(float(g) for g in gravity),
            timestep=float(timestep),
            contact_offset=float(contact_offset),
            rest_offset=float(rest_offset),
            bounce_threshold=float(bounce_threshold),
            position_iterations=int(position_iterations),
            velocity_iterations=int(velocity_iterations),
            erp=float(erp),
            max_depenetration_velocity=float(max_depenetration_velocity),
            sleep_threshold=float(sleep_threshold),
        )
        return CompiledModel(
            arrays=A,
            scalars=scalars,
            link_names=link_names,
            joint_names=joint_names,
            active_joint_names=active_joint_names,
            free_names=free_names,
            kin_names=kin_names,
            static_names=static_names,
            shape_owner=[s["owner"] for s in shapes],
            n_rows=n_link + n_free + n_kin,
        )
