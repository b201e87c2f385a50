"""URDF / SRDF parsing into plain records (host side, stdlib xml only).

The reference delegates URDF parsing to SAPIEN's C++ loader
(mani_skill/utils/building/urdf_loader.py:28-47 -> sapien.wrapper.urdf_loader.URDFLoader);
this module is the in-repo counterpart. Only what the rigid-body hot path needs is kept:
links (inertial + collision shapes), joints (type/origin/axis/limits/dynamics/mimic) and the
SRDF `disable_collisions` pairs.
"""
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import geom


def _floats(s, n=None, default=None):
    if s is None:
        return np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None:
        assert len(v) == n, f"expected {n} floats, got {s!r}"
    return v


def _origin(elem) -> np.ndarray:
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return geom.pose()
    xyz = _floats(o.get("xyz"), 3, [0, 0, 0])
    rpy = _floats(o.get("rpy"), 3, [0, 0, 0])
    return geom.pose(xyz, geom.rpy_to_quat(rpy))


@dataclass
class CollisionRecord:
    type: str  # "box" | "sphere" | "cylinder" | "capsule" | "mesh" | "plane"
    pose: np.ndarray  # in the link frame
    size: np.ndarray = None  # box: full size; sphere: [r]; cylinder/capsule: [r, length]
    filename: Optional[str] = None
    scale: np.ndarray = None


@dataclass
class LinkRecord:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about com, link frame
    has_inertial: bool = False
    collisions: List[CollisionRecord] = field(default_factory=list)


@dataclass
class JointRecord:
    name: str
    type: str  # revolute | prismatic | continuous | fixed
    parent: str
    child: str
    origin: np.ndarray  # joint frame in the parent link frame (== child link frame at q=0)
    axis: np.ndarray
    limit: Tuple[float, float] = (-np.inf, np.inf)
    effort: float = np.inf
    velocity: float = np.inf
    damping: float = 0.0
    friction: float = 0.0
    mimic: Optional[Tuple[str, float, float]] = None  # (joint, multiplier, offset)


@dataclass
class RobotDescription:
    name: str
    links: Dict[str, LinkRecord]
    joints: List[JointRecord]
    root: str
    link_order: List[str]  # parents before children (depth-first, URDF joint order)
    parent_joint: Dict[str, JointRecord]
    disabled_pairs: List[Tuple[str, str]] = field(default_factory=list)
    package_dir: str = ""


def parse_urdf(urdf_file: str, srdf_file: Optional[str] = None) -> RobotDescription:
    tree = ET.parse(urdf_file)
    robot = tree.getroot()
    package_dir = os.path.dirname(os.path.abspath(urdf_file))
    links: Dict[str, LinkRecord] = {}
    for le in robot.findall("link"):
        lr = LinkRecord(name=le.get("name"))
        ie = le.find("inertial")
        if ie is not None:
            T = _origin(ie)
            m = float(ie.find("mass").get("value"))
            it = ie.find("inertia")
            I = np.zeros((3, 3))
            if it is not None:
                g = lambda k: float(it.get(k, 0.0))
                I = np.array(
                    [
                        [g("ixx"), g("ixy"), g("ixz")],
                        [g("ixy"), g("iyy"), g("iyz")],
                        [g("ixz"), g("iyz"), g("izz")],
                    ]
                )
            R = geom.quat_to_mat(T[3:])
            lr.mass, lr.com, lr.inertia, lr.has_inertial = m, T[:3].copy(), R @ I @ R.T, True
        for ce in le.findall("collision"):
            ge = ce.find("geometry")
            if ge is None:
                continue
            T = _origin(ce)
            if ge.find("box") is not None:
                lr.collisions.append(CollisionRecord("box", T, _floats(ge.find("box").get("size"), 3)))
            elif ge.find("sphere") is not None:
                lr.collisions.append(
                    CollisionRecord("sphere", T, np.array([float(ge.find("sphere").get("radius"))]))
                )
            elif ge.find("cylinder") is not None:
                c = ge.find("cylinder")
                lr.collisions.append(
                    CollisionRecord("cylinder", T, np.array([float(c.get("radius")), float(c.get("length"))]))
                )
            elif ge.find("capsule") is not None:
                c = ge.find("capsule")
                lr.collisions.append(
                    CollisionRecord("capsule", T, np.array([float(c.get("radius")), float(c.get("length"))]))
                )
            elif ge.find("mesh") is not None:
                me = ge.find("mesh")
                fn = me.get("filename")
                if fn.startswith("package://"):
                    fn = fn[len("package://"):]
                lr.collisions.append(
                    CollisionRecord(
                        "mesh", T, filename=os.path.join(package_dir, fn), scale=_floats(me.get("scale"), 3, [1, 1, 1])
                    )
                )
        links[lr.name] = lr

    joints: List[JointRecord] = []
    for je in robot.findall("joint"):
        jt = je.get("type")
        if jt not in ("revolute", "prismatic", "continuous", "fixed"):
            raise NotImplementedError(f"joint type {jt!r} of {je.get('name')} is not supported")
        ax = je.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0, 0])
        n = np.linalg.norm(axis)
        axis = axis / n if n > 0 else np.array([1.0, 0, 0])
        jr = JointRecord(
            name=je.get("name"),
            type=jt,
            parent=je.find("parent").get("link"),
            child=je.find("child").get("link"),
            origin=_origin(je),
            axis=axis,
        )
        lim = je.find("limit")
        if lim is not None and jt in ("revolute", "prismatic"):
            jr.limit = (float(lim.get("lower", 0.0)), float(lim.get("upper", 0.0)))
        if lim is not None:
            jr.effort = float(lim.get("effort", np.inf))
            jr.velocity = float(lim.get("velocity", np.inf))
        dyn = je.find("dynamics")
        if dyn is not None:
            jr.damping = float(dyn.get("damping", 0.0))
            jr.friction = float(dyn.get("friction", 0.0))
        mim = je.find("mimic")
        if mim is not None:
            jr.mimic = (mim.get("joint"), float(mim.get("multiplier", 1.0)), float(mim.get("offset", 0.0)))
        joints.append(jr)

    children = {j.child for j in joints}
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, f"URDF must have exactly one root link, found {roots}"
    root = roots[0]
    parent_joint = {j.child: j for j in joints}
    by_parent: Dict[str, List[JointRecord]] = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)
    # breadth-first like SAPIEN's articulation builder (parents always precede children)
    order, queue = [], [root]
    while queue:
        cur = queue.pop(0)
        order.append(cur)
        for j in by_parent.get(cur, []):
            queue.append(j.child)

    disabled = []
    if srdf_file is None:
        cand = os.path.splitext(urdf_file)[0] + ".srdf"
        if os.path.exists(cand):
            srdf_file = cand
    if srdf_file is not None and os.path.exists(srdf_file):
        for de in ET.parse(srdf_file).getroot().findall("disable_collisions"):
            disabled.append((de.get("link1"), de.get("link2")))

    return RobotDescription(
        name=robot.get("name", "robot"),
        links=links,
        joints=joints,
        root=root,
        link_order=order,
        parent_joint=parent_joint,
        disabled_pairs=disabled,
        package_dir=package_dir,
    )
