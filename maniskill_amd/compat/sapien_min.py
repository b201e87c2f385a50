"""Minimal `sapien` stand-in: an unbatched `Pose` plus inert records task files construct.

Semantics of `sapien.Pose` as the reference relies on them (tests/structs/test_pose.py:10-120):
`p` (3,), `q` (4,) wxyz, `inv()`, `*`, `to_transformation_matrix()`, construction from p/q or a
4x4 matrix.
"""
import sys
import types

import numpy as np

from ..model import geom


class Pose:
    def __init__(self, p=None, q=None):
        if p is not None and np.ndim(p) == 2 and np.shape(p) == (4, 4) and q is None:
            T = np.asarray(p, dtype=np.float64)
            self.p = T[:3, 3].astype(np.float32)
            self.q = geom.mat_to_quat(T[:3, :3]).astype(np.float32)
            return
        self.p = np.zeros(3, dtype=np.float32) if p is None else np.asarray(p, dtype=np.float32).reshape(3).copy()
        self.q = np.array([1, 0, 0, 0], dtype=np.float32) if q is None else np.asarray(q, dtype=np.float32).reshape(4).copy()

    def set_p(self, p):
        self.p = np.asarray(p, dtype=np.float32).reshape(3)

    def set_q(self, q):
        self.q = np.asarray(q, dtype=np.float32).reshape(4)

    def get_p(self):
        return self.p

    def get_q(self):
        return self.q

    def _pose7(self):
        return geom.pose(self.p, self.q)

    def inv(self):
        r = geom.inverse(self._pose7())
        return Pose(r[:3], r[3:])

    def __mul__(self, other):
        r = geom.compose(self._pose7(), other._pose7())
        return Pose(r[:3], r[3:])

    def to_transformation_matrix(self):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = geom.quat_to_mat(self.q)
        T[:3, 3] = self.p
        return T

    def __repr__(self):
        return f"Pose({self.p.tolist()}, {self.q.tolist()})"


class Device:
    def __init__(self, name="cuda"):
        self.name = str(name)

    def is_cuda(self):
        return self.name.startswith("cuda")

    def is_cpu(self):
        return self.name == "cpu"


class RenderMaterial:
    """inert record (state observations only; no renderer in this build)"""

    def __init__(self, base_color=(1, 1, 1, 1), **kw):
        self.base_color = list(base_color)
        self.__dict__.update(kw)


def install_as(name):
    root = types.ModuleType(name)
    root.Pose, root.Device = Pose, Device
    render = types.ModuleType(name + ".render")
    render.RenderMaterial = RenderMaterial
    root.render = render
    physx = types.ModuleType(name + ".physx")
    root.physx = physx
    root.__maniskill_amd_shim__ = True
    sys.modules[name] = root
    sys.modules[name + ".render"] = render
    sys.modules[name + ".physx"] = physx
