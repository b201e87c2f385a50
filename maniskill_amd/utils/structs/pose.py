"""Batched rigid transforms: `Pose.raw_pose` is a `[B, 7]` tensor, position then unit quaternion (w, x, y, z).

API counterpart of mani_skill/utils/structs/pose.py:31-272 (same constructors, same rule that a batch of one combines
with a batch of B). Everything funnels through two helpers: `_as_raw` turns whatever a caller hands in (a Pose, one or
several sapien-style poses, a position array, a 7-vector array) into the `[B, 7]` tensor, and `_same_rows` repeats a
single row to match its partner.
"""
from dataclasses import dataclass
from typing import List, Optional, Union  # noqa: F401

import numpy as np
import torch

from maniskill_amd.utils import common
from maniskill_amd.utils.geometry.rotation_conversions import quaternion_apply, quaternion_multiply, quaternion_to_matrix


def _is_sapien_pose(x) -> bool:
    """duck-typed single (unbatched) pose object with numpy `p` and `q`, e.g. the sapien.Pose stand-in"""
    return not isinstance(x, (Pose, torch.Tensor)) and hasattr(x, "p") and hasattr(x, "q")


def add_batch_dim(x: torch.Tensor) -> torch.Tensor:
    return x if x.dim() != 1 else x.unsqueeze(0)


def to_batched_tensor(x, device=None):
    if x is None:
        return None
    return add_batch_dim(common.to_tensor(x, device=device))


def _same_rows(a: torch.Tensor, b: torch.Tensor):
    """repeat whichever of the two has a single row so that both have the same number of rows"""
    na, nb = a.shape[0], b.shape[0]
    if na == nb:
        return a, b
    assert min(na, nb) == 1, f"cannot combine batches of {na} and {nb}"
    return (a.repeat(nb, 1), b) if na == 1 else (a, b.repeat(na, 1))


def _pq_of(obj, device):
    """(p, q) tensors of one sapien-style pose"""
    return common.to_tensor(np.asarray(obj.p), device=device), common.to_tensor(np.asarray(obj.q), device=device)


@dataclass
class Pose:
    raw_pose: torch.Tensor

    # ------------------------------------------------------------------ construction
    @classmethod
    def create_from_pq(cls, p=None, q=None, device=None):
        """from positions and / or quaternions; a missing part is the identity, a single row is repeated"""
        if device is None:
            device = next((t.device for t in (p, q) if isinstance(t, torch.Tensor)), None)
        pos = torch.zeros((1, 3), device=device) if p is None else to_batched_tensor(p, device)
        quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=device) if q is None else to_batched_tensor(q, device)
        pos, quat = _same_rows(pos, quat)
        return cls(raw_pose=torch.cat([pos, quat], dim=1))

    @classmethod
    def _as_raw(cls, obj, device=None) -> torch.Tensor:
        if isinstance(obj, cls):
            return obj.raw_pose if device is None else obj.raw_pose.to(device)
        if _is_sapien_pose(obj):
            return torch.cat(_pq_of(obj, device)).unsqueeze(0)
        if isinstance(obj, list) and obj and _is_sapien_pose(obj[0]):
            return torch.stack([torch.cat(_pq_of(o, device)) for o in obj])
        t = add_batch_dim(common.to_tensor(obj, device=device))
        assert t.dim() == 2 and t.shape[1] in (3, 7), f"expected [B, 3] positions or [B, 7] poses, got {tuple(t.shape)}"
        if t.shape[1] == 3:  # positions only
            return cls.create_from_pq(p=t, device=t.device).raw_pose
        return t

    @classmethod
    def create(cls, pose, device=None) -> "Pose":
        return cls(raw_pose=cls._as_raw(pose, device))

    # ------------------------------------------------------------------ container behaviour
    def __len__(self):
        return self.raw_pose.shape[0]

    def __getitem__(self, i):
        return Pose.create(self.raw_pose[i])

    @property
    def shape(self):
        return self.raw_pose.shape

    @property
    def device(self):
        return self.raw_pose.device

    def to(self, device):
        return self if self.raw_pose.device == torch.device(device) else Pose(raw_pose=self.raw_pose.to(device))

    # ------------------------------------------------------------------ algebra
    def __mul__(self, other) -> "Pose":
        """self o other: `other` expressed in self's frame"""
        a, b = _same_rows(self.raw_pose, Pose._as_raw(other, device=self.device))
        pa, qa, pb, qb = a[:, :3], a[:, 3:], b[:, :3], b[:, 3:]
        return Pose.create_from_pq(pa + quaternion_apply(qa, pb), quaternion_multiply(qa, qb))

    def inv(self) -> "Pose":
        quat = self.q
        conj = torch.cat([quat[..., :1], -quat[..., 1:]], dim=-1)
        return Pose.create(torch.cat([quaternion_apply(conj, -self.p), conj], dim=-1))

    def to_transformation_matrix(self):
        T = torch.eye(4, device=self.raw_pose.device).repeat(len(self), 1, 1)
        T[:, :3, :3] = quaternion_to_matrix(self.q)
        T[:, :3, 3] = self.p
        return T

    # ------------------------------------------------------------------ parts
    @property
    def p(self):
        return self.raw_pose[..., :3]

    @p.setter
    def p(self, v):
        self.raw_pose[..., :3] = common.to_tensor(v, device=self.raw_pose.device)

    @property
    def q(self):
        return self.raw_pose[..., 3:]

    @q.setter
    def q(self, v):
        self.raw_pose[..., 3:] = common.to_tensor(v, device=self.raw_pose.device)

    def get_p(self):
        return self.p

    def get_q(self):
        return self.q

    def set_p(self, p):
        self.p = p

    def set_q(self, q):
        self.q = q

    @property
    def sp(self):
        return to_sapien_pose(self)


def vectorize_pose(pose, device=None) -> torch.Tensor:
    """the 7 numbers of a pose object (batched for a Pose, flat for a single sapien-style pose), arrays pass through"""
    if isinstance(pose, Pose):
        return Pose._as_raw(pose, device)
    if _is_sapien_pose(pose):
        return torch.cat(_pq_of(pose, device))
    return common.to_tensor(pose, device=device)


def to_sapien_pose(pose):
    import sapien

    if _is_sapien_pose(pose):
        return pose
    raw = pose.raw_pose if isinstance(pose, Pose) else pose
    if raw.dim() == 2:
        assert raw.shape[0] == 1, "pose is batched; sapien poses are not"
        raw = raw[0]
    values = common.to_numpy(raw)
    return sapien.Pose(values[:3], values[3:])
