"""Batched SE(3) pose over a `[B, 7]` tensor (p, q wxyz) -- counterpart of
mani_skill/utils/structs/pose.py:31-272 with the same creation / broadcasting rules."""
from dataclasses import dataclass
from typing import List, Optional, Union

import numpy as np
import torch

from maniskill_amd.utils import common
from maniskill_amd.utils.geometry.rotation_conversions import quaternion_apply, quaternion_multiply, quaternion_to_matrix


def _is_sapien_pose(x):
    return hasattr(x, "p") and hasattr(x, "q") and not isinstance(x, Pose) and not isinstance(x, torch.Tensor)


def add_batch_dim(x):
    return x[None, :] if x.dim() == 1 else x


def to_batched_tensor(x, device=None):
    return None if x is None else add_batch_dim(common.to_tensor(x, device=device))


@dataclass
class Pose:
    raw_pose: torch.Tensor

    @classmethod
    def create_from_pq(cls, p=None, q=None, device=None):
        if device is None:
            device = p.device if isinstance(p, torch.Tensor) else (q.device if isinstance(q, torch.Tensor) else None)
        if p is None:
            p = torch.zeros((1, 3), device=device)
        if q is None:
            q = torch.zeros((1, 4), device=device)
            q[:, 0] = 1
        p, q = to_batched_tensor(p, device), to_batched_tensor(q, device)
        if p.shape[0] > q.shape[0]:
            assert q.shape[0] == 1
            q = q.repeat(p.shape[0], 1)
        elif p.shape[0] < q.shape[0]:
            assert p.shape[0] == 1
            p = p.repeat(q.shape[0], 1)
        return cls(raw_pose=torch.hstack([p, q]))

    @classmethod
    def create(cls, pose, device=None) -> "Pose":
        if isinstance(pose, cls):
            return cls(raw_pose=pose.raw_pose.to(device) if device is not None else pose.raw_pose)
        if _is_sapien_pose(pose):
            raw = torch.hstack([common.to_tensor(np.asarray(pose.p), device=device), common.to_tensor(np.asarray(pose.q), device=device)])
            return cls(raw_pose=add_batch_dim(raw))
        if isinstance(pose, list) and len(pose) > 0 and _is_sapien_pose(pose[0]):
            ps = common.to_tensor(np.array([np.asarray(x.p) for x in pose]), device=device)
            qs = common.to_tensor(np.array([np.asarray(x.q) for x in pose]), device=device)
            return cls(raw_pose=torch.hstack([ps, qs]))
        pose = add_batch_dim(common.to_tensor(pose, device=device))
        assert pose.dim() == 2
        if pose.shape[-1] == 3:
            return cls.create_from_pq(p=pose, device=pose.device)
        assert pose.shape[-1] == 7
        return cls(raw_pose=pose)

    def __getitem__(self, i):
        return Pose.create(self.raw_pose[i])

    def __len__(self):
        return len(self.raw_pose)

    @property
    def shape(self):
        return self.raw_pose.shape

    @property
    def device(self):
        return self.raw_pose.device

    def to(self, device):
        if self.raw_pose.device == torch.device(device):
            return self
        return Pose.create(self.raw_pose.to(device))

    def __mul__(self, other) -> "Pose":
        other = Pose.create(other, device=self.device)
        a = self
        if len(other) == 1 and len(a) > 1:
            other = Pose.create(other.raw_pose.repeat(len(a), 1))
        elif len(a) == 1 and len(other) > 1:
            a = Pose.create(a.raw_pose.repeat(len(other), 1))
        return Pose.create_from_pq(a.p + quaternion_apply(a.q, other.p), quaternion_multiply(a.q, other.q))

    def inv(self) -> "Pose":
        q = torch.cat((self.raw_pose[..., 3:4], -self.raw_pose[..., 4:]), -1)
        return Pose.create(torch.cat((quaternion_apply(q, -self.p), q), -1))

    def to_transformation_matrix(self):
        b = self.raw_pose.shape[0]
        mat = torch.zeros((b, 4, 4), device=self.raw_pose.device)
        mat[..., :3, :3] = quaternion_to_matrix(self.q)
        mat[..., :3, 3] = self.p
        mat[..., 3, 3] = 1
        return mat

    @property
    def sp(self):
        return to_sapien_pose(self)

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


def vectorize_pose(pose, device=None) -> torch.Tensor:
    if isinstance(pose, Pose):
        return pose.raw_pose.to(device) if device is not None else pose.raw_pose
    if _is_sapien_pose(pose):
        return torch.cat([common.to_tensor(np.asarray(pose.p), device=device), common.to_tensor(np.asarray(pose.q), device=device)])
    return common.to_tensor(pose, device=device)


def to_sapien_pose(pose):
    import sapien

    if _is_sapien_pose(pose):
        return pose
    raw = pose.raw_pose if isinstance(pose, Pose) else pose
    assert raw.dim() == 1 or (raw.dim() == 2 and raw.shape[0] == 1), "pose is batched; sapien poses are not"
    raw = common.to_numpy(raw[0] if raw.dim() == 2 else raw)
    return sapien.Pose(raw[:3], raw[3:])
