"""Batched kinematics for the end-effector controllers (API counterpart of
mani_skill/agents/controllers/utils/kinematics.py:29-186).

The reference builds a pytorch_kinematics serial chain from the URDF and, on the GPU, uses
  * `chain.jacobian(q)` + `torch.linalg.pinv` for the delta controllers (one pseudo-inverse step,
    kinematics.py:156-171), and
  * pytorch_kinematics' `PseudoInverseIK` (200 iterations) for absolute / target-pose control.
Here the Jacobian of the delta path comes from the simulation core itself
(`px.link_jacobian`, include/mssim.h: world joint axes / anchors of the last FK), and the iterative
path is an own damped-least-squares solver over a torch FK of the same chain built from the compiled
model tables. pytorch_kinematics is not installed: the delta step follows the reference formula
exactly; the iterative solver's convergence path is parity-unpinned (same fixed point: the target pose).
"""
from typing import List

import numpy as np
import torch

from maniskill_amd.utils.geometry.rotation_conversions import quaternion_apply, quaternion_multiply
from maniskill_amd.utils.structs.pose import Pose


def _axis_angle_quat(axis: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
    h = 0.5 * angle
    return torch.cat((torch.cos(h)[..., None], axis * torch.sin(h)[..., None]), -1)


class Kinematics:
    def __init__(self, urdf_path: str, end_link_name: str, articulation, active_joint_indices: torch.Tensor):
        self.urdf_path = urdf_path
        self.articulation = articulation
        self.device = articulation.device
        self.end_link = articulation.links_map[end_link_name]
        self.end_link_idx = articulation.links.index(self.end_link)
        self.active_joint_indices = active_joint_indices
        # active joints on the path root -> end link (everything past the end link is ignored)
        chain = []
        link = self.end_link
        while link is not None and link.joint is not None:
            if link.joint.active_index_int is not None:
                chain.append(link.joint)
            link = link.joint.parent_link
        self.active_ancestor_joints = chain[::-1]
        self.active_ancestor_joint_idxs: List[int] = [j.active_index_int for j in self.active_ancestor_joints]
        self.controlled_joints_idx_in_qmask = [self.active_ancestor_joint_idxs.index(int(i)) for i in active_joint_indices]
        self.qmask = torch.zeros(len(self.active_ancestor_joints), dtype=torch.bool, device=self.device)
        self.qmask[self.controlled_joints_idx_in_qmask] = True
        self.use_gpu_ik = True
        self._chain = None

    # ------------------------------------------------------------------ torch FK of the serial chain
    def _build_chain(self):
        model = self.articulation.scene.model
        A = model.arrays
        idx = self.active_ancestor_joint_idxs
        for a, b in zip(idx[1:], idx[:-1]):
            assert int(A["dof_parent"][a]) == b, "end link's active ancestors must form a serial chain"
        assert int(A["dof_parent"][idx[0]]) == -1
        dev = self.device
        self._chain = dict(
            frame=torch.tensor(A["dof_frame"][idx], dtype=torch.float32, device=dev),
            axis=torch.tensor(A["dof_axis"][idx], dtype=torch.float32, device=dev),
            revolute=[int(A["dof_type"][i]) == 0 for i in idx],
            tip=torch.tensor(A["link_frame"][self.end_link_idx], dtype=torch.float32, device=dev),
            lim=torch.tensor(A["dof_limit"][idx], dtype=torch.float32, device=dev),
        )
        assert int(A["link_body"][self.end_link_idx]) == idx[-1]

    def fk_jacobian(self, q: torch.Tensor):
        """end-link pose (p [B,3], quat [B,4]) and geometric Jacobian [B,6,n] (linear; angular) in the
        root frame for chain joint positions q [B,n]"""
        if self._chain is None:
            self._build_chain()
        c = self._chain
        B, n = q.shape
        p = torch.zeros((B, 3), device=q.device)
        r = torch.zeros((B, 4), device=q.device)
        r[:, 0] = 1
        axes, anchors = [], []
        for i in range(n):
            fp, fq = c["frame"][i, :3], c["frame"][i, 3:]
            jp = p + quaternion_apply(r, fp.expand(B, 3))
            jq = quaternion_multiply(r, fq.expand(B, 4))
            a = quaternion_apply(jq, c["axis"][i].expand(B, 3))
            axes.append(a)
            anchors.append(jp)
            if c["revolute"][i]:
                p, r = jp, quaternion_multiply(jq, _axis_angle_quat(c["axis"][i].expand(B, 3), q[:, i]))
            else:
                p, r = jp + a * q[:, i : i + 1], jq
        pe = p + quaternion_apply(r, c["tip"][:3].expand(B, 3))
        qe = quaternion_multiply(r, c["tip"][3:].expand(B, 4))
        J = torch.zeros((B, 6, n), device=q.device)
        for i in range(n):
            if c["revolute"][i]:
                J[:, :3, i] = torch.linalg.cross(axes[i], pe - anchors[i])
                J[:, 3:, i] = axes[i]
            else:
                J[:, :3, i] = axes[i]
        return pe, qe, J

    # ------------------------------------------------------------------ IK
    def compute_ik(self, target_pose: Pose, q0: torch.Tensor, pos_only: bool = False, action=None, use_delta_ik_solver: bool = False):
        """Target joint positions of the chain joints for an end-link target pose given in the ROOT frame
        (same contract as the reference, kinematics.py:124-186)."""
        q0 = q0[:, self.active_ancestor_joint_idxs]
        if use_delta_ik_solver:
            # one pseudo-inverse step of the commanded delta (Buss, "Introduction to inverse kinematics")
            J = self.articulation.px.link_jacobian(self.end_link_idx)[:, :, self.active_ancestor_joint_idxs]
            if pos_only:
                J = J[:, 0:3]
            # J^+ a = J^T (J J^T)^-1 a for the full-row-rank J of a 7-joint arm (what pinv returns away
            # from singularities); the tiny ridge keeps the 3x3 / 6x6 solve finite at a singular pose
            # where pinv would truncate. A batched SVD over 4096 envs costs ~8 ms, this ~0.2 ms.
            JJt = J @ J.transpose(1, 2)
            JJt = JJt + 1e-9 * torch.eye(JJt.shape[1], device=J.device)
            return q0 + (J.transpose(1, 2) @ torch.linalg.solve(JJt, action.unsqueeze(-1))).squeeze(-1)
        # damped least squares on the pose error (own solver, see module docstring)
        tp, tq = target_pose.p, target_pose.q
        q = q0.clone()
        lam = 1e-3
        if self._chain is None:
            self._build_chain()
        lo, hi = self._chain["lim"][:, 0], self._chain["lim"][:, 1]
        conj = torch.tensor([1.0, -1.0, -1.0, -1.0], device=q.device)
        for _ in range(60):
            pe, qe, J = self.fk_jacobian(q)
            err = tp - pe
            if not pos_only:
                dq = quaternion_multiply(tq, qe * conj)
                dq = torch.where(dq[:, :1] < 0, -dq, dq)
                nv = torch.linalg.norm(dq[:, 1:], dim=1, keepdim=True)
                ang = 2 * torch.atan2(nv, dq[:, :1])
                err = torch.cat((err, dq[:, 1:] / nv.clamp_min(1e-9) * ang), 1)  # rotation vector of the remaining turn
            else:
                J = J[:, :3]
            if float(err.abs().max()) < 1e-5:
                break
            JJt = J @ J.transpose(1, 2) + lam * torch.eye(J.shape[1], device=q.device)
            step = (J.transpose(1, 2) @ torch.linalg.solve(JJt, err.unsqueeze(-1))).squeeze(-1)
            step = step * (0.3 / step.abs().amax(dim=1, keepdim=True).clamp_min(0.3))  # at most 0.3 rad per joint per iteration
            q = torch.minimum(torch.maximum(q + step, lo), hi)
        return q
