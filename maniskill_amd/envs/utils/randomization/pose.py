"""random orientations (counterpart of mani_skill/envs/utils/randomization/pose.py:13-34)"""
import numpy as np
import torch

from maniskill_amd.utils.geometry.rotation_conversions import euler_angles_to_matrix, matrix_to_quaternion


def random_quaternions(n: int, device=None, lock_x: bool = False, lock_y: bool = False, lock_z: bool = False, bounds=(0, np.pi * 2)):
    """uniform XYZ Euler angles in [bounds) with optional locked axes -> quaternions (wxyz)"""
    ang = torch.rand((n, 3), device=device) * (bounds[1] - bounds[0]) + bounds[0]
    if lock_x:
        ang[:, 0] *= 0
    if lock_y:
        ang[:, 1] *= 0
    if lock_z:
        ang[:, 2] *= 0
    if lock_x and lock_y and not lock_z:
        return _yaw_quaternions(ang[:, 2])
    return matrix_to_quaternion(euler_angles_to_matrix(ang, convention="XYZ"))


def _yaw_quaternions(t: torch.Tensor) -> torch.Tensor:
    """matrix_to_quaternion(euler_angles_to_matrix([0, 0, t], "XYZ")) for the common yaw-only draw (cube / peg / box
    placement of the tabletop tasks), written out: the same floating-point expressions in the same order -- equal
    bit for bit to the generic route (tests/test_rotation_golden.py) -- in 20 elementwise ops instead of ~60 with
    batched 3x3 products (a third of a partial reset's host time)."""
    c, s = torch.cos(t), torch.sin(t)
    zero = torch.zeros_like(c)
    # Rz: m00 = m11 = c, m10 = -m01 = s, m22 = 1; q_abs = sqrt(max(0, [1+m00+m11+m22, ., ., 1-m00-m11+m22]))
    a0, a3 = 1.0 + c + c + 1.0, 1.0 - c - c + 1.0
    q0 = torch.where(a0 > 0, torch.sqrt(a0.clamp_min(0)), zero)
    q3 = torch.where(a3 > 0, torch.sqrt(a3.clamp_min(0)), zero)
    two_s = s - (-s)
    d0, d3 = 2.0 * q0.clamp_min(0.1), 2.0 * q3.clamp_min(0.1)
    use0 = q0 >= q3  # argmax over (q0, 0, 0, q3): the first maximum
    w = torch.where(use0, q0 * q0 / d0, two_s / d3)
    z = torch.where(use0, two_s / d0, q3 * q3 / d3)
    neg = w < 0  # standardize_quaternion
    return torch.stack((torch.where(neg, -w, w), zero, zero, torch.where(neg, -z, z)), -1)
