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
    return matrix_to_quaternion(euler_angles_to_matrix(ang, convention="XYZ"))
