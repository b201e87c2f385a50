"""`transforms3d.euler.euler2quat` / `quat2euler` (static 'sxyz' convention, wxyz quaternions)."""
import sys
import types

import numpy as np


def euler2quat(ai, aj, ak, axes="sxyz"):
    if axes != "sxyz":
        raise NotImplementedError(axes)
    ci, si = np.cos(ai / 2), np.sin(ai / 2)
    cj, sj = np.cos(aj / 2), np.sin(aj / 2)
    ck, sk = np.cos(ak / 2), np.sin(ak / 2)
    return np.array(
        [
            ci * cj * ck + si * sj * sk,
            si * cj * ck - ci * sj * sk,
            ci * sj * ck + si * cj * sk,
            ci * cj * sk - si * sj * ck,
        ]
    )


def quat2euler(q, axes="sxyz"):
    if axes != "sxyz":
        raise NotImplementedError(axes)
    w, x, y, z = q
    roll = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    pitch = np.arcsin(np.clip(2 * (w * y - z * x), -1, 1))
    yaw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return roll, pitch, yaw


def quat2mat(q):
    from ..model.geom import quat_to_mat

    return quat_to_mat(q)


def mat2quat(m):
    from ..model.geom import mat_to_quat

    return mat_to_quat(m)


def install_as(name):
    root = types.ModuleType(name)
    euler = types.ModuleType(name + ".euler")
    euler.euler2quat, euler.quat2euler = euler2quat, quat2euler
    quats = types.ModuleType(name + ".quaternions")
    quats.quat2mat, quats.mat2quat = quat2mat, mat2quat
    root.euler, root.quaternions = euler, quats
    sys.modules[name] = root
    sys.modules[name + ".euler"] = euler
    sys.modules[name + ".quaternions"] = quats
