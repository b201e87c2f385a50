"""Small numpy SE(3) helpers used by the host-side model compiler (float64, wxyz quaternions).

Conventions follow the reference's `Pose` (mani_skill/utils/structs/pose.py:31-272): a pose is
p(3), q(4) with q = (w, x, y, z); `compose(a, b)` is a*b (b expressed in a's frame).
"""
import numpy as np


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ]
    )


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_to_mat(q):
    w, x, y, z = quat_normalize(q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def mat_to_quat(R):
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    if q[0] < 0:
        q = -q
    return quat_normalize(q)


def rpy_to_mat(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix (R = Rz(y) Ry(p) Rx(r))."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def rpy_to_quat(rpy):
    return mat_to_quat(rpy_to_mat(rpy))


def quat_rotate(q, v):
    return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


def pose(p=(0, 0, 0), q=(1, 0, 0, 0)):
    return np.concatenate([np.asarray(p, dtype=np.float64), quat_normalize(q)])


IDENTITY = pose()


def compose(a, b):
    """a * b"""
    return np.concatenate([a[:3] + quat_rotate(a[3:], b[:3]), quat_normalize(quat_mul(a[3:], b[3:]))])


def inverse(a):
    qi = quat_conj(a[3:])
    return np.concatenate([-quat_rotate(qi, a[:3]), qi])


def transform_point(a, x):
    return a[:3] + quat_rotate(a[3:], x)


def inertia_vec_to_mat(v):
    xx, yy, zz, xy, xz, yz = v
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def inertia_mat_to_vec(I):
    return np.array([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])


def combine_inertials(items):
    """items: list of (mass, com(3), I_com(3x3)) in one common frame -> combined (mass, com, I_com)."""
    m = sum(i[0] for i in items)
    if m <= 0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    com = sum(i[0] * np.asarray(i[1]) for i in items) / m
    I = np.zeros((3, 3))
    for mi, ci, Ii in items:
        d = np.asarray(ci) - com
        I += Ii + mi * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, com, I


def transform_inertial(T, mass, com, I):
    """Express an inertial given in frame B in frame A, T = pose of B in A."""
    R = quat_to_mat(T[3:])
    return mass, transform_point(T, com), R @ I @ R.T
