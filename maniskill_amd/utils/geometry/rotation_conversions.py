"""Batched quaternion / rotation-matrix / Euler conversions in torch (wxyz, real part first).

Own restatement of the functions of mani_skill/utils/geometry/rotation_conversions.py that the
hot-path surface uses (`quaternion_multiply :407`, `quaternion_apply :441`,
`quaternion_to_matrix :44`, `matrix_to_quaternion :105`, `euler_angles_to_matrix :197`, ...),
checked against golden vectors generated from that file (tests/golden/rotation_golden.npz).
"""
import torch


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Hamilton product in scalar / vector form (w = aw bw - av.bv, v = aw bv + bw av + av x bv):
    the same product as the component-wise expansion, in ~9 launches instead of ~30"""
    aw, av = a[..., :1], a[..., 1:]
    bw, bv = b[..., :1], b[..., 1:]
    w = aw * bw - (av * bv).sum(-1, keepdim=True)
    v = aw * bv + bw * av + torch.linalg.cross(av, bv, dim=-1)
    return torch.cat((w, v), -1)


def standardize_quaternion(q: torch.Tensor) -> torch.Tensor:
    """flip so that the real part is non-negative"""
    return torch.where(q[..., 0:1] < 0, -q, q)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def quaternion_invert(q: torch.Tensor) -> torch.Tensor:
    return torch.cat((q[..., :1], -q[..., 1:]), -1)


def quaternion_apply(q: torch.Tensor, point: torch.Tensor) -> torch.Tensor:
    """rotate points by unit quaternions: p + 2 w (v x p) + 2 v x (v x p), which equals the vector
    part of q (0,p) q* for unit q"""
    if point.size(-1) != 3:
        raise ValueError(f"Points are not in 3D, {point.shape}.")
    w, v = q[..., :1], q[..., 1:]
    v, point = torch.broadcast_tensors(v, point)
    t = 2 * torch.linalg.cross(v, point, dim=-1)
    return point + w * t + torch.linalg.cross(v, t, dim=-1)


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    # sqrt(max(0, x)) with a zero result for x <= 0 -- as a select: boolean-mask indexing would synchronise the
    # host with the device on every call (every reset draws random orientations through this)
    return torch.where(x > 0, torch.sqrt(x.clamp_min(0)), torch.zeros_like(x))


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    batch = matrix.shape[:-2]
    m = matrix.reshape(batch + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, -1)
    q_abs = _sqrt_positive_part(
        torch.stack((1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22), -1)
    )
    cand = torch.stack(
        (
            torch.stack((q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01), -1),
            torch.stack((m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20), -1),
            torch.stack((m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21), -1),
            torch.stack((m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2), -1),
        ),
        -2,
    )
    cand = cand / (2.0 * q_abs[..., None].max(q_abs.new_tensor(0.1)))
    # the candidate with the largest denominator (gather on the argmax: no boolean-mask indexing, no host sync)
    idx = q_abs.argmax(-1)[..., None, None].expand(batch + (1, 4))
    return standardize_quaternion(torch.gather(cand, -2, idx).reshape(batch + (4,)))


def _axis_angle_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    c, s = torch.cos(angle), torch.sin(angle)
    one, zero = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == "X":
        flat = (one, zero, zero, zero, c, -s, zero, s, c)
    elif axis == "Y":
        flat = (c, zero, s, zero, one, zero, -s, zero, c)
    elif axis == "Z":
        flat = (c, -s, zero, s, c, zero, zero, zero, one)
    else:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    mats = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(mats[0], mats[1]), mats[2])


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    k = torch.empty_like(angles)
    k[~small] = torch.sin(half[~small]) / angles[~small]
    k[small] = 0.5 - (angles[small] * angles[small]) / 48
    return torch.cat([torch.cos(half), axis_angle * k], dim=-1)


def quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    k = torch.empty_like(angles)
    k[~small] = torch.sin(half[~small]) / angles[~small]
    k[small] = 0.5 - (angles[small] * angles[small]) / 48
    return q[..., 1:] / k


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def euler_angles_to_quaternion(euler_angles: torch.Tensor, convention: str = "XYZ") -> torch.Tensor:
    return matrix_to_quaternion(euler_angles_to_matrix(euler_angles, convention))
