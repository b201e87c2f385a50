"""rotation_conversions restatement vs golden vectors produced by the reference's own file
(tests/golden/make_rotation_golden.py)."""
import os

import numpy as np
import pytest
import torch

from maniskill_amd.utils.geometry import rotation_conversions as rc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotation_golden.npz"))
# the golden vectors are the only reference-generated fixture of the path: they are checked on CPU tensors here and
# on the MI355X (`cuda` tensors, the device every struct getter of the product returns) under the gpu mark
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
_dev = "cpu"
t = lambda k: torch.from_numpy(G[k]).to(_dev)


@pytest.fixture(autouse=True, params=DEVICES)
def on_device(request):
    global _dev
    _dev = request.param
    yield
    _dev = "cpu"


def close(a, k, atol=1e-6):
    assert a.device.type == _dev
    np.testing.assert_allclose(a.cpu().numpy(), G[k], atol=atol, rtol=1e-5)


def test_quaternion_ops_match_reference():
    qa, qb, pts = t("qa"), t("qb"), t("pts")
    close(rc.quaternion_multiply(qa, qb), "quaternion_multiply")
    close(rc.quaternion_raw_multiply(qa, qb), "quaternion_raw_multiply")
    close(rc.quaternion_apply(qa, pts), "quaternion_apply")
    close(rc.quaternion_invert(qa), "quaternion_invert")
    close(rc.standardize_quaternion(qa), "standardize_quaternion")


def test_matrix_conversions_match_reference():
    qa = t("qa")
    close(rc.quaternion_to_matrix(qa), "quaternion_to_matrix")
    close(rc.matrix_to_quaternion(t("quaternion_to_matrix")), "matrix_to_quaternion", atol=1e-5)
    close(rc.euler_angles_to_matrix(t("eul"), "XYZ"), "euler_XYZ")
    close(rc.euler_angles_to_matrix(t("eul"), "ZYX"), "euler_ZYX")
    close(rc.axis_angle_to_quaternion(t("aa")), "axis_angle_to_quaternion")
    close(rc.quaternion_to_axis_angle(qa), "quaternion_to_axis_angle", atol=1e-5)


def test_yaw_only_random_quaternions_equal_the_generic_route():
    if _dev != "cpu":
        pytest.skip("device variant: tests/test_gpu_env.py")
    """the yaw-only fast path of random_quaternions is the generic euler -> matrix -> quaternion route, bit for bit"""
    import numpy as np
    import torch

    from maniskill_amd.envs.utils.randomization.pose import _yaw_quaternions
    from maniskill_amd.utils.geometry.rotation_conversions import euler_angles_to_matrix, matrix_to_quaternion

    g = torch.Generator().manual_seed(0)
    t = torch.rand(100000, generator=g) * 2 * np.pi
    t[:8] = torch.tensor([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi - 1e-7, 1e-8, np.pi - 1e-7, np.pi + 1e-7])
    ang = torch.zeros(len(t), 3)
    ang[:, 2] = t
    assert torch.equal(_yaw_quaternions(t), matrix_to_quaternion(euler_angles_to_matrix(ang, "XYZ")))
