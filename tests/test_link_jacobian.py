"""`link_jacobian` (include/mssim.h) pinned against central finite differences of the oracle's own FK
(CPU, f64), for every link of the Panda. The reference gets this matrix from pytorch_kinematics
(`chain.jacobian`, controllers/utils/kinematics.py:156-171), which is not installed: parity with that
library is unpinned, the definition (d link-origin velocity / d joint velocity, root frame, rows
linear then angular) is what is tested."""
import numpy as np
import torch

from maniskill_amd.model.scenes import panda_tabletop_model
from tests import oracle_backend as ob


def _link_poses(px, model, N):
    px.gpu_apply_all()
    px.gpu_update_articulation_kinematics()
    px.gpu_fetch_all()
    return px.cuda_rigid_body_data.torch()[: model.n_link * N].reshape(model.n_link, N, 13)[:, :, :7].double().clone()


def _quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def test_link_jacobian_matches_finite_differences():
    model = panda_tabletop_model()
    N = 6
    px = ob.make_system(model, N, precision="f64")
    g = torch.Generator().manual_seed(5)
    lo = torch.tensor(model.arrays["dof_limit"][:, 0]), torch.tensor(model.arrays["dof_limit"][:, 1])
    q0 = (lo[0] + (lo[1] - lo[0]) * torch.rand(N, model.n_dof, generator=g)).float()
    # a tilted, displaced base so that the root-frame convention is exercised
    root = px.cuda_articulation_root_pose.torch() if hasattr(px, "cuda_articulation_root_pose") else None
    eps = 1e-3  # buffers are f32: a larger step keeps the difference quotient above the rounding noise
    for link in (model.link_names.index("panda_hand_tcp"), model.link_names.index("panda_link4"), model.link_names.index("panda_leftfinger")):
        px.cuda_articulation_qpos.torch()[:] = q0
        _link_poses(px, model, N)
        J = px.link_jacobian(link).double()
        for j in range(model.n_dof):
            qp, qm = q0.clone(), q0.clone()
            qp[:, j] += eps
            qm[:, j] -= eps
            px.cuda_articulation_qpos.torch()[:] = qp
            Pp = _link_poses(px, model, N)[link]
            px.cuda_articulation_qpos.torch()[:] = qm
            Pm = _link_poses(px, model, N)[link]
            lin = (Pp[:, :3] - Pm[:, :3]) / (2 * eps)
            dq = _quat_mul(Pp[:, 3:], Pm[:, 3:] * torch.tensor([1.0, -1, -1, -1], dtype=torch.float64))
            ang = 2 * dq[:, 1:] / (2 * eps) * torch.sign(dq[:, :1])
            assert torch.allclose(J[:, :3, j], lin, atol=2e-4), (link, j, (J[:, :3, j] - lin).abs().max())
            assert torch.allclose(J[:, 3:, j], ang, atol=2e-4), (link, j, (J[:, 3:, j] - ang).abs().max())
    # columns of joints that do not move the link are exactly zero (finger joints for link 4)
    px.cuda_articulation_qpos.torch()[:] = q0
    _link_poses(px, model, N)
    J4 = px.link_jacobian(model.link_names.index("panda_link4"))
    assert torch.all(J4[:, :, 4:] == 0)
