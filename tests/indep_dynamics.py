"""Independent numpy restatements used to PIN the oracle (not the product, not the oracle itself).

* `urdf_link_poses`   -- forward kinematics as a plain product of 4x4 matrices over the raw URDF
                         tree (all links, no fixed-joint folding).
* `aba_forward_dynamics` -- Featherstone's Articulated Body Algorithm in link (body) coordinates
                         with 6x6 Pluecker transforms (RBDA Table 7.1), over the raw URDF tree
                         (fixed joints are 0-DoF links). Structurally different from the oracle,
                         which uses world-frame CRBA + RNEA + dense LDL^T over folded bodies.
"""
import numpy as np

from maniskill_amd.model import geom
from maniskill_amd.model.urdf import RobotDescription


def _T(pose7):
    T = np.eye(4)
    T[:3, :3] = geom.quat_to_mat(pose7[3:])
    T[:3, 3] = pose7[:3]
    return T


def _axis_angle_mat(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def active_joint_names(rb: RobotDescription):
    return [rb.parent_joint[l].name for l in rb.link_order if l in rb.parent_joint and rb.parent_joint[l].type != "fixed"]


def urdf_link_poses(rb: RobotDescription, q, root_T=np.eye(4)):
    names = active_joint_names(rb)
    qmap = dict(zip(names, q))
    T = {rb.root: root_T}
    for l in rb.link_order:
        if l == rb.root:
            continue
        j = rb.parent_joint[l]
        X = T[j.parent] @ _T(j.origin)
        if j.type in ("revolute", "continuous"):
            M = np.eye(4)
            M[:3, :3] = _axis_angle_mat(j.axis, qmap[j.name])
            X = X @ M
        elif j.type == "prismatic":
            M = np.eye(4)
            M[:3, 3] = j.axis * qmap[j.name]
            X = X @ M
        T[l] = X
    return T


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _xform(E, r):
    """Pluecker motion transform from frame A to frame B, where B is at position r (in A coords)
    with rotation E (A coords -> B coords)."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ _skew(r)
    return X


def _crm(v):
    M = np.zeros((6, 6))
    M[:3, :3] = _skew(v[:3])
    M[3:, 3:] = _skew(v[:3])
    M[3:, :3] = _skew(v[3:])
    return M


def _crf(v):
    return -_crm(v).T


def _spatial_inertia(m, c, I):
    C = _skew(c)
    M = np.zeros((6, 6))
    M[:3, :3] = I + m * C @ C.T
    M[:3, 3:] = m * C
    M[3:, :3] = m * C.T
    M[3:, 3:] = m * np.eye(3)
    return M


def aba_forward_dynamics(rb: RobotDescription, q, qd, tau, gravity=(0, 0, -9.81), link_gravity=None, root_T=np.eye(4)):
    """qdd for the raw URDF tree with fixed base. `link_gravity[name]` False disables gravity on
    that link (applied as an external force, like the oracle)."""
    names = active_joint_names(rb)
    idx = {n: i for i, n in enumerate(names)}
    links = rb.link_order
    li = {l: i for i, l in enumerate(links)}
    n = len(links)
    g = np.asarray(gravity, dtype=np.float64)
    Tw = urdf_link_poses(rb, q, root_T)
    Xup, S, v, c, IA, pA = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    parent = [-1] * n
    for i, l in enumerate(links):
        link = rb.links[l]
        I = _spatial_inertia(link.mass, link.com, link.inertia) if link.has_inertial else np.zeros((6, 6))
        IA[i] = I.copy()
        if l == rb.root:
            v[i] = np.zeros(6)
            S[i] = None
            continue
        j = rb.parent_joint[l]
        parent[i] = li[j.parent]
        # transform parent link frame -> this link frame
        Tpl = np.linalg.inv(Tw[j.parent]) @ Tw[l]
        E = Tpl[:3, :3].T
        Xup[i] = _xform(E, Tpl[:3, 3])
        if j.type in ("revolute", "continuous"):
            S[i] = np.concatenate([j.axis, np.zeros(3)])
            qdi = qd[idx[j.name]]
        elif j.type == "prismatic":
            S[i] = np.concatenate([np.zeros(3), j.axis])
            qdi = qd[idx[j.name]]
        else:
            S[i] = None
            qdi = 0.0
        vJ = S[i] * qdi if S[i] is not None else np.zeros(6)
        v[i] = Xup[i] @ v[parent[i]] + vJ
        c[i] = _crm(v[i]) @ vJ
    for i, l in enumerate(links):
        link = rb.links[l]
        I = IA[i]
        pA[i] = _crf(v[i]) @ (I @ v[i]) if v[i] is not None else np.zeros(6)
        use_g = True if link_gravity is None else link_gravity.get(l, True)
        if use_g and link.has_inertial and link.mass > 0:
            # external force m g at the COM, expressed in link coordinates
            Rw = Tw[l][:3, :3]
            f_l = Rw.T @ (link.mass * g)
            pA[i] = pA[i] - np.concatenate([np.cross(link.com, f_l), f_l])
    U, d, u = [None] * n, [None] * n, [None] * n
    for i in range(n - 1, 0, -1):
        p = parent[i]
        if S[i] is not None:
            U[i] = IA[i] @ S[i]
            d[i] = S[i] @ U[i]
            u[i] = tau[idx[rb.parent_joint[links[i]].name]] - S[i] @ pA[i]
            Ia = IA[i] - np.outer(U[i], U[i]) / d[i]
            pa = pA[i] + Ia @ c[i] + U[i] * u[i] / d[i]
        else:
            Ia = IA[i]
            pa = pA[i] + Ia @ c[i]
        IA[p] = IA[p] + Xup[i].T @ Ia @ Xup[i]
        pA[p] = pA[p] + Xup[i].T @ pa
    a = [None] * n
    a[0] = np.zeros(6)
    qdd = np.zeros(len(names))
    for i in range(1, n):
        p = parent[i]
        ap = Xup[i] @ a[p] + c[i]
        if S[i] is not None:
            k = idx[rb.parent_joint[links[i]].name]
            qdd[k] = (u[i] - U[i] @ ap) / d[i]
            a[i] = ap + S[i] * qdd[k]
        else:
            a[i] = ap
    return qdd
