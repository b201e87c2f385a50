"""An independent, DIRECT solution of the contact problem of one substep, to pin the oracle's (and the kernel's) iterative
solver -- the way tests/indep_dynamics.py pins the articulated dynamics with a different algorithm.

The product and the oracle run projected Gauss-Seidel sweeps over rows (normal + two friction rows per point, box
pyramid |lambda_t| <= mu lambda_n per tangent axis, speculative bias gap / dt). Here the same physical model is written as
one linear complementarity problem over free rigid bodies and solved by LEMKE's complementary pivoting (Cottle, Pang,
Stone, "The Linear Complementarity Problem", ch. 4.4): no sweeps, no iteration count, no warm start. Per tangent axis j
of a contact the friction impulse is beta_j+ - beta_j- with beta >= 0 and a slack sigma_j (Stewart & Trinkle 1996;
Anitescu & Potra 1997, written per axis so that the admissible set is the solver's box pyramid, not the diamond):

    w_n   = Jn v+ + gap / dt                     >= 0   _|_  lambda_n >= 0
    w_j+  =  (Jt_j v+) + sigma_j                 >= 0   _|_  beta_j+  >= 0
    w_j-  = -(Jt_j v+) + sigma_j                 >= 0   _|_  beta_j-  >= 0
    w_sj  = mu lambda_n - beta_j+ - beta_j-      >= 0   _|_  sigma_j  >= 0
    v+    = v* + M^-1 J^T lambda,     v* = v + dt (g + f / m)  (bodies start without spin in the scenes used: no gyroscopic term)

The multipliers of coplanar points are not unique (a box on four corners is statically indeterminate); what is unique, and
what the tests compare, are the bodies' velocities after the substep and the impulse each body pair exchanges.
"""
import numpy as np


def lemke(M, q, max_iter=2000):
    """w = M z + q >= 0, z >= 0, z.w = 0 by Lemke's algorithm with covering vector d = 1 (Cottle, Pang, Stone, algorithm 4.4.5)."""
    n = len(q)
    if np.all(q >= 0):
        return np.zeros(n)
    # tableau [I | -M | -d | q]: basis starts as w
    T = np.hstack([np.eye(n), -M, -np.ones((n, 1)), q.reshape(-1, 1)]).astype(np.float64)
    basis = list(range(n))  # 0..n-1: w, n..2n-1: z, 2n: z0
    r = int(np.argmin(q))
    entering = 2 * n

    def pivot(row, col):
        T[row] /= T[row, col]
        for i in range(n):
            if i != row:
                T[i] -= T[i, col] * T[row]

    pivot(r, entering)
    leaving, basis[r] = basis[r], entering
    for _ in range(max_iter):
        entering = leaving + n if leaving < n else leaving - n  # the complement of what left
        col = T[:, entering]
        rows = np.where(col > 1e-12)[0]
        if len(rows) == 0:
            raise RuntimeError("Lemke: ray termination")
        ratios = T[rows, -1] / col[rows]
        best = ratios.min()
        ties = rows[ratios <= best + 1e-14]
        # if z0 can leave, let it (the solution is reached)
        row = next((i for i in ties if basis[i] == 2 * n), ties[0])
        pivot(row, entering)
        leaving, basis[row] = basis[row], entering
        if leaving == 2 * n:
            z = np.zeros(2 * n + 1)
            for i, b in enumerate(basis):
                z[b] = T[i, -1]
            return z[n : 2 * n]
    raise RuntimeError("Lemke: iteration limit")


def tangents(n):
    """the friction axes of a contact normal: the solver's documented convention (include/mssim.h: pyramid friction along
    t1 = normalize(n x e_x) -- n x e_y when n is within 54.7 degrees of e_x --, t2 = n x t1)"""
    n = np.asarray(n, dtype=np.float64)
    helper = np.array([1.0, 0, 0]) if abs(n[0]) < 0.57735 else np.array([0, 1.0, 0])
    t1 = np.cross(n, helper)
    t1 /= np.linalg.norm(t1)
    return t1, np.cross(n, t1)


def solve_substep(bodies, contacts, dt, gravity=(0, 0, -9.81)):
    """bodies: dicts(m, I = 3x3 world inertia about the centre of mass, x = centre of mass, v, w, f = external force, gravity: bool);
    contacts: (a, b, x, n, gap, mu): body indices (-1 = fixed), point, unit normal from b towards a, signed gap, friction.
    -> (v+ [nb, 6] lin | ang, impulses [nc, 3] world impulse on body a of each contact)"""
    nb, nc = len(bodies), len(contacts)
    g = np.asarray(gravity, dtype=np.float64)
    Minv = np.zeros((6 * nb, 6 * nb))
    vstar = np.zeros(6 * nb)
    for i, B in enumerate(bodies):
        Minv[6 * i : 6 * i + 3, 6 * i : 6 * i + 3] = np.eye(3) / B["m"]
        Minv[6 * i + 3 : 6 * i + 6, 6 * i + 3 : 6 * i + 6] = np.linalg.inv(B["I"])
        acc = (g if B.get("gravity", True) else 0 * g) + np.asarray(B.get("f", np.zeros(3))) / B["m"]
        vstar[6 * i : 6 * i + 3] = np.asarray(B["v"]) + dt * acc
        vstar[6 * i + 3 : 6 * i + 6] = np.asarray(B["w"])

    def row(a, b, x, d):
        J = np.zeros(6 * nb)
        for body, sgn in ((a, 1.0), (b, -1.0)):
            if body >= 0:
                J[6 * body : 6 * body + 3] = sgn * d
                J[6 * body + 3 : 6 * body + 6] = sgn * np.cross(np.asarray(x) - bodies[body]["x"], d)
        return J

    Jn = np.array([row(a, b, x, np.asarray(n, float)) for a, b, x, n, gap, mu in contacts])
    Jt = np.array([row(a, b, x, t) for a, b, x, n, gap, mu in contacts for t in tangents(n)])  # [2 nc]
    mu = np.array([c[5] for c in contacts])
    bias = np.array([c[4] for c in contacts]) / dt
    nt = 2 * nc
    # unknowns z = [lambda_n (nc) | beta+ (nt) | beta- (nt) | sigma (nt)]
    Ann, Ant, Att = Jn @ Minv @ Jn.T, Jn @ Minv @ Jt.T, Jt @ Minv @ Jt.T
    E = np.eye(nt)
    P = np.zeros((nt, nc))  # tangent row -> its contact
    for k in range(nt):
        P[k, k // 2] = 1.0
    Z = np.zeros
    M = np.block([
        [Ann, Ant, -Ant, Z((nc, nt))],
        [Ant.T, Att, -Att, E],
        [-Ant.T, -Att, Att, E],
        [P * mu[None, :], -E, -E, Z((nt, nt))],
    ])
    q = np.concatenate([Jn @ vstar + bias, Jt @ vstar, -(Jt @ vstar), np.zeros(nt)])
    z = lemke(M, q)
    lam_n, lam_t = z[:nc], z[nc : nc + nt] - z[nc + nt : nc + 2 * nt]
    v = vstar + Minv @ (Jn.T @ lam_n + Jt.T @ lam_t)
    imp = np.zeros((nc, 3))
    for k, (a, b, x, n, gap, mu_k) in enumerate(contacts):
        t1, t2 = tangents(n)
        imp[k] = lam_n[k] * np.asarray(n, float) + lam_t[2 * k] * t1 + lam_t[2 * k + 1] * t2
    return v.reshape(nb, 6), imp
