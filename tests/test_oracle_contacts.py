"""Physical known-answer tests for the oracle's narrowphase + contact solver.

rest / slide-threshold / drop tests, box-box manifold shape, and MPR cross-checked against
analytic distances and against the SAT path.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from maniskill_amd.model import geom
from maniskill_amd.model.compile import ActorRecord, SceneModelBuilder, ShapeRecord
from maniskill_amd.model.scenes import cube_record, ground_record, panda_tabletop_model, table_record
from tests import oracle_backend as ob


def cube_on_table_model(gravity=(0, 0, -9.81), dt=0.01, **kw):
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(cube_record())
    return b.compile(timestep=dt, gravity=gravity, **kw)


def cube_state(px, model, N):
    r = model.row_of("cube")
    return px.cuda_rigid_body_data.torch()[r * N : (r + 1) * N]


def test_cube_rests_on_table():
    model = cube_on_table_model()
    N = 4
    px = ob.make_system(model, N)
    s = cube_state(px, model, N)
    s[:, 0] = torch.tensor([0.0, 0.05, -0.05, 0.1])
    s[:, 2] = 0.02
    yaw = torch.tensor([0.0, 0.3, 1.0, 2.0])
    s[:, 3], s[:, 6] = torch.cos(yaw / 2), torch.sin(yaw / 2)
    p0 = s[:, :3].clone()
    px.gpu_apply_all()
    px.step(30)
    cnt = px.read_internal("contact_count", model.n_pair)
    assert torch.all(cnt.sum(0) == 4)
    px.step(70)  # 1 s in all
    px.gpu_fetch_all()
    s = cube_state(px, model, N)
    assert torch.max(torch.abs(s[:, :3] - p0)) < 1e-4
    assert torch.max(torch.abs(s[:, 7:13])) < 1e-3


def test_resting_cube_goes_to_sleep_and_wakes_when_touched():
    """sleep_threshold = 0.005 (mani_skill/utils/structs/types.py:39): a cube at rest for MSSIM_WAKE_TIME = 0.4 s goes to
    sleep -- exactly zero velocity, pose frozen, out of the solver (no contact rows) -- and wakes when the user moves it,
    pushes it, or when a moving body touches it; a cube dropped on it wakes it; with sleep_threshold = 0 nothing sleeps"""
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(cube_record())
    b.add_actor(cube_record(name="cube2", p=(0.3, 0, 0.02)))
    model = b.compile()
    px = ob.make_system(model, 1)
    r1, r2 = model.row_of("cube"), model.row_of("cube2")
    px.step(39)
    assert torch.all(px.read_internal("free_wake", 2) > 0) and int(px.read_internal("contact_count", model.n_pair).sum()) == 8
    px.step(3)  # (the counter runs out in substep 41; substep 42 is the first without the bodies in the solver)
    px.gpu_fetch_all()
    rb = px.cuda_rigid_body_data.torch()
    assert torch.all(px.read_internal("free_wake", 2) == 0)
    assert torch.all(rb[[r1, r2], 7:13] == 0) and int(px.read_internal("contact_count", model.n_pair).sum()) == 0
    frozen = rb[[r1, r2], :7].clone()
    px.step(100)
    px.gpu_fetch_all()
    assert torch.equal(px.cuda_rigid_body_data.torch()[[r1, r2], :7], frozen)
    # an unchanged apply keeps them asleep; a new pose for cube2 wakes cube2 only: it drops onto cube (0.1 m above it)
    px.gpu_apply_all()
    px.step(1)
    assert torch.all(px.read_internal("free_wake", 2) == 0)
    rb = px.cuda_rigid_body_data.torch()
    rb[r2, :3] = torch.tensor([0.0, 0.0, 0.16])
    px.gpu_apply_all()
    px.step(5)
    w = px.read_internal("free_wake", 2)[:, 0]
    assert w[0] == 0 and w[1] > 0
    woke_at = None
    for i in range(40):
        px.step(1)
        if px.read_internal("free_wake", 2)[0, 0] > 0:
            woke_at = i
            break
    assert woke_at is not None  # touched by the falling (awake, not calm) cube
    px.step(200)
    px.gpu_fetch_all()
    rb = px.cuda_rigid_body_data.torch()
    assert abs(rb[r1, 2].item() - 0.02) < 2e-3 and abs(rb[r2, 2].item() - 0.06) < 5e-3  # a stack, asleep again together
    assert torch.all(px.read_internal("free_wake", 2) == 0)
    # a force wakes
    px.cuda_rigid_body_force.torch()[r1, 0] = 0.5
    px.gpu_apply_rigid_dynamic_force()
    px.step(1)
    assert px.read_internal("free_wake", 2)[0, 0] > 0
    # never with sleep_threshold = 0
    px0 = ob.make_system(cube_on_table_model(sleep_threshold=0.0), 1)
    px0.step(200)
    assert px0.read_internal("free_wake", 1)[0, 0] > 0 and int(px0.read_internal("contact_count", 3).sum()) == 4


@pytest.mark.parametrize("theta_deg,slides", [(14.0, False), (16.0, False), (17.5, True), (25.0, True)])
def test_slide_threshold_mu_0p3(theta_deg, slides):
    # tilt gravity instead of the table: slides iff tan(theta) > mu = 0.3  (theta* = 16.7 deg)
    th = np.deg2rad(theta_deg)
    g = 9.81 * np.array([np.sin(th), 0, -np.cos(th)])
    # (sleep_threshold = 0: at 17.5 degrees the cube gains speed so slowly -- 0.14 m/s^2 -- that its energy is still below
    # the reference's sleep threshold when the 0.4 s wake counter runs out; this test is about the friction cone)
    model = cube_on_table_model(gravity=tuple(g), sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    px.step(100)
    px.gpu_fetch_all()
    s = cube_state(px, model, 1)
    if slides:
        a = 9.81 * (np.sin(th) - 0.3 * np.cos(th))
        assert s[0, 0].item() > 0.5 * 0.5 * a * 1.0  # at least half of the ideal travel
        assert s[0, 0].item() < 1.2 * 0.5 * a * 1.0 + 1e-3
    else:
        assert abs(s[0, 0].item()) < 2e-3


def test_drop_no_bounce():
    model = cube_on_table_model()
    px = ob.make_system(model, 1)
    s = cube_state(px, model, 1)
    s[:, 2] = 0.12
    px.gpu_apply_all()
    zmin_after_landing = []
    for i in range(100):
        px.step(1)
        px.gpu_fetch_all()
        z = cube_state(px, model, 1)[0, 2].item()
        vz = cube_state(px, model, 1)[0, 9].item()
        if i > 20:
            zmin_after_landing.append((z, vz))
    z = np.array(zmin_after_landing)
    assert np.all(np.abs(z[-40:, 0] - 0.02) < 5e-4)  # at rest on the table
    assert z[:, 1].max() < 0.05  # restitution 0: no upward rebound


def test_stack_two_cubes_is_stable():
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(cube_record(name="c0", p=(0, 0, 0.02)))
    b.add_actor(cube_record(name="c1", p=(0.005, 0.003, 0.06)))
    model = b.compile()
    px = ob.make_system(model, 1)
    px.step(200)
    px.gpu_fetch_all()
    rb = px.cuda_rigid_body_data.torch()
    assert abs(rb[model.row_of("c1"), 2].item() - 0.06) < 1e-3
    assert abs(rb[model.row_of("c1"), 0].item() - 0.005) < 2e-3


# ----------------------------------------------------------------------------- narrowphase hooks
def _collide(lib, ta, pa, prm_a, tb, pb, prm_b, offset=0.02, force_mpr=0, va=None, vb=None):
    fn = lib.lib.mssim_ref_test_collide
    fn.restype = C.c_int
    F = C.POINTER(C.c_float)

    def arr(x, n=None):
        a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        return a, a.ctypes.data_as(F)

    a1, p1 = arr(pa)
    a2, p2 = arr(list(prm_a) + [0] * (4 - len(prm_a)))
    a3, p3 = arr(pb)
    a4, p4 = arr(list(prm_b) + [0] * (4 - len(prm_b)))
    va_ = np.zeros(3, np.float32) if va is None else np.ascontiguousarray(va, np.float32)
    vb_ = np.zeros(3, np.float32) if vb is None else np.ascontiguousarray(vb, np.float32)
    out = np.zeros(20, np.float32)
    fn(
        C.c_int(ta), p1, p2, va_.ctypes.data_as(F), C.c_int(0 if va is None else len(va_)),
        C.c_int(tb), p3, p4, vb_.ctypes.data_as(F), C.c_int(0 if vb is None else len(vb_)),
        C.c_float(offset), C.c_int(force_mpr), out.ctypes.data_as(F),
    )
    cnt = int(out[0])
    return cnt, out[1:4].astype(np.float64), out[4 : 4 + 4 * cnt].reshape(cnt, 4).astype(np.float64)


BOX, SPHERE, CONVEX = 1, 2, 5


def test_box_box_face_manifold(oracle_lib):
    # 4 cm cube resting 1 mm above a big box: 4 corner points, normal +z (from B=table to A=cube)
    cube = geom.pose([0.01, 0.02, 0.021], geom.rpy_to_quat([0, 0, 0.4]))
    table = geom.pose([0, 0, -0.5])
    cnt, n, pts = _collide(oracle_lib, BOX, cube, [0.02] * 3, BOX, table, [1, 1, 0.5])
    assert cnt == 4
    np.testing.assert_allclose(n, [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(pts[:, 3], 0.001, atol=1e-6)
    R = geom.quat_to_mat(cube[3:])
    corners = np.array([cube[:3] + R @ np.array([sx * 0.02, sy * 0.02, -0.02]) for sx in (-1, 1) for sy in (-1, 1)])
    for c in corners:
        assert np.min(np.linalg.norm(pts[:, :2] - c[:2], axis=1)) < 1e-6


def test_box_box_edge_edge(oracle_lib):
    # two cubes, edges crossing at 90 degrees (A rotated 45deg about x, B rotated 45deg about y)
    a = geom.pose([0, 0, 0.0], geom.rpy_to_quat([np.pi / 4, 0, 0]))
    h = 0.02 * np.sqrt(2)
    b = geom.pose([0, 0, -2 * h - 0.003], geom.rpy_to_quat([0, np.pi / 4, 0]))
    cnt, n, pts = _collide(oracle_lib, BOX, a, [0.02] * 3, BOX, b, [0.02] * 3)
    assert cnt == 1
    np.testing.assert_allclose(n, [0, 0, 1], atol=1e-5)
    assert abs(pts[0, 3] - 0.003) < 1e-5
    np.testing.assert_allclose(pts[0, :3], [0, 0, -h - 0.0015], atol=1e-5)


def test_mpr_sphere_sphere_and_sphere_box(oracle_lib):
    for d in (0.05, 0.061, 0.069):
        cnt, n, pts = _collide(oracle_lib, SPHERE, geom.pose([0, 0, d]), [0.03], SPHERE, geom.pose(), [0.03])
        assert cnt == 1
        np.testing.assert_allclose(n, [0, 0, 1], atol=1e-4)
        assert abs(pts[0, 3] - (d - 0.06)) < 1e-4
    cnt, _, _ = _collide(oracle_lib, SPHERE, geom.pose([0, 0, 0.09]), [0.03], SPHERE, geom.pose(), [0.03])
    assert cnt == 0
    # sphere above a box face
    cnt, n, pts = _collide(oracle_lib, BOX, geom.pose([0, 0, -0.5]), [1, 1, 0.5], SPHERE, geom.pose([0.1, 0.2, 0.035]), [0.03])
    assert cnt == 1
    np.testing.assert_allclose(n, [0, 0, -1], atol=1e-4)  # from B (sphere) to A (box)
    assert abs(pts[0, 3] - 0.005) < 1e-4


def test_mpr_agrees_with_sat_on_box_faces(oracle_lib):
    rng = np.random.default_rng(3)
    for _ in range(50):
        yaw = rng.uniform(-np.pi, np.pi)
        tilt = rng.uniform(-0.05, 0.05, size=2)
        gap = rng.uniform(-0.004, 0.015)
        qa = geom.rpy_to_quat([tilt[0], tilt[1], yaw])
        R = geom.quat_to_mat(qa)
        # lowest corner of the cube sits `gap` above z = 0
        low = min((R @ np.array([sx, sy, sz]) * 0.02)[2] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1))
        a = geom.pose([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), gap - low], qa)
        b = geom.pose([0, 0, -0.5])
        c1, n1, p1 = _collide(oracle_lib, BOX, a, [0.02] * 3, BOX, b, [1, 1, 0.5])
        c2, n2, p2 = _collide(oracle_lib, BOX, a, [0.02] * 3, BOX, b, [1, 1, 0.5], force_mpr=1)
        assert c1 >= 1 and c2 == 1
        np.testing.assert_allclose(n1, n2, atol=2e-3)
        assert abs(p1[:, 3].min() - p2[0, 3]) < 2e-4
        assert abs(p2[0, 3] - gap) < 2e-4


def test_convex_hull_vs_box_matches_box_box(oracle_lib):
    # a hull made of a cube's 8 corners must behave like the box
    verts = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float32) * 0.02
    rng = np.random.default_rng(5)
    for _ in range(20):
        q = geom.rpy_to_quat(rng.uniform(-np.pi, np.pi, size=3))
        R = geom.quat_to_mat(q)
        low = (verts.astype(np.float64) @ R.T)[:, 2].min()
        gap = rng.uniform(-0.003, 0.015)
        pose_h = geom.pose([0.05, -0.03, gap - low], q)
        cnt, n, pts = _collide(oracle_lib, BOX, geom.pose([0, 0, -0.5]), [1, 1, 0.5], CONVEX, pose_h, [0, 0, 0], vb=verts)
        assert cnt == 1
        np.testing.assert_allclose(n, [0, 0, -1], atol=5e-3)
        assert abs(pts[0, 3] - gap) < 3e-4


def test_panda_grasp_holds_cube():
    """close the gripper on the cube, lift: the cube must follow (friction 2.0 pads)."""
    model = panda_tabletop_model()
    N = 1
    px = ob.make_system(model, N)
    rest = np.array([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04], dtype=np.float32)
    tcp = model.link_names.index("panda_hand_tcp")

    def tcp_pos(qv):
        px.cuda_articulation_qpos.torch()[:] = qv
        px.gpu_apply_articulation_qpos()
        px.gpu_update_articulation_kinematics()
        px.gpu_fetch_articulation_link_pose()
        return px.cuda_rigid_body_data.torch()[tcp * N, :3].clone()

    def ik(q0, target):
        # a few Gauss-Newton steps on joints 2, 4, 6 (keeps the hand pointing down approximately)
        q = q0.clone()
        idx = [1, 3, 5]
        for _ in range(30):
            p = tcp_pos(q)
            J = torch.zeros(3, 3)
            for k, j in enumerate(idx):
                dq = q.clone()
                dq[0, j] += 1e-4
                J[:, k] = (tcp_pos(dq) - p) / 1e-4
            step = torch.linalg.solve(J.T @ J + 1e-6 * torch.eye(3), J.T @ (target - p))
            for k, j in enumerate(idx):
                q[0, j] += step[k]
        assert torch.norm(tcp_pos(q) - target) < 1e-3
        return q

    q = ik(torch.from_numpy(rest).clone()[None], torch.tensor([0.0, 0.0, 0.02]))
    q_lift = ik(q, torch.tensor([0.0, 0.0, 0.15]))
    px.cuda_articulation_qpos.torch()[:] = q
    px.cuda_articulation_target_qpos.torch()[:] = q
    px.gpu_apply_all()
    # close
    tq = q.clone()
    tq[0, 7:] = -0.01
    px.cuda_articulation_target_qpos.torch()[:] = tq
    px.gpu_apply_articulation_target_position()
    px.step(50)
    px.gpu_fetch_all()
    fingers = px.cuda_articulation_qpos.torch()[0, 7:]
    # (warm-started multipliers: the saturated 100 N squeeze converges, the pads do not sink into the 20 mm half-width cube;
    # a cold-started sweep leaves them 3.3 mm inside, scripts/tgs_vs_pgs.py)
    assert torch.all(fingers > 0.0198) and torch.all(fingers < 0.0203), fingers
    steps = 100
    for i in range(steps):
        a = (i + 1) / steps
        tq = q * (1 - a) + q_lift * a
        tq[0, 7:] = -0.01
        px.cuda_articulation_target_qpos.torch()[:] = tq
        px.gpu_apply_articulation_target_position()
        px.step(1)
    px.step(50)
    px.gpu_fetch_all()
    cube = px.cuda_rigid_body_data.torch()[model.row_of("cube") * N]
    assert cube[2].item() > 0.12, cube


def test_fast_spinning_thin_body_stays_bounded():
    """a thin box (a peg) thrown into a 2000 rad/s spin about a non-principal axis, far from everything: the explicitly
    integrated gyroscopic term would pump the spin up to overflow within a few hundred substeps; with the angular
    velocity limit (MSSIM_MAX_ANGULAR_VELOCITY = 100 rad/s, PhysX's default) the state stays finite, the quaternion
    unit and the spin bounded"""
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("peg", "dynamic", [ShapeRecord("box", geom.pose(), half_size=np.array([0.06, 0.02, 0.02]))], initial_pose=geom.pose((0, 0, 5.0))))
    model = b.compile(timestep=0.01, gravity=(0, 0, 0))
    px = ob.make_system(model, 1)
    r = model.row_of("peg")
    s = px.cuda_rigid_body_data.torch()[r : r + 1]
    s[:, 10:13] = torch.tensor([1200.0, 1500.0, -600.0])
    px.gpu_apply_all()
    for _ in range(40):
        px.step(10)
        px.gpu_fetch_all()
        s = px.cuda_rigid_body_data.torch()[r : r + 1]
        assert torch.isfinite(s).all()
        assert abs(float(s[0, 3:7].norm()) - 1.0) < 1e-4
        assert float(s[0, 10:13].norm()) < 1.5 * 100.0  # the limit bounds what a substep starts from; one explicit step may overshoot a little
    assert float(s[0, :3].sub(torch.tensor([0.0, 0.0, 5.0])).norm()) < 1e-4  # no force: the centre does not move


def _sphere_on_ground(patch_radius):
    b = SceneModelBuilder()
    b.add_actor(ground_record(altitude=0.0))
    b.add_actor(ActorRecord("ball", "dynamic", [ShapeRecord("sphere", geom.pose(), radius=0.05, patch_radius=patch_radius, min_patch_radius=patch_radius)],
                            initial_pose=geom.pose([0, 0, 0.05])))
    return b.compile()


def test_torsional_friction_spins_a_ball_down():
    """A ball spinning about the vertical on a plane touches it in one point on the spin axis: the tangential friction rows
    have no lever arm, only the torsional row of a shape with a patch radius r (agents/robots/panda/panda.py:24-31) brakes
    the spin, at the constant rate mu r m g / I = mu r g / (0.4 R^2) until it stops; without a patch radius it spins on."""
    R, mu, g = 0.05, 0.3, 9.81
    for r_patch in (0.0, 0.01):
        model = _sphere_on_ground(r_patch)
        px = ob.make_system(model, 1)
        row = model.row_of("ball")
        s = px.cuda_rigid_body_data.torch()[row : row + 1]
        s[:, 12] = 10.0  # spin about z
        px.gpu_apply_all()
        px.step(10)
        px.gpu_fetch_all()
        w10 = px.cuda_rigid_body_data.torch()[row, 12].item()
        px.step(60)
        px.gpu_fetch_all()
        w70 = px.cuda_rigid_body_data.torch()[row, 12].item()
        z = px.cuda_rigid_body_data.torch()[row, 2].item()
        assert abs(z - R) < 1e-3
        if r_patch == 0.0:
            assert abs(w10 - 10.0) < 1e-3 and abs(w70 - 10.0) < 1e-2
        else:
            alpha = mu * r_patch * g / (0.4 * R * R)  # 29.4 rad/s^2
            assert abs(w10 - (10.0 - alpha * 0.1)) < 0.05 * alpha * 0.1 + 0.02, w10
            assert abs(w70) < 1e-3, w70  # stopped (after 0.34 s) and held: the row is bilateral inside its bound


def test_contact_patches_cut_a_compound_body_to_four_points():
    """two boxes of ONE body side by side on the table give 8 manifold points with one normal: the body-pair patch keeps
    4 (the extreme ones), the body rests exactly as before; a body whose two boxes touch a floor and a wall keeps both
    patches (normals 90 degrees apart)"""
    half = np.array([0.02, 0.02, 0.02])
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(ActorRecord("twin", "dynamic", [ShapeRecord("box", geom.pose([-0.03, 0, 0]), half_size=half), ShapeRecord("box", geom.pose([0.03, 0, 0]), half_size=half)],
                            initial_pose=geom.pose([0, 0, 0.02])))
    model = b.compile()
    px = ob.make_system(model, 1)
    px.step(30)
    px.gpu_fetch_all()
    row = model.row_of("twin")
    s = px.cuda_rigid_body_data.torch()[row]
    assert abs(s[2].item() - 0.02) < 1e-4 and torch.max(torch.abs(s[7:13])) < 1e-3
    assert int(px.read_internal("raw_contact_count", 1)[0, 0]) == 8
    cnt = px.read_internal("contact_count", model.n_pair)
    assert int(cnt.sum()) == 4
    # the four survivors span the whole footprint: x from -0.05 to 0.05 (not one box's 4 corners)
    assert (cnt > 0).sum() == 2


def _slab_on_table(as_hull=True):
    """a 0.2 x 0.12 x 0.04 m slab on the table: as a convex hull (8 vertices -> the generic convex path: one point per
    MPR query, persistent manifold) or as a box (box-box manifold, 4 points at once)"""
    half = np.array([0.1, 0.06, 0.02])
    if as_hull:
        verts = np.array([[sx * half[0], sy * half[1], sz * half[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
        shape = ShapeRecord("convex", geom.pose(), vertices=verts)
    else:
        shape = ShapeRecord("box", geom.pose(), half_size=half)
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(ActorRecord("slab", "dynamic", [shape], initial_pose=geom.pose([0, 0, 0.02])))
    return b.compile()


def test_persistent_manifold_lets_a_hull_rest_without_rocking():
    """enable_pcm (mani_skill/utils/structs/types.py:44): a convex hull lying on the table gets ONE point per full
    query; while the pair is not moving the persistent manifold is grown by one tilted query per substep (3-4 points after
    as many substeps), after which the slab rests flat like its box twin -- no rocking, no queries while nothing
    moves -- and goes to sleep"""
    model = _slab_on_table(True)
    px = ob.make_system(model, 1)
    row = model.row_of("slab")
    counts, queries, tilt = [], [], []
    for i in range(30):
        px.step(1)
        px.gpu_fetch_all()
        counts.append(int(px.read_internal("contact_count", model.n_pair).sum()))
        queries.append(int(px.read_internal("mpr_queries", 1)[0, 0]))
        q = px.cuda_rigid_body_data.torch()[row, 3:7]
        tilt.append(float(2 * torch.acos(torch.clamp(q[0].abs(), max=1.0))))
    assert counts[0] == 1 and counts[5] >= 3 and min(counts[5:]) >= 3, counts
    assert sum(queries[:6]) >= 3 and sum(queries[12:]) == 0, queries  # a resting manifold is refreshed, not regenerated
    assert max(tilt) < np.deg2rad(2.0) and max(tilt[20:]) < np.deg2rad(0.3), np.rad2deg(tilt)
    s = px.cuda_rigid_body_data.torch()[row]
    assert abs(s[2].item() - 0.02) < 5e-4 and torch.max(torch.abs(s[7:13])) < 5e-3
    px.step(60)
    assert px.read_internal("free_wake", 1)[0, 0] == 0  # at rest long enough: asleep
    # pushed sideways (a force for one substep at a time), the manifold follows: points drift out, new ones come in
    for _ in range(20):
        px.cuda_rigid_body_force.torch()[row, 0] = 6.0  # (mu m g = 2.8 N)
        px.gpu_apply_rigid_dynamic_force()
        px.step(1)
    px.step(50)
    px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[row]
    assert s[0].item() > 0.03 and abs(s[2].item() - 0.02) < 1e-3
    q = s[3:7]
    assert float(2 * torch.acos(torch.clamp(q[0].abs(), max=1.0))) < np.deg2rad(3.0)


def _per_env_hull_model(N, n_kinds=4, seed=0):
    """synthetic object set (per-env object sets, no downloaded assets): `n_kinds` random convex polyhedra
    (hulls of points on ellipsoids with different axes, 12-40 vertices), env i carries kind i % n_kinds"""
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(seed)
    kinds = []
    for k in range(n_kinds):
        axes = np.array([0.03 + 0.02 * k, 0.025 + 0.01 * ((k * 3) % 4), 0.015 + 0.005 * k])
        pts = rng.normal(size=(14 + 10 * k, 3))
        pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * axes
        verts = pts[ConvexHull(pts).vertices]
        kinds.append(np.ascontiguousarray(verts))
    env_shapes = [[ShapeRecord("convex", geom.pose(), vertices=kinds[i % n_kinds])] for i in range(N)]
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(ActorRecord("obj", "dynamic", list(env_shapes[0]), initial_pose=geom.pose([0, 0, 0.08]), env_shapes=env_shapes))
    return b.compile(num_envs=N), kinds


def test_per_env_hulls_each_env_simulates_its_own_object():
    """a merged actor whose convex hull differs per env (the reference builds one object per sub-scene and merges the
    views, utils/structs/actor.py:99-126; PickSingleYCB-style tasks): every env drops ITS polyhedron on the table. Envs
    with the same hull behave identically, envs with different hulls come to rest at different heights, each with its lowest
    vertices on the table top (z = 0)."""
    N, K = 8, 4
    model, kinds = _per_env_hull_model(N, K)
    px = ob.make_system(model, N)
    row = model.row_of("obj")
    px.step(150)
    px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[row * N : (row + 1) * N].double().numpy()
    assert np.isfinite(s).all()
    for i in range(K):
        assert np.allclose(s[i], s[i + K], atol=1e-9), i  # same hull, same initial state: the same trajectory
    z = s[:, 2]
    assert len(set(np.round(z[:K], 4))) == K, z  # four different objects, four different rest heights
    from maniskill_amd.utils.geometry.rotation_conversions import quaternion_to_matrix

    R = quaternion_to_matrix(torch.from_numpy(s[:, 3:7])).numpy()
    for i in range(N):
        low = (kinds[i % K] @ R[i].T + s[i, :3])[:, 2].min()
        assert abs(low) < 1e-3, (i, low)  # resting ON the table: the lowest vertex within a millimetre of its top
    assert np.abs(s[:, 7:13]).max() < 0.05


def _mixed_object_model(N):
    """per-env object sets with different shape types and counts (Actor.merge over heterogeneous sub-scenes,
    utils/structs/actor.py:99-126): env i carries kind i % 5 -- a box, a sphere, a hull, a compound of two boxes, nothing"""
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(2)
    pts = rng.normal(size=(24, 3))
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * np.array([0.04, 0.03, 0.025])
    hull = np.ascontiguousarray(pts[ConvexHull(pts).vertices])
    kinds = [
        [ShapeRecord("box", geom.pose(), half_size=np.array([0.03, 0.02, 0.02]))],
        [ShapeRecord("sphere", geom.pose(), radius=0.03)],
        [ShapeRecord("convex", geom.pose(), vertices=hull)],
        [ShapeRecord("box", geom.pose([0, 0, 0]), half_size=np.array([0.04, 0.01, 0.01])),
         ShapeRecord("box", geom.pose([0, 0, 0.02]), half_size=np.array([0.01, 0.01, 0.01]))],
        [],
    ]
    env_shapes = [list(kinds[i % 5]) for i in range(N)]
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(ActorRecord("obj", "dynamic", list(env_shapes[0]), initial_pose=geom.pose([0, 0, 0.08]), env_shapes=env_shapes))
    return b.compile(num_envs=N), hull


def test_per_env_object_sets_with_different_shape_types():
    """every env simulates the object IT carries: a box rests at its half height, a sphere at its radius, the hull on its
    lowest vertices, the two-box compound on its long box; in the envs that carry nothing the body does not exist (mass 0):
    it keeps the pose it was given, is never awake and touches nothing"""
    N = 10
    model, hull = _mixed_object_model(N)
    px = ob.make_system(model, N)
    row = model.row_of("obj")
    px.step(200)
    px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[row * N : (row + 1) * N].double().numpy()
    assert np.isfinite(s).all()
    for i in range(5):
        assert np.allclose(s[i], s[i + 5], atol=1e-9), i
    assert abs(s[0, 2] - 0.02) < 2e-4, s[0]          # box: half height
    assert abs(s[1, 2] - 0.03) < 2e-4, s[1]          # sphere: radius
    from maniskill_amd.utils.geometry.rotation_conversions import quaternion_to_matrix

    R = quaternion_to_matrix(torch.from_numpy(s[2:3, 3:7])).numpy()[0]
    assert abs((hull @ R.T + s[2, :3])[:, 2].min()) < 1e-3  # hull: lowest vertices on the table
    assert abs(s[3, 2] - 0.01) < 1e-3 or s[3, 2] < 0.045    # compound: lying on the long box (possibly tipped over)
    assert np.allclose(s[4, :3], [0, 0, 0.08]) and np.allclose(s[4, 7:], 0)  # nothing there: the body stays where it was put
    assert px.read_internal("free_wake", 1).reshape(-1)[4] <= 0
    assert px.overflow_count() == 0


def _grid_mesh(n=8, size=0.6, tilt_deg=0.0, height=None):
    """an n x n grid of quads (2 n^2 triangles) of side `size`, tilted about y, optionally with a height field on top"""
    xs, ys = np.meshgrid(np.linspace(-size / 2, size / 2, n + 1), np.linspace(-size / 2, size / 2, n + 1), indexing="ij")
    z = np.zeros_like(xs) if height is None else height(xs, ys)
    V = np.stack([xs.ravel(), ys.ravel(), z.ravel()], 1)
    F = []
    for i in range(n):
        for j in range(n):
            a = i * (n + 1) + j
            F += [[a, a + n + 1, a + 1], [a + 1, a + n + 1, a + n + 2]]
    t = np.deg2rad(tilt_deg)
    R = np.array([[np.cos(t), 0, np.sin(t)], [0, 1, 0], [-np.sin(t), 0, np.cos(t)]])
    return V @ R.T, np.asarray(F)


def _mesh_scene(tilt_deg=0.0, cube_z=0.03, height=None, n=8):
    V, F = _grid_mesh(n=n, tilt_deg=tilt_deg, height=height)
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("terrain", "static", [ShapeRecord("trimesh", geom.pose(), vertices=V, triangles=F)], initial_pose=geom.pose([0, 0, 0])))
    b.add_actor(cube_record(p=(0, 0, cube_z)))
    return b.compile(sleep_threshold=0.0)  # (awake throughout: the tests look at its contacts)


def test_triangle_mesh_floor_and_slopes():
    """static triangle meshes (the reference's nonconvex collision, actor_builder.py:136-150; MSSIM_SHAPE_TRIMESH): a cube
    dropped on a flat 128-triangle grid rests at its half height like on a box (the contacts of the coplanar triangles
    merge into one 4-point patch); on a mesh tilted by 10 degrees it stays (tan 10 < mu = 0.3), tilted by 25 degrees it slides
    with g (sin - mu cos)"""
    model = _mesh_scene()
    px = ob.make_system(model, 1)
    row = model.row_of("cube")
    px.step(100)
    px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[row]
    assert abs(s[2].item() - 0.02) < 2e-4 and s[7:10].abs().max() < 2e-3 and s[10:13].abs().max() < 3e-2, s  # (manifolds rebuilt every substep: a residual rocking of ~1e-2 rad/s)
    assert px.read_internal("contact_count", model.n_pair).sum().item() == 4  # one patch of four points
    assert px.overflow_count() == 0

    for tilt, slides in ((10.0, False), (25.0, True)):
        t = np.deg2rad(tilt)
        model = _mesh_scene(tilt_deg=tilt)
        px = ob.make_system(model, 1)
        rb = px.cuda_rigid_body_data.torch()
        # the cube lying on the slope: centre 0.02 above the surface along its normal (sin t, 0, cos t), turned with it
        rb[row, :3] = torch.tensor([0.02 * np.sin(t), 0.0, 0.02 * np.cos(t)], dtype=rb.dtype)
        rb[row, 3:7] = torch.tensor([np.cos(t / 2), 0.0, np.sin(t / 2), 0.0], dtype=rb.dtype)
        px.gpu_apply_all()
        px.wake_all()
        px.step(30)
        px.gpu_fetch_all()
        v = px.cuda_rigid_body_data.torch()[row, 7:10].double().numpy()
        down = np.array([np.cos(t), 0.0, -np.sin(t)])  # downhill direction of a surface whose normal is (sin t, 0, cos t)
        if slides:
            expect = 9.81 * (np.sin(t) - 0.3 * np.cos(t)) * 0.3
            assert abs(v @ down - expect) < 0.05 * expect, (v, expect)
        else:
            assert np.linalg.norm(v) < 2e-3, v


def test_triangle_mesh_terrain_holds_a_rolling_ball():
    """a ball released on the wall of a paraboloid bowl (512 triangles) rolls down, through the bottom and up the other
    side: it stays on the surface all along -- never sinks in, never trips on an inner edge: its energy only decays"""
    height = lambda x, y: 0.8 * (x * x + y * y)
    V, F = _grid_mesh(n=16, height=height)
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("terrain", "static", [ShapeRecord("trimesh", geom.pose(), vertices=V, triangles=F)]))
    b.add_actor(ActorRecord("ball", "dynamic", [ShapeRecord("sphere", geom.pose(), radius=0.03)], initial_pose=geom.pose([0.18, 0.1, height(0.18, 0.1) + 0.04])))
    model = b.compile(sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    row = model.row_of("ball")
    energy = []
    for _ in range(80):
        px.step(5)
        px.gpu_fetch_all()
        s = px.cuda_rigid_body_data.torch()[row].double().numpy()
        p = s[:3]
        assert p[2] - height(p[0], p[1]) > 0.03 * 0.9, p  # on the surface (vertical clearance >= radius on a slope)
        assert abs(p[0]) < 0.28 and abs(p[1]) < 0.28, p
        energy.append(9.81 * p[2] + 0.5 * (s[7:10] @ s[7:10]) + 0.5 * 0.4 * 0.03**2 * (s[10:13] @ s[10:13]))
    assert max(np.diff(energy)) < 2e-3 * 9.81  # no energy is gained at the edges between triangles (2 mm of height)
    assert energy[-1] < energy[0]  # (there is no rolling resistance: the ball keeps swinging through the bowl, a little lower each time)
    assert px.overflow_count() == 0


def test_hull_comes_to_rest_on_a_triangle_mesh():
    """a tumbling convex polyhedron dropped on a flat triangle mesh must end up lying ON it: within the contact offset a hull
    has many vertices above the plane, and before MSSIM_TRI_SLACK kept them out, those speculative points crowded the
    load-carrying ones out of the 4-point patch and the hull sank a centimetre into the mesh"""
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(7)
    pts = rng.normal(size=(24, 3))
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * np.array([0.035, 0.025, 0.02])
    hull = np.ascontiguousarray(pts[ConvexHull(pts).vertices])
    V, F = _grid_mesh(n=8)
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("terrain", "static", [ShapeRecord("trimesh", geom.pose(), vertices=V, triangles=F)]))
    b.add_actor(ActorRecord("rock", "dynamic", [ShapeRecord("convex", geom.pose(), vertices=hull)], initial_pose=geom.pose([0.03, -0.02, 0.1], [0.8, 0.3, -0.4, 0.33166])))
    model = b.compile(sleep_threshold=0.0)
    px = ob.make_system(model, 1)
    row = model.row_of("rock")
    from maniskill_amd.utils.geometry.rotation_conversions import quaternion_to_matrix

    worst = 1.0
    for _ in range(150):
        px.step(1)
        px.gpu_fetch_all()
        s = px.cuda_rigid_body_data.torch()[row].double()
        R = quaternion_to_matrix(s[3:7][None] / s[3:7].norm()).numpy()[0]
        worst = min(worst, float((hull @ R.T + s[:3].numpy())[:, 2].min()))
    assert worst > -2.5e-3, worst  # (the landing at 1.3 m/s dips 1-2 mm)
    low = float((hull @ R.T + s[:3].numpy())[:, 2].min())
    assert abs(low) < 1e-3 and float(s[7:10].abs().max()) < 0.02, (low, s)
    assert px.overflow_count() == 0


def _bar_on_fine_mesh():
    """a 0.30 x 0.036 x 0.04 bar lying along one row of a 5 cm grid: within the contact offset (2 cm) of its oriented box lie
    3 rows x 8 cells = 48 triangles -- more than MSSIM_MAX_TRI_HITS; within a quarter of it only its own row's 16"""
    V, F = _grid_mesh(n=12, size=0.6)
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("terrain", "static", [ShapeRecord("trimesh", geom.pose(), vertices=V, triangles=F)], initial_pose=geom.pose([0, 0, 0])))
    b.add_actor(ActorRecord("bar", "dynamic", [ShapeRecord("box", geom.pose(), half_size=np.array([0.15, 0.018, 0.02]))], initial_pose=geom.pose([0, 0.025, 0.03])))
    model = b.compile(sleep_threshold=0.0)
    lo, hi = V[F].min(1), V[F].max(1)
    c, h = np.array([0, 0.025, 0.02]), np.array([0.15, 0.018, 0.02])

    def in_range(r):
        return int(np.all((lo - (c + h) <= r) & ((c - h) - hi <= r), axis=1).sum())

    assert in_range(0.02) == 48 and in_range(0.01) == 48 and in_range(0.005) == 16
    return model


def test_triangle_search_range_narrows_before_anything_is_dropped():
    """MSSIM_TRI_RANGE_STEPS: more triangles within the contact offset than one shape pair holds -- the search is repeated
    with half, then a quarter of the range, which keeps the row the bar lies on; nothing is reported, the bar rests on its
    four corners (plus the mesh vertices under it) at its half height"""
    model = _bar_on_fine_mesh()
    px = ob.make_system(model, 1)
    row = model.row_of("bar")
    px.step(60)
    px.gpu_fetch_all()
    s = px.cuda_rigid_body_data.torch()[row]
    assert abs(s[2].item() - 0.02) < 2e-4 and s[7:13].abs().max() < 2e-2, s
    assert px.read_internal("contact_count", model.n_pair).sum().item() >= 4
    assert px.overflow_count() == 0


def kinematic_platform_scene():
    """a kinematic slab carrying a cube, a second kinematic block beside the cube, no static support anywhere near"""
    b = SceneModelBuilder()
    b.add_actor(ActorRecord("slab", "kinematic", [ShapeRecord("box", geom.pose(), half_size=np.array([0.2, 0.2, 0.01]))], initial_pose=geom.pose([0, 0, -0.01])))
    b.add_actor(ActorRecord("pusher", "kinematic", [ShapeRecord("box", geom.pose(), half_size=np.array([0.02, 0.05, 0.02]))], initial_pose=geom.pose([-0.1, 0, 0.02])))
    b.add_actor(cube_record())
    return b.compile()


def check_kinematic_bodies_wake_sleepers(px, model, N=1):
    """the advisor's round-2 finding: a sleeping body must not stay asleep when a kinematic body is moved into it (it used to
    pass straight through: the manifolds of a body that stays asleep are dropped) or away from under it (it used to float)"""
    rc, rs, rp = model.row_of("cube"), model.row_of("slab"), model.row_of("pusher")
    px.step(60)  # 0.6 s: the cube has been asleep on the slab for a while
    px.gpu_fetch_all()
    rb = px.cuda_rigid_body_data.torch().reshape(model.n_rows, N, 13)
    assert float(px.read_internal("free_wake", 1).max()) == 0 and float(rb[rc, :, 7:].abs().max()) == 0
    # an apply that changes nothing keeps it asleep
    px.gpu_apply_all()
    px.step(2)
    assert float(px.read_internal("free_wake", 1).max()) == 0
    # the pusher is moved 1 cm per control step towards and through the cube's place (a kinematic body has no velocity here:
    # what it overlaps is pushed out by the penetration bias): the cube is shoved along, it is not passed through
    for k in range(16):
        rb[rp, :, 0] = -0.1 + 0.01 * (k + 1)
        px.gpu_apply_all()
        px.step(5)
        px.gpu_fetch_all()
    assert float(rb[rc, :, 0].min()) > 0.085, rb[rc, :, :3]               # ahead of the block, whose front face is at x = 0.08
    assert float(rb[rc, :, 2].min()) > 0.018                              # still on the slab
    # let it come to rest and fall asleep again, then take the slab away from under it: it wakes and falls
    px.step(80)
    assert float(px.read_internal("free_wake", 1).max()) == 0
    px.gpu_fetch_all()
    rb[rs, :, 2] = -0.5
    px.gpu_apply_all()
    px.step(20)  # 0.2 s of free fall: 0.196 m
    px.gpu_fetch_all()
    z = rb[rc, :, 2]
    assert float(z.max()) < 0.02 - 0.15 and float(px.read_internal("free_wake", 1).min()) > 0, z


def test_kinematic_bodies_wake_sleepers():
    model = kinematic_platform_scene()
    check_kinematic_bodies_wake_sleepers(ob.make_system(model, 1), model)


def test_wake_envs_resets_the_hidden_state_of_the_listed_envs_only():
    """`mssim_wake_envs` (what BaseEnv.reset calls for the envs it resets): sleep counters re-armed, manifold cache and
    warm-start multipliers dropped -- for those envs, not for the others"""
    b = SceneModelBuilder()
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(cube_record())
    model = b.compile()
    px = ob.make_system(model, 3)
    px.step(60)
    assert float(px.read_internal("free_wake", 1).max()) == 0
    px.wake_envs(torch.tensor([1]))
    w = px.read_internal("free_wake", 1)[0]
    assert w[0] == 0 and w[2] == 0 and abs(float(w[1]) - 0.4) < 1e-6
    px.step(1)
    cnt = px.read_internal("contact_count", model.n_pair).sum(0)
    assert cnt.tolist() == [0.0, 4.0, 0.0]  # only env 1's cube is back in the solver
