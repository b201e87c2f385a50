"""GPU parity for models with MORE THAN 16 velocity components per env (two 16-lane rows per env in the control-step
kernel, `k_solve16<.., NR = 2>`): the Panda with two free bodies (9 + 12 = 21 components), HIP f32 vs the f64 oracle on
identical seeded inputs. The reference builds such scenes routinely -- several dynamic actors next to the robot
(utils/scene_builder/replicacad/scene_builder.py:177-185, envs/tasks/tabletop/stack_cube.py) -- and PhysX has no
per-env component limit.

Tolerances as in tests/test_gpu_parity.py: one substep from identical state |dq| <= 1e-5 rad, |dqvel| <= 5e-4 for envs
with <= 8 contact points; free-body position <= 2e-5 m, velocity <= 2e-3; contact counts equal in >= 99 % of envs;
every env -- also the ones whose counts differ -- within 2 mm / 0.02 rad of the oracle after the substep.
"""
import numpy as np
import pytest
import torch

from maniskill_amd.model import geom
from maniskill_amd.model.compile import ActorRecord, SceneModelBuilder, ShapeRecord
from maniskill_amd.model.scenes import cube_record, ground_record, panda_record, table_record
from maniskill_amd.physx.system import MssimSystem
from tests import oracle_backend as ob

pytestmark = pytest.mark.gpu

REST = torch.tensor([0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04])


def two_body_model(second="cube", with_robot=True):
    b = SceneModelBuilder()
    if with_robot:
        b.set_articulation(panda_record())
    b.add_actor(table_record())
    b.add_actor(ground_record())
    b.add_actor(cube_record(name="cubeA", p=(0, 0, 0.02)))
    if second == "cube":
        b.add_actor(cube_record(name="cubeB", half_size=0.025, p=(0.1, 0, 0.025)))
    else:  # a compound body: two boxes forming an L (multi-shape free body)
        shapes = [ShapeRecord("box", geom.pose([0, 0, 0]), half_size=np.array([0.04, 0.015, 0.015])),
                  ShapeRecord("box", geom.pose([0.025, 0.03, 0]), half_size=np.array([0.015, 0.015, 0.015]))]
        b.add_actor(ActorRecord("cubeB", "dynamic", shapes, initial_pose=geom.pose([0.1, 0, 0.015])))
    b.add_actor(ActorRecord("goal_site", "kinematic", [], initial_pose=geom.pose()))
    return b.compile()


def make_pair(model, N):
    gpu = MssimSystem(device="cuda:0")
    gpu.gpu_init(model, N)
    return gpu, ob.make_system(model, N, precision="f64")


def random_state(N, seed, stack_frac=0.3):
    g = torch.Generator().manual_seed(seed)
    q = REST + 0.3 * (2 * torch.rand(N, 9, generator=g) - 1)
    q[:, 3] = torch.clamp(q[:, 3], -3.0, -0.1)
    q[:, 5] = torch.clamp(q[:, 5], 0.0, 3.7)
    q[:, 7:] = 0.04 * torch.rand(N, 2, generator=g)
    qd = 0.5 * (2 * torch.rand(N, 9, generator=g) - 1)
    qd[:, 7:] *= 0.05
    tq = q.clone()
    tq[:, :7] += 0.1 * (2 * torch.rand(N, 7, generator=g) - 1)
    tq[:, 7:] = (0.05 * torch.rand(N, 1, generator=g) - 0.01).expand(N, 2)

    def body(half, xy_lo, xy_hi):
        s = torch.zeros(N, 13)
        s[:, :2] = xy_lo + (xy_hi - xy_lo) * torch.rand(N, 2, generator=g)
        s[:, 2] = half
        yaw = 2 * np.pi * torch.rand(N, generator=g)
        s[:, 3], s[:, 6] = torch.cos(yaw / 2), torch.sin(yaw / 2)
        s[:, 7:10] = 0.05 * (2 * torch.rand(N, 3, generator=g) - 1)
        s[:, 10:13] = 0.2 * (2 * torch.rand(N, 3, generator=g) - 1)
        return s

    a = body(0.02, -0.1, 0.1)
    b = body(0.025, -0.1, 0.1)
    # a share of the envs: B stacked on A (slightly off centre), the rest side by side or overlapping in plan -> pushed apart
    stacked = torch.rand(N, generator=g) < stack_frac
    b[stacked, :2] = a[stacked, :2] + 0.01 * (2 * torch.rand(int(stacked.sum()), 2, generator=g) - 1)
    b[stacked, 2] = 0.04 + 0.025
    apart = ~stacked
    d = b[apart, :2] - a[apart, :2]
    near = d.norm(dim=1) < 0.07
    d[near] = 0.08 * torch.nn.functional.normalize(d[near] + 1e-3, dim=1)
    b[apart, :2] = a[apart, :2] + d
    return q, qd, tq, a, b


def set_state(px, model, N, q, qd, tq, a, b):
    dev = px.device
    px.cuda_articulation_qpos.torch()[:] = q.to(dev)
    px.cuda_articulation_qvel.torch()[:] = qd.to(dev)
    px.cuda_articulation_target_qpos.torch()[:] = tq.to(dev)
    rb = px.cuda_rigid_body_data.torch()
    ra, rbb = model.row_of("cubeA"), model.row_of("cubeB")
    rb[ra * N : (ra + 1) * N] = a.to(dev)
    rb[rbb * N : (rbb + 1) * N] = b.to(dev)
    px.gpu_apply_all()


def get_state(px, model, N):
    px.gpu_fetch_all()
    return dict(
        q=px.cuda_articulation_qpos.torch().cpu().clone(),
        qd=px.cuda_articulation_qvel.torch().cpu().clone(),
        rb=px.cuda_rigid_body_data.torch().cpu().clone().reshape(model.n_rows, N, 13),
        cnt=px.read_internal("contact_count", max(model.n_pair, 1)).cpu().clone(),
    )


def overflow_bits(px, N):
    """per env the MSSIM_OVERFLOW_* bits since the last overflow_count (include/mssim.h)"""
    return px.read_internal("overflow", 1).cpu().reshape(-1)[:N].to(torch.int64)


def assert_no_overflow_beyond_oracle(gpu, cpu, N, max_envs=None):
    """a capacity exceeded on the HIP side must be exceeded in the oracle too (same tables), and rarely"""
    og, oc = overflow_bits(gpu, N), overflow_bits(cpu, N)
    if int((og != 0).sum()) or int((oc != 0).sum()):
        print("overflow envs (hip):", [(int(e), int(og[e])) for e in torch.nonzero(og).flatten()], "(oracle):", [(int(e), int(oc[e])) for e in torch.nonzero(oc).flatten()])
    assert torch.equal(og != 0, oc != 0)
    assert int((og != 0).sum()) <= (max(1, N // 50) if max_envs is None else max_envs)
    gpu.overflow_count(), cpu.overflow_count()


def test_model_takes_two_rows():
    model = two_body_model()
    assert model.n_dof + 6 * model.n_free == 21
    gpu = MssimSystem(device="cuda:0")
    gpu.gpu_init(model, 4)  # (mssim_create used to reject this model: "16 velocity components")
    gpu.step(1)
    assert gpu.overflow_count() == 0


@pytest.mark.parametrize("second", ["cube", "compound"])
def test_one_substep_with_two_free_bodies_matches_oracle(second):
    model = two_body_model(second)
    N = 1024
    gpu, cpu = make_pair(model, N)
    st = random_state(N, 11)
    for px in (gpu, cpu):
        set_state(px, model, N, *st)
        px.step(1)
    a, b = get_state(gpu, model, N), get_state(cpu, model, N)
    same = (a["cnt"] == b["cnt"]).all(0)
    print(f"contact counts equal in {same.float().mean():.4f} of {N} envs; contacts/env mean {b['cnt'].sum(0).float().mean():.2f} max {int(b['cnt'].sum(0).max())}")
    assert same.float().mean() >= 0.99
    light = same & (b["cnt"].sum(0) <= 8)
    assert light.float().mean() > 0.6
    assert torch.max(torch.abs(a["q"] - b["q"])[light]) < 1e-5
    assert torch.max(torch.abs(a["qd"] - b["qd"])[light]) < 5e-4
    for name in ("cubeA", "cubeB"):
        r = model.row_of(name)
        assert torch.max(torch.abs(a["rb"][r, :, :7] - b["rb"][r, :, :7])[light]) < 2e-5, name
        assert torch.max(torch.abs(a["rb"][r, :, 7:] - b["rb"][r, :, 7:])[light]) < 2e-3, name
        # every env, whatever its contact count: bounded against the oracle
        assert torch.max(torch.abs(a["rb"][r, :, :3] - b["rb"][r, :, :3])) < 2e-3, name
    assert torch.max(torch.abs(a["q"] - b["q"])) < 0.02
    assert_no_overflow_beyond_oracle(gpu, cpu, N)


def test_stack_settles_and_sleeps_like_the_oracle():
    """two cubes stacked on the table next to an idle arm: 1.5 s; both come to rest, go to sleep, and stay where the oracle has them"""
    model = two_body_model()
    N = 64
    gpu, cpu = make_pair(model, N)
    q = REST.expand(N, 9).clone()
    a = torch.zeros(N, 13)
    a[:, 2] = 0.02
    a[:, 3] = 1
    b = a.clone()
    g = torch.Generator().manual_seed(3)
    b[:, :2] = 0.008 * (2 * torch.rand(N, 2, generator=g) - 1)
    b[:, 2] = 0.04 + 0.025 + 0.002
    out = []
    for px in (gpu, cpu):
        set_state(px, model, N, q, torch.zeros(N, 9), q, a, b)
        for _ in range(30):
            px.step(5)
        out.append(get_state(px, model, N))
    ra, rb = model.row_of("cubeA"), model.row_of("cubeB")
    for r, z in ((ra, 0.02), (rb, 0.065)):
        assert torch.max(torch.abs(out[0]["rb"][r, :, 2] - z)) < 2e-4
        assert torch.max(torch.abs(out[0]["rb"][r, :, :3] - out[1]["rb"][r, :, :3])) < 2e-4
        assert torch.max(torch.abs(out[0]["rb"][r, :, 7:])) == 0.0  # asleep: exactly at rest
    assert gpu.overflow_count() == 0


def test_resynchronised_rollout_with_two_free_bodies():
    """contact-rich: the arm sweeps through two cubes; per substep from re-synchronised state (one-step error)"""
    model = two_body_model()
    N = 256
    gpu, cpu = make_pair(model, N)
    q, qd, tq, a, b = random_state(N, 5, stack_frac=0.5)
    # drive the hand down towards the cubes
    tq = q.clone()
    tq[:, 1] += 0.6
    tq[:, 3] += 0.3
    worst_q = worst_p = worst_p_any = 0.0
    agree = []
    state = (q, qd, tq, a, b)
    for step in range(40):
        for px in (gpu, cpu):
            set_state(px, model, N, *state)
            px.step(1)
        A, B = get_state(gpu, model, N), get_state(cpu, model, N)
        same = (A["cnt"] == B["cnt"]).all(0)
        light = same & (B["cnt"].sum(0) <= 8)
        agree.append(float(same.float().mean()))
        worst_q = max(worst_q, float(torch.max(torch.abs(A["q"] - B["q"])[light])))
        for name in ("cubeA", "cubeB"):
            r = model.row_of(name)
            dp = torch.abs(A["rb"][r, :, :3] - B["rb"][r, :, :3]).max(1).values
            worst_p = max(worst_p, float(dp[light].max()))
            worst_p_any = max(worst_p_any, float(dp.max()))  # every env, whatever its contacts
        # continue from the oracle's state
        state = (B["q"], B["qd"], tq, B["rb"][model.row_of("cubeA")], B["rb"][model.row_of("cubeB")])
    print(f"40 resynchronised substeps: counts agree in {min(agree):.3f}..{max(agree):.3f} of envs; <= 8 contacts: worst |dq| {worst_q:.2e}, worst |dp| {worst_p:.2e}; any env: worst |dp| {worst_p_any:.2e}")
    assert min(agree) >= 0.97 and worst_q < 5e-5 and worst_p < 5e-5 and worst_p_any < 2e-3
    assert_no_overflow_beyond_oracle(gpu, cpu, N)


def test_wide_envs_are_independent_of_their_wave_mates():
    """size-independent property: the first 8 envs of a 512-env run are bit-identical to an 8-env run from the same states"""
    model = two_body_model()
    big, small = 512, 8
    st = random_state(big, 21, stack_frac=0.5)
    outs = []
    for n in (big, small):
        px = MssimSystem(device="cuda:0")
        px.gpu_init(model, n)
        set_state(px, model, n, *[t[:n] for t in st])
        for _ in range(10):
            px.step(5)
        outs.append(get_state(px, model, n))
    assert torch.equal(outs[0]["q"][:small], outs[1]["q"])
    assert torch.equal(outs[0]["rb"][:, :small], outs[1]["rb"])


# ---------------------------------------------------------------------------------------------- four rows per env: 3..6 free bodies
def many_body_model(n_bodies, with_robot=True):
    b = SceneModelBuilder()
    if with_robot:
        b.set_articulation(panda_record())
    b.add_actor(table_record())
    b.add_actor(ground_record())
    for k in range(n_bodies):
        half = 0.02 + 0.002 * (k % 3)
        b.add_actor(cube_record(name=f"body{k}", half_size=half, p=(0.06 * (k % 3) - 0.06, 0.07 * (k // 3) - 0.035, half)))
    b.add_actor(ActorRecord("goal_site", "kinematic", [], initial_pose=geom.pose()))
    return b.compile()


def random_many_state(model, N, n_bodies, seed):
    g = torch.Generator().manual_seed(seed)
    q = REST + 0.3 * (2 * torch.rand(N, 9, generator=g) - 1)
    q[:, 3] = torch.clamp(q[:, 3], -3.0, -0.1)
    q[:, 5] = torch.clamp(q[:, 5], 0.0, 3.7)
    q[:, 7:] = 0.04 * torch.rand(N, 2, generator=g)
    qd = 0.5 * (2 * torch.rand(N, 9, generator=g) - 1)
    qd[:, 7:] *= 0.05
    tq = q.clone()
    tq[:, :7] += 0.1 * (2 * torch.rand(N, 7, generator=g) - 1)
    bodies = []
    for k in range(n_bodies):
        half = 0.02 + 0.002 * (k % 3)
        s = torch.zeros(N, 13)
        # a loose grid on the table; every other env stacks the odd bodies on the even ones
        # (0.10 m apart: two cubes of up to 48 mm turned by 45 degrees stay clear of each other, some within the contact offset)
        s[:, 0] = 0.10 * (k % 3) - 0.10 + 0.01 * (2 * torch.rand(N, generator=g) - 1)
        s[:, 1] = 0.10 * (k // 3) - 0.05 + 0.01 * (2 * torch.rand(N, generator=g) - 1)
        s[:, 2] = half
        yaw = 2 * np.pi * torch.rand(N, generator=g)
        s[:, 3], s[:, 6] = torch.cos(yaw / 2), torch.sin(yaw / 2)
        s[:, 7:10] = 0.05 * (2 * torch.rand(N, 3, generator=g) - 1)
        s[:, 10:13] = 0.2 * (2 * torch.rand(N, 3, generator=g) - 1)
        if k % 2 == 1:
            stack = torch.arange(N) % 2 == 0
            s[stack, :2] = bodies[k - 1][stack, :2]
            s[stack, 2] = 2 * (0.02 + 0.002 * ((k - 1) % 3)) + half
        bodies.append(s)
    return q, qd, tq, bodies


def set_many(px, model, N, q, qd, tq, bodies, with_robot=True):
    dev = px.device
    if with_robot:
        px.cuda_articulation_qpos.torch()[:] = q.to(dev)
        px.cuda_articulation_qvel.torch()[:] = qd.to(dev)
        px.cuda_articulation_target_qpos.torch()[:] = tq.to(dev)
    rb = px.cuda_rigid_body_data.torch()
    for k, s in enumerate(bodies):
        r = model.row_of(f"body{k}")
        rb[r * N : (r + 1) * N] = s.to(dev)
    px.gpu_apply_all()


@pytest.mark.parametrize("n_bodies", [2, 3, 4, 6])
def test_one_substep_with_many_free_bodies_matches_oracle(n_bodies):
    """the Panda with 3, 4 and 6 free bodies (27 .. 45 velocity components): four 16-lane rows -- a whole wave -- per env"""
    model = many_body_model(n_bodies)
    assert model.n_dof + 6 * model.n_free == 9 + 6 * n_bodies
    N = 512
    gpu, cpu = make_pair(model, N)
    q, qd, tq, bodies = random_many_state(model, N, n_bodies, 40 + n_bodies)
    for px in (gpu, cpu):
        set_many(px, model, N, q, qd, tq, bodies)
        px.step(1)
    a, b = get_state(gpu, model, N), get_state(cpu, model, N)
    same = (a["cnt"] == b["cnt"]).all(0)
    light = same & (b["cnt"].sum(0) <= 4 * n_bodies + 6)  # (every body on the table or on another one, and little else)
    dq = torch.abs(a["q"] - b["q"]).max(1).values
    dp = torch.stack([torch.abs(a["rb"][model.row_of(f"body{k}"), :, :7] - b["rb"][model.row_of(f"body{k}"), :, :7]).max(1).values for k in range(n_bodies)]).max(0).values
    dv = torch.stack([torch.abs(a["rb"][model.row_of(f"body{k}"), :, 7:] - b["rb"][model.row_of(f"body{k}"), :, 7:]).max(1).values for k in range(n_bodies)]).max(0).values
    print(f"{n_bodies} bodies: contact counts equal in {same.float().mean():.4f} of {N} envs; contacts/env mean {b['cnt'].sum(0).float().mean():.2f} max {int(b['cnt'].sum(0).max())}; {int(light.sum())} light envs: "
          f"|dq| {float(dq[light].max()):.2e}, body pose worst {float(dp[light].max()):.2e} (99 %: {float(dp[light].quantile(0.99)):.2e}), body velocity worst {float(dv[light].max()):.2e}; all envs: |dq| {float(dq.max()):.2e} pose {float(dp.max()):.2e}")
    assert same.float().mean() >= 0.99 and light.float().mean() > 0.4
    assert dq[light].max() < 1e-4
    # (half of the envs stack cubes of different sizes at random relative yaw, spinning: the clipped contact polygons of the
    # f32 and the f64 build keep different corners now and then, and 16 cold-started sweeps amplify that into ~0.1 rad/s -- the
    # two-body case, on two rows, shows the same figures as the three-to-six-body cases on four)
    assert dp[light].median() < 1e-6 and dp[light].quantile(0.99) < 1e-3 and dp[light].max() < 5e-3
    assert dq.max() < 0.02 and dp.max() < 1e-2  # every env, whatever its contacts
    assert_no_overflow_beyond_oracle(gpu, cpu, N)


def test_six_bodies_without_a_robot_settle_and_sleep_like_the_oracle():
    """no articulation at all, six cubes (three stacks of two): 36 velocity components in rows 1..3, row 0 idle; 1.5 s: everything
    at rest, asleep, where the oracle has it"""
    model = many_body_model(6, with_robot=False)
    N = 32
    gpu, cpu = make_pair(model, N)
    _, _, _, bodies = random_many_state(model, N, 6, 7)
    for s in bodies:
        s[:, 7:] = 0
    out = []
    for px in (gpu, cpu):
        set_many(px, model, N, None, None, None, bodies, with_robot=False)
        for _ in range(30):
            px.step(5)
        px.gpu_fetch_all()
        out.append(px.cuda_rigid_body_data.torch().cpu().clone().reshape(model.n_rows, N, 13))
    for k in range(6):
        r = model.row_of(f"body{k}")
        assert torch.max(torch.abs(out[0][r, :, :3] - out[1][r, :, :3])) < 5e-4, k
        assert torch.max(torch.abs(out[0][r, :, 7:])) == 0.0, k  # asleep: exactly at rest
    assert gpu.overflow_count() == 0


def test_resynchronised_rollout_with_five_free_bodies():
    model = many_body_model(5)
    N = 128
    gpu, cpu = make_pair(model, N)
    q, qd, tq, bodies = random_many_state(model, N, 5, 9)
    tq = q.clone()
    tq[:, 1] += 0.6
    tq[:, 3] += 0.3
    agree, worst_q, worst_p, worst_any = [], 0.0, 0.0, 0.0
    for step in range(30):
        for px in (gpu, cpu):
            set_many(px, model, N, q, qd, tq, bodies)
            px.step(1)
        A, B = get_state(gpu, model, N), get_state(cpu, model, N)
        same = (A["cnt"] == B["cnt"]).all(0)
        light = same & (B["cnt"].sum(0) <= 26)
        agree.append(float(same.float().mean()))
        assert light.any()
        worst_q = max(worst_q, float(torch.max(torch.abs(A["q"] - B["q"])[light])))
        for k in range(5):
            r = model.row_of(f"body{k}")
            dp = torch.abs(A["rb"][r, :, :3] - B["rb"][r, :, :3]).max(1).values
            worst_p, worst_any = max(worst_p, float(dp[light].max())), max(worst_any, float(dp.max()))
        q, qd = B["q"], B["qd"]
        bodies = [B["rb"][model.row_of(f"body{k}")] for k in range(5)]
    print(f"30 resynchronised substeps, 5 bodies: counts agree in {min(agree):.3f}..{max(agree):.3f}; <= 26 contacts: worst |dq| {worst_q:.2e} |dp| {worst_p:.2e}; any env |dp| {worst_any:.2e}")
    # (an env a contact apart from the oracle with bodies pressed into each other: one side pushes them apart at the
    # depenetration speed limit, 1 m/s = 1 cm in the substep, the other does so a substep later)
    assert min(agree) >= 0.95 and worst_q < 5e-5 and worst_p < 5e-4 and worst_any < 2e-2
    # (the arm pressed into a pile of five cubes, some stacked: a tenth of these envs carry more than the 52 solver blocks an env
    # holds -- reported, by both sides for the same envs)
    assert_no_overflow_beyond_oracle(gpu, cpu, N, max_envs=N // 5)
