"""Fetch (mani_skill/agents/robots/fetch/fetch.py) in Empty-v1 (envs/tasks/empty_env.py) on the oracle back end: the
mobile base as three root joints driven by ego-centric velocity actions (agents/controllers/pd_base_vel.py), 15 velocity
components in one env. The HIP side of the same env is tests/test_gpu_env.py::test_fetch_empty_env_matches_oracle."""
import numpy as np
import pytest
import torch

import maniskill_amd.envs  # noqa: F401
import gymnasium as gym
from tests import oracle_backend as ob

BACKEND = "oracle_f64_fetch"


@pytest.fixture(scope="module")
def env():
    ob.register("f64", BACKEND)
    e = gym.make("Empty-v1", robot_uids="fetch", num_envs=3, obs_mode="state", sim_backend=BACKEND)
    yield e
    e.close()


def test_fetch_spaces_and_rest_pose(env):
    obs, _ = env.reset(seed=0)
    base = env.unwrapped
    agent = base.agent
    assert agent.robot.max_dof == 15 and obs.shape == (3, 30)
    # arm 7 + gripper 1 (mimic) + head / torso 3 + base (forward, turn) 2
    assert base.single_action_space.shape == (13,)
    assert agent.controller.action_mapping == {"arm": (0, 7), "gripper": (7, 8), "body": (8, 11), "base": (11, 13)}
    assert set(agent.supported_control_modes) >= {"pd_joint_delta_pos", "pd_joint_pos", "pd_ee_delta_pos", "pd_ee_delta_pose", "pd_joint_target_delta_pos",
                                                   "pd_ee_target_delta_pos", "pd_ee_target_delta_pose", "pd_joint_vel", "pd_joint_pos_vel", "pd_joint_delta_pos_vel",
                                                   "pd_joint_delta_pos_stiff_body"}
    rest = torch.as_tensor(agent.keyframes["rest"].qpos, dtype=torch.float32)
    hold = torch.zeros(3, 13)
    hold[:, 7] = -0.1666667  # the gripper target that equals the rest opening (0.015 in [-0.01, 0.05])
    for _ in range(20):
        env.step(hold)
    q = agent.robot.get_qpos()
    assert torch.allclose(q[:, 3:13], rest[3:13].expand(3, -1), atol=2e-3), q[0]  # (gravity is balanced: the links are weightless)
    assert float(q[:, :3].abs().max()) < 1e-4 and torch.allclose(q[:, 13:], torch.full((3, 2), 0.015), atol=1e-3)
    # the wheels share a collision bit with the ground (empty_env.py:41); the base's hull hovers within the contact offset
    # of it and carries nothing (its height is fixed by the root joints)
    for contact in base.scene.get_contacts(0):
        assert {b.name for b in contact.bodies} == {"ground", "base_link"}
        assert all(float(np.abs(p.impulse).max()) < 1e-6 for p in contact.points)
    # the gripper frame sits in front of the torso at the rest pose
    p = agent.tcp.pose.p[0]
    assert 0.3 < float(p[0]) < 1.0 and abs(float(p[1])) < 0.3 and 0.3 < float(p[2]) < 1.3, p


def test_base_actions_are_ego_centric(env):
    env.reset(seed=0)
    agent = env.unwrapped.agent
    a = torch.zeros(3, 13)
    a[:, 7] = -0.1666667
    a[0, 11] = 1.0   # env 0: forward at the upper bound, 1 m/s
    a[1, 12] = 0.5   # env 1: turn at half the bound, 1.57 rad/s
    for _ in range(20):  # 1 s
        env.step(a)
    q = agent.robot.get_qpos()
    assert 0.85 < float(q[0, 0]) < 1.0 and abs(float(q[0, 1])) < 1e-3 and abs(float(q[0, 2])) < 1e-3, q[0, :3]
    assert abs(float(q[1, 2]) - np.pi / 2) < 0.12 and float(q[1, :2].abs().max()) < 1e-2, q[1, :3]  # (the arm's mass off the turning axis pulls the base a few mm against its velocity drives)
    assert float(q[2, :3].abs().max()) < 1e-4
    # env 1 now faces +y: "forward" moves it along y
    b = torch.zeros(3, 13)
    b[:, 7] = -0.1666667
    b[1, 11] = 1.0
    y0 = float(q[1, 1])
    for _ in range(10):
        env.step(b)
    q2 = agent.robot.get_qpos()
    yaw = float(q2[1, 2])
    dx, dy = float(q2[1, 0] - q[1, 0]), float(q2[1, 1]) - y0
    assert dy > 0.35 and abs(np.arctan2(dy, dx) - yaw) < 0.15, (dx, dy, yaw)
    assert agent.is_static().tolist() == [False, False, True] or agent.is_static()[2]


def test_fetch_arm_reaches_joint_targets(env):
    env.reset(seed=0)
    agent = env.unwrapped.agent
    q0 = agent.robot.get_qpos().clone()
    a = torch.zeros(3, 13)
    a[:, 7] = -0.1666667
    a[:, 0] = 1.0    # shoulder pan +0.1 rad per step
    a[:, 10] = 1.0   # torso lift +0.1 m per step (limit 0.386 -> stays at the limit)
    for _ in range(5):
        env.step(a)
    for _ in range(15):
        b = torch.zeros(3, 13)
        b[:, 7] = -0.1666667
        env.step(b)
    q = agent.robot.get_qpos()
    # (a delta is added to the joint's current position, not to the previous target: a lagging joint covers less than 5 x 0.1)
    moved = q[:, 5] - q0[:, 5]
    assert torch.all(moved > 0.1) and torch.all(moved < 0.5), moved
    assert torch.all(q[:, 3] <= 0.39)


def test_fetch_in_rooms_that_differ_from_env_to_env(tmp_path):
    """static triangle-mesh scenery per sub-scene: every env's Fetch is stopped by the wall of its own room"""
    from tests.fetch_rooms import make_rooms_env

    ob.register("f64", BACKEND)
    env = make_rooms_env(str(tmp_path), 4, BACKEND)
    env.reset(seed=0)
    assert [w._own_idx.tolist() for w in env.walls] == [[0, 2], [1, 3]]
    a = torch.zeros(4, 13)
    a[:, 7] = -0.1666667
    a[:, 11] = 1.0
    for _ in range(50):
        env.step(a)
    q = env.agent.robot.get_qpos()
    # the base (0.3 m radius, the folded arm reaching a little further) stands in front of its own wall
    assert torch.all((q[0::2, 0] > 0.35) & (q[0::2, 0] < 0.6)) and torch.all((q[1::2, 0] > 0.95) & (q[1::2, 0] < 1.2)), q[:, 0]
    assert float(q[:, 1:3].abs().max()) < 0.02 and env.scene.px.overflow_count() == 0
    names = [{b.name for b in c.bodies} for c in env.scene.get_contacts(0)]
    assert any("wall_0" in n for n in names) and not any("wall_1" in n for n in names)
    env.close()


def test_scene_manipulation_env_with_synthetic_rooms():
    """SceneManipulation-v1 (envs/scenes/base_env.py) with the SyntheticRooms scene builder: one of three static layouts
    per sub-scene (triangle-mesh walls and furniture that exist in that layout's envs only), start arrangements chosen at
    reset, build configs only changeable with reconfigure"""
    from maniskill_amd.utils.scene_builder import REGISTERED_SCENE_BUILDERS

    ob.register("f64", BACKEND)
    assert "SyntheticRooms" in REGISTERED_SCENE_BUILDERS
    env = gym.make("SceneManipulation-v1", num_envs=6, obs_mode="state", sim_backend=BACKEND, build_config_idxs=[0, 1, 2, 0, 1, 2], scene_builder_cls="SyntheticRoomsStatic")
    base = env.unwrapped
    sb = base.scene_builder
    assert sb.build_configs == ["study", "corridor", "kitchen", "lab", "hall"] and sb.build_config_names_to_idxs["kitchen"] == 2
    # the layouts' meshes are merged into two actors (one shape slot each, a different mesh per sub-scene): 20 robot hulls + ground + 2
    assert base.agent.uid == "fetch" and base.scene.model.scalars["n_shape"] == 23 and sorted(sb.scene_objects) == ["furniture", "ground", "walls"]
    first_tri = base.scene.model.arrays["env_shape_param"]
    assert len({tuple(first_tri[:2, e]) for e in range(6)}) == 3 and len(sb.navigable_positions) == 6  # (three different meshes over the six envs)
    env.reset(seed=0, options=dict(init_config_idxs=[0, 0, 0, 1, 1, 1]))
    q = base.agent.robot.get_qpos()
    expect = torch.tensor([[-0.6, 0.0, 0.0], [-0.5, 0.0, 0.0], [-0.6, 0.4, -np.pi / 2], [0.0, -0.6, np.pi / 2], [2.0, -0.1, np.pi], [-0.6, -0.5, 0.0]])
    assert torch.allclose(q[:, :3], expect, atol=1e-5), q[:, :3]
    with pytest.raises(AssertionError):
        env.reset(options=dict(build_config_idxs=[0] * 6))
    # drive forward for 3 s: every base ends in front of what its own room has there
    a = torch.zeros(6, 13)
    a[:, 7] = -0.1666667
    a[:, 11] = 1.0
    for _ in range(60):
        env.step(a)
    q = base.agent.robot.get_qpos()
    assert 0.45 < float(q[0, 0]) < 0.7        # study, facing the desk at x = 0.9
    assert 2.0 < float(q[1, 0]) < 2.7         # corridor from its closed end: past the shelf (it leaves 0.35 m beside the axis), towards the door
    assert -1.4 < float(q[2, 1]) < -0.6       # kitchen, facing the counter along y = -1.2
    assert 0.9 < float(q[3, 1]) < 1.3         # study, facing the wall at y = 1.5
    assert -0.8 < float(q[4, 0]) < -0.55      # corridor from the far end, back to the wall at x = -1
    assert -0.2 < float(q[5, 0]) < 0.2        # kitchen, facing the island at x = 0.2
    assert base.scene.px.overflow_count() == 0
    # a new set of layouts needs a rebuild
    env.reset(seed=1, options=dict(reconfigure=True, build_config_idxs=[4, 4, 3, 3, 1, 1]))
    assert base.scene_builder.build_config_idxs == [4, 4, 3, 3, 1, 1]
    rows = base.scene.model.arrays["env_shape_param"]
    assert len({tuple(rows[:2, e]) for e in range(6)}) == 3 and tuple(rows[:2, 0]) == tuple(rows[:2, 1]) != tuple(rows[:2, 2])
    env.close()


def test_rooms_carry_their_own_movable_objects():
    """BASELINE config 5's other half (utils/scene_builder/replicacad/scene_builder.py:156-185): every layout of SyntheticRooms
    has two movable objects of its own, each a convex decomposition of several hulls, built per layout with `set_scene_idxs`
    and never merged. Ten distinct objects, two per sub-scene: they share two body rows (envs/scene.py `_setup`), and each
    stays a batched object over its own envs. With the Fetch that is 15 + 12 = 27 velocity components per env."""
    ob.register("f64", BACKEND)
    env = gym.make("SceneManipulation-v1", num_envs=10, obs_mode="state", sim_backend=BACKEND, build_config_idxs=[i % 5 for i in range(10)])
    base = env.unwrapped
    sb, model = base.scene_builder, base.scene.model
    assert model.n_dof == 15 and model.n_free == 2 and model.n_dof + 6 * model.n_free == 27
    assert len(sb.movable_objects) == 20 and len({id(a) for a in sb.movable_objects.values()}) == 10
    mug, tee = sb.movable_objects["env-0_mug"], sb.movable_objects["env-1_tee"]
    assert mug is sb.movable_objects["env-5_mug"] and mug._own_idx.tolist() == [0, 5] and tee._own_idx.tolist() == [1, 6]
    assert mug._body_row == tee._body_row and mug.pose.p.shape == (2, 3)  # one row of the state, different envs
    assert sorted(k for k in sb.scene_objects if not k.startswith("env-")) == ["furniture", "ground", "walls"]
    env.reset(seed=0)
    # each object where its layout has it, and the masses of different objects in one row differ per env
    assert torch.allclose(mug.pose.p, torch.tensor([[1.05, -0.25, 0.75]] * 2), atol=1e-6) and torch.allclose(tee.pose.p, torch.tensor([[1.85, 0.5, 0.9]] * 2), atol=1e-6)
    assert abs(float(mug.mass[0]) - float(tee.mass[0])) > 0.05
    hold = torch.zeros(10, 13)
    hold[:, 7] = -0.1666667
    for _ in range(20):
        env.step(hold)
    # at rest on their furniture / on the floor (1 s): nothing falls through a table top, nothing sinks into the ground
    for name, actor in sb.movable_objects.items():
        env_i, oname = int(name.split("_", 1)[0][4:]), name.split("_", 1)[1]
        z0 = next(xyz[2] for o, xyz, _ in sb_objects()[sb.build_configs[sb.build_config_idxs[env_i]]] if o == oname)
        assert float((actor.pose.p[:, 2] - z0).abs().max()) < 1.5e-3, (name, actor.pose.p[:, 2], z0)
        assert float(actor.linear_velocity.abs().max()) < 5e-3, name
    # the state dict lists every object with the rows of its own envs; restoring it restores the objects
    st = base.get_state_dict()
    assert st["actors"]["study_mug"].shape == (2, 13) and len(st["actors"]) == 10
    # env 0: the Fetch drives into the bracket lying 0.75 m in front of it and pushes it along
    bracket = sb.movable_objects["env-0_bracket"]
    x0 = float(bracket.pose.p[0, 0])
    go = hold.clone()
    go[:, 11] = 1.0
    for _ in range(30):
        env.step(go)
    assert float(bracket.pose.p[0, 0]) > x0 + 0.2, bracket.pose.p
    assert abs(float(bracket.pose.p[1, 0]) - float(bracket.pose.p[0, 0])) > 1e-3  # (env 5 starts from another arrangement)
    base.set_state_dict(st)
    assert torch.allclose(bracket.pose.p[:, 0], torch.tensor([x0, x0]), atol=2e-3)
    assert base.scene.px.overflow_count() == 0
    # a partial reset puts the objects of the reset envs back, and only those
    for _ in range(30):
        env.step(go)
    moved = bracket.pose.p.clone()
    env.reset(options=dict(env_idx=torch.tensor([5])))
    assert torch.allclose(bracket.pose.p[0], moved[0]) and torch.allclose(bracket.pose.p[1], torch.tensor([0.15, 0.05, 0.0]), atol=1e-6)
    env.close()


def sb_objects():
    from maniskill_amd.utils.scene_builder.synthetic_rooms.scene_builder import LAYOUT_OBJECTS

    return LAYOUT_OBJECTS
