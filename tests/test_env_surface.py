"""Env-surface behaviour on CPU: the env layer (host logic) driven by the oracle registered as a
test backend. The same checks run against the HIP backend in tests/test_gpu_env.py."""
import pytest
import torch

from tests import env_checks as ec
from tests import oracle_backend as ob

BACKEND = "oracle_f64_env"


@pytest.fixture(scope="module", autouse=True)
def _register():
    ob.register("f64", BACKEND)


def test_shapes_and_devices():
    ec.check_shapes_and_devices(BACKEND, "cpu")


def test_state_dict_roundtrip():
    ec.check_state_dict_roundtrip(BACKEND)


def test_partial_reset_isolation():
    ec.check_partial_reset_isolation(BACKEND)


def test_seeded_reset_determinism():
    ec.check_seeded_reset_determinism(BACKEND)


def test_timelimit_and_vector_autoreset():
    ec.check_timelimit_and_vector_autoreset(BACKEND)


def test_hidden_object_semantics():
    ec.check_hidden_object_semantics(BACKEND)


def test_push_cube():
    ec.check_push_cube(BACKEND)


def test_scripted_pick_and_lift():
    ec.check_scripted_pick_and_lift(BACKEND)


def test_ee_controllers():
    ec.check_ee_controllers(BACKEND)


def test_peg_insertion_per_env_geometry():
    ec.check_peg_insertion(BACKEND, "cpu")


def test_object_that_exists_in_some_envs_only():
    """an actor built for a subset of the sub-scenes and not merged (set_scene_idxs, actor_builder.py:166-260) is a batched
    object over ITS envs: getters return one row per such env, setters take as many, partial resets address the object's
    envs among the ones being reset; in the other envs the body does not exist -- nothing rests on it, it never moves"""
    from maniskill_amd.envs.tasks.tabletop.pick_cube import PickCubeEnv
    from maniskill_amd.utils.structs.pose import Pose

    class Partial(PickCubeEnv):
        def _load_scene(self, options):
            super()._load_scene(options)
            b = self.scene.create_actor_builder()
            b.add_box_collision(half_size=[0.02, 0.02, 0.03])
            b.initial_pose = Pose.create_from_pq([0.2, 0.25, 0.05])
            b.set_scene_idxs([0, 2])
            self.extra = b.build("only_in_envs_0_and_2")

    N = 4
    env = Partial(num_envs=N, sim_backend=BACKEND, obs_mode="state")
    env.reset(seed=0)
    extra = env.extra
    assert extra._num_objs == 2 and extra.pose.raw_pose.shape == (2, 7) and extra.linear_velocity.shape == (2, 3)
    assert extra.mass.shape == (2,) and torch.allclose(extra.mass, torch.full((2,), 1000 * 8 * 0.02 * 0.02 * 0.03))
    zero = torch.zeros(N, env.single_action_space.shape[0])
    for _ in range(30):
        env.step(zero)
    assert torch.allclose(extra.pose.p[:, 2], torch.full((2,), 0.03), atol=5e-4)  # dropped from 0.05, resting on its 0.03 half height
    # the body's rows in the envs without it: where they were put, at rest
    rows = env.scene.px.cuda_rigid_body_data.torch()[extra._body_row * N : (extra._body_row + 1) * N]
    assert torch.allclose(rows[[1, 3], :3].cpu(), torch.tensor([[0.2, 0.25, 0.05]] * 2)) and float(rows[[1, 3], 7:].abs().max()) == 0.0
    # setters address the object's own envs; a partial reset of env 2 (and of env 1, where it does not exist) moves one row
    extra.set_pose(Pose.create_from_pq(torch.tensor([[0.3, 0.2, 0.1], [0.3, -0.2, 0.1]])))
    assert torch.allclose(extra.pose.p[:, 1], torch.tensor([0.2, -0.2]))
    with env.scene._narrow_reset_mask(torch.tensor([1, 2])):
        extra.set_pose(Pose.create_from_pq(torch.tensor([[0.0, 0.3, 0.2]])))
    assert torch.allclose(extra.pose.p, torch.tensor([[0.3, 0.2, 0.1], [0.0, 0.3, 0.2]]))
    state = env.get_state_dict()
    assert state["actors"]["only_in_envs_0_and_2"].shape == (2, 13)
    env.step(zero)
    env.set_state_dict(state)
    assert torch.allclose(extra.pose.p, torch.tensor([[0.3, 0.2, 0.1], [0.0, 0.3, 0.2]]))


def test_per_env_object_set_from_mesh_files(tmp_path):
    """the PickSingleYCB pattern (mani_skill/envs/tasks/tabletop/pick_single_ycb.py:110-140: one object model per sub-scene,
    built with set_scene_idxs([i]) and merged with Actor.merge) on synthetic assets: four polyhedra written as OBJ files,
    env i carries model i % 4 as a convex collision mesh. The merged actor is one batched body whose geometry, mass and
    rest height differ per env."""
    import numpy as np
    from scipy.spatial import ConvexHull

    from maniskill_amd.envs.tasks.tabletop.pick_cube import PickCubeEnv
    from maniskill_amd.utils.structs.actor import Actor
    from maniskill_amd.utils.structs.pose import Pose

    rng = np.random.default_rng(5)
    files = []
    for k in range(4):
        pts = rng.normal(size=(30, 3))
        pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * np.array([0.03 + 0.01 * k, 0.03, 0.02 + 0.005 * k])
        hull = ConvexHull(pts)
        f = tmp_path / f"model{k}.obj"
        with open(f, "w") as fh:
            for v in pts:
                fh.write(f"v {v[0]} {v[1]} {v[2]}\n")
            for t in hull.simplices:
                fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
        files.append(str(f))

    class ObjectSet(PickCubeEnv):
        def _load_scene(self, options):
            super()._load_scene(options)
            frags = []
            for i in range(self.num_envs):
                b = self.scene.create_actor_builder()
                b.add_convex_collision_from_file(files[i % 4], density=500)
                b.initial_pose = Pose.create_from_pq([0.25, 0.2, 0.06])
                b.set_scene_idxs([i])
                frags.append(b.build(name=f"model-{i}"))
            self.obj_set = Actor.merge(frags, name="object_set")
            self.add_to_state_dict_registry(self.obj_set)

    N = 8
    env = ObjectSet(num_envs=N, sim_backend=BACKEND, obs_mode="state")
    env.reset(seed=0)
    assert env.obj_set.pose.raw_pose.shape == (N, 7)
    mass = env.obj_set._mass_per_env
    assert torch.allclose(mass[:4], mass[4:]) and len(set(np.round(mass[:4].numpy(), 5))) == 4
    for _ in range(30):
        env.step(torch.zeros(N, env.single_action_space.shape[0]))
    z = env.obj_set.pose.p[:, 2].cpu().numpy()
    assert np.all(z > 0.005) and np.all(z < 0.06)
    assert np.allclose(z[:4], z[4:], atol=1e-6) and len(set(np.round(z[:4], 4))) >= 3  # per-env geometry, per-env rest height
    assert env.scene.px.overflow_count() == 0


def test_nonconvex_collision_from_file(tmp_path):
    """`add_nonconvex_collision_from_file` (actor_builder.py:136-150) on a synthetic asset: a ramp with a rim written as an
    OBJ (quads, fanned by the loader) becomes a static triangle mesh; a box put on it slides down (tan 21.8 deg = 0.4 > mu)
    and is stopped by the rim -- the same in every env"""
    import numpy as np

    from maniskill_amd.envs.scene import ManiSkillScene
    from maniskill_amd.utils.structs.pose import Pose
    from tests import oracle_backend as ob

    # a 0.3 x 0.2 m ramp falling 0.12 m along x, ending in a 3 cm wall
    V = [(0, -0.1, 0.12), (0, 0.1, 0.12), (0.3, 0.1, 0.0), (0.3, -0.1, 0.0), (0.3, -0.1, 0.03), (0.3, 0.1, 0.03)]
    f = tmp_path / "ramp.obj"
    with open(f, "w") as fh:
        for v in V:
            fh.write("v %g %g %g\n" % v)
        fh.write("f 1 4 3 2\nf 4 5 6 3\n")
    ob.register("f64", "oracle_f64_env")
    scene = ManiSkillScene(2, device="cpu", backend_name="oracle_f64_env")
    b = scene.create_actor_builder()
    b.add_nonconvex_collision_from_file(str(f))
    b.initial_pose = Pose.create_from_pq([0.0, 0.0, 0.0])
    b.build_static("ramp")
    t = np.arctan2(0.12, 0.3)
    b = scene.create_actor_builder()
    b.add_box_collision(half_size=[0.02, 0.02, 0.02])
    b.initial_pose = Pose.create_from_pq([0.05 + 0.02 * np.sin(t), 0.0, 0.12 - 0.05 * np.tan(t) + 0.02 * np.cos(t) + 0.001], [np.cos(t / 2), 0, np.sin(t / 2), 0])
    slider = b.build("slider")
    scene._setup()
    assert scene.model.arrays["tri_soup"].shape == (4, 12) and scene.model.arrays["tri_bvh"].shape[0] == 1
    px = scene.px
    x0 = slider.pose.p[:, 0].clone()
    for _ in range(12):
        px.step(10)
        px.gpu_fetch_all()
    p = slider.pose.p
    assert torch.all(p[:, 0] > x0 + 0.1) and torch.all(p[:, 0] < 0.3 - 0.008), p  # slid down and stopped at the rim (tilted: its lower front edge at the wall)
    assert torch.all(p[:, 2] > 0.015) and torch.all(p[:, 2] < 0.06), p
    assert torch.allclose(p[0], p[1]) and px.overflow_count() == 0


def test_physx_module_config_flow():
    ec.check_physx_module_config(BACKEND)


def test_hip_backend_fails_loudly_without_gpu():
    import gymnasium as gym

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        gym.make("PickCube-v1", num_envs=4)  # default backend = HIP: must not fall back to any CPU path


def test_mani_skill_alias_imports():
    import mani_skill  # noqa: F401
    from mani_skill.envs.sapien_env import BaseEnv
    from mani_skill.utils.structs.pose import Pose
    from mani_skill.utils.registration import register_env
    from mani_skill.agents.robots import Panda
    from maniskill_amd.envs.sapien_env import BaseEnv as B2

    assert BaseEnv is B2 and Panda.uid == "panda"


def test_time_limit_wrapper_takes_over_gymnasiums_own(monkeypatch):
    """real gymnasium wraps the env in its scalar TimeLimit before the registered additional wrappers run; the ManiSkill
    wrapper must take that limit over (it carries `gym.make(..., max_episode_steps=K)`) and remove the wrapper, as the
    reference does (utils/registration.py:131-150). gymnasium itself is absent here: its wrapper is stood in for."""
    import types

    import gymnasium as gym

    from maniskill_amd.utils.registration import REGISTERED_ENVS, TimeLimitWrapper

    class FakeTimeLimit(gym.Wrapper):
        def __init__(self, env, max_episode_steps):
            super().__init__(env)
            self._max_episode_steps = max_episode_steps

        def step(self, action):
            raise AssertionError("gymnasium's own TimeLimit must not stay in the chain")

    monkeypatch.setattr(gym, "wrappers", types.SimpleNamespace(TimeLimit=FakeTimeLimit), raising=False)
    base = REGISTERED_ENVS["PickCube-v1"].make(num_envs=2, sim_backend=BACKEND, obs_mode="state")
    env = TimeLimitWrapper(FakeTimeLimit(base, 7), 50)  # registered limit 50, the user asked for 7
    assert env._max_episode_steps == 7 and env.env is base
    env.reset(seed=0)
    for i in range(7):
        _, _, _, truncated, _ = env.step(torch.zeros(2, base.single_action_space.shape[0]))
        assert bool(truncated.all()) == (i == 6)
