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


def test_unmerged_fragment_is_rejected():
    """objects that exist in only some envs are not supported: the scene must say so, not mis-simulate"""
    import gymnasium as gym
    from maniskill_amd.envs.tasks.tabletop.pick_cube import PickCubeEnv

    class Partial(PickCubeEnv):
        def _load_scene(self, options):
            super()._load_scene(options)
            b = self.scene.create_actor_builder()
            b.add_box_collision(half_size=[0.02] * 3)
            b.set_scene_idxs([0])
            b.build("only_in_env0")

    with pytest.raises(NotImplementedError):
        Partial(num_envs=4, sim_backend=BACKEND)


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
