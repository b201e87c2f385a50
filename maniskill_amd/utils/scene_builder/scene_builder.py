"""SceneBuilder: scenery that several tasks share, built into an env's scene and re-initialised at every reset (interface of
mani_skill/utils/scene_builder/scene_builder.py:18-95).

A builder may know several static layouts (`build_configs`: which one an env gets is decided when the scene is built,
one index per env) and several start arrangements (`init_configs`: chosen per reset). Both are plain lists of whatever the
builder understands; the samplers draw one index per env from torch's generator, which `BaseEnv.reset` seeds.
"""
from typing import Any, Dict, List, Optional

import sapien
import torch


class SceneBuilder:
    robot_init_qpos_noise: float = 0.02
    robot_initial_pose = sapien.Pose()  # handed to `_load_agent`
    builds_lighting: bool = False
    build_configs: Optional[List[Any]] = None
    init_configs: Optional[List[Any]] = None
    # what a build leaves behind, by name: every actor / the dynamic ones / the articulations
    scene_objects: Optional[Dict[str, Any]] = None
    movable_objects: Optional[Dict[str, Any]] = None
    articulations: Optional[Dict[str, Any]] = None
    # where a mobile robot may stand, one entry per env (positions, a box, ...)
    navigable_positions: Optional[List[Any]] = None

    def __init__(self, env, robot_init_qpos_noise=0.02):
        self.env = env
        self.robot_init_qpos_noise = robot_init_qpos_noise

    def build(self, build_config_idxs: List[int] = None):
        """create the scenery (no poses, no joint states: that is `initialize`)"""
        raise NotImplementedError()

    def initialize(self, env_idx: torch.Tensor, init_config_idxs: List[int] = None):
        """put the scenery and the robot of the envs in `env_idx` into a start arrangement"""
        raise NotImplementedError()

    def _one_index_per_env(self, choices) -> List[int]:
        return torch.randint(low=0, high=len(choices), size=(self.env.num_envs,)).tolist()

    def sample_build_config_idxs(self) -> List[int]:
        return self._one_index_per_env(self.build_configs)

    def sample_init_config_idxs(self) -> List[int]:
        return self._one_index_per_env(self.init_configs)

    @property
    def build_config_names_to_idxs(self) -> Dict[str, int]:
        return {name: i for i, name in enumerate(self.build_configs)}

    @property
    def init_config_names_to_idxs(self) -> Dict[str, int]:
        return {name: i for i, name in enumerate(self.init_configs)}

    @property
    def scene(self):
        return self.env.scene
