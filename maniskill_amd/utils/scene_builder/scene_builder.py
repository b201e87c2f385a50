"""SceneBuilder base (counterpart of mani_skill/utils/scene_builder/scene_builder.py)."""
from typing import List, Optional


class SceneBuilder:
    builds_lighting: bool = False
    build_configs: Optional[List] = None
    init_configs: Optional[List] = None

    def __init__(self, env, robot_init_qpos_noise=0.02):
        self.env = env
        self.robot_init_qpos_noise = robot_init_qpos_noise

    def build(self, build_config_idxs=None):
        raise NotImplementedError()

    def initialize(self, env_idx, init_config_idxs=None):
        raise NotImplementedError()

    @property
    def scene(self):
        return self.env.scene
