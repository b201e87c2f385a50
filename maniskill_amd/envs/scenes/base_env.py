"""SceneManipulation-v1 (mani_skill/envs/scenes/base_env.py:19-128): a robot -- the Fetch by default -- in scenery made by
a scene builder; no task, no reward. Each sub-scene gets one of the builder's static layouts (`build_config_idxs`, fixed
when the scene is built: changing them needs `reset(options=dict(reconfigure=True, build_config_idxs=...))`) and, at
every reset, one of its start arrangements (`init_config_idxs`).

The reference's default builder is ReplicaCAD; its assets are not available here, so the default is `SyntheticRooms`
(utils/scene_builder/synthetic_rooms), and asking for a builder that is not registered says so.
"""
from typing import Union

import torch

from maniskill_amd.envs.sapien_env import BaseEnv
from maniskill_amd.utils.registration import register_env
from maniskill_amd.utils.scene_builder import REGISTERED_SCENE_BUILDERS, SceneBuilder
from maniskill_amd.utils.scene_builder import synthetic_rooms  # noqa: F401  (registers "SyntheticRooms")


def _as_list(idxs):
    return [idxs] if isinstance(idxs, int) else idxs


@register_env("SceneManipulation-v1", max_episode_steps=200)
class SceneManipulationEnv(BaseEnv):
    SUPPORTED_ROBOTS = ["panda", "fetch"]
    SUPPORTED_REWARD_MODES = ["none"]

    def __init__(self, *args, robot_uids="fetch", scene_builder_cls: Union[str, type] = "SyntheticRooms", build_config_idxs=None, init_config_idxs=None,
                 num_envs=1, reconfiguration_freq=None, **kwargs):
        if isinstance(scene_builder_cls, str):
            if scene_builder_cls not in REGISTERED_SCENE_BUILDERS:
                raise KeyError(f"scene builder {scene_builder_cls!r} is not registered here (registered: {sorted(REGISTERED_SCENE_BUILDERS)}); "
                               "ReplicaCAD / AI2THOR need dataset downloads this build does not have")
            scene_builder_cls = REGISTERED_SCENE_BUILDERS[scene_builder_cls].scene_builder_cls
        self.scene_builder: SceneBuilder = scene_builder_cls(self)
        self.build_config_idxs, self.init_config_idxs = _as_list(build_config_idxs), _as_list(init_config_idxs)
        if reconfiguration_freq is None:  # (a single env shows a new scene at every reset, a batch keeps its scenes)
            reconfiguration_freq = 1 if num_envs == 1 else 0
        super().__init__(*args, robot_uids=robot_uids, num_envs=num_envs, reconfiguration_freq=reconfiguration_freq, **kwargs)

    def reset(self, seed=None, options=None):
        options = dict(reconfigure=False) if options is None else options
        if options.get("reconfigure"):
            self.build_config_idxs = _as_list(options.get("build_config_idxs", self.build_config_idxs))
            self.init_config_idxs = _as_list(options.get("init_config_idxs"))
        else:
            assert "build_config_idxs" not in options, "options dict cannot contain build_config_idxs without reconfigure=True"
            self.init_config_idxs = _as_list(options.get("init_config_idxs", self.init_config_idxs))
        return super().reset(seed=seed, options=options)

    def _load_lighting(self, options: dict):
        if not self.scene_builder.builds_lighting:
            super()._load_lighting(options)

    def _load_agent(self, options: dict):
        super()._load_agent(options, self.scene_builder.robot_initial_pose)

    def _load_scene(self, options: dict):
        sb = self.scene_builder
        if sb.build_configs is None:
            sb.build()
        else:
            sb.build(self.build_config_idxs if self.build_config_idxs is not None else sb.sample_build_config_idxs())

    def _initialize_episode(self, env_idx: torch.Tensor, options: dict):
        sb = self.scene_builder
        if sb.init_configs is None:
            sb.initialize(env_idx)
        else:
            sb.initialize(env_idx, self.init_config_idxs if self.init_config_idxs is not None else sb.sample_init_config_idxs())

    def evaluate(self):
        return dict()

    def _get_obs_extra(self, info: dict):
        return dict()
