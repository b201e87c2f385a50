"""ManiSkillVectorEnv: gym VectorEnv facade with partial auto-reset and episode metrics
(counterpart of mani_skill/vector/wrappers/gymnasium.py:16-173)."""
from typing import Dict, List, Optional, Tuple, Union

import gymnasium as gym
import torch
from gymnasium.vector import VectorEnv

from maniskill_amd.utils.common import torch_clone_dict


def _wrapper_attr(env, name):
    cur = env
    while cur is not None:
        if name in getattr(cur, "__dict__", {}) or hasattr(type(cur), name):
            return getattr(cur, name)
        cur = getattr(cur, "env", None) if "env" in getattr(cur, "__dict__", {}) else None
    return getattr(env.unwrapped, name)


class ManiSkillVectorEnv(VectorEnv):
    def __init__(self, env, num_envs: int = None, auto_reset: bool = True, ignore_terminations: bool = False, record_metrics: bool = False, **kwargs):
        if isinstance(env, str):
            self._env = gym.make(env, num_envs=num_envs, **kwargs)
        else:
            self._env = env
        num_envs = self.base_env.num_envs
        self.auto_reset = auto_reset
        self.ignore_terminations = ignore_terminations
        self.record_metrics = record_metrics
        self.spec = getattr(self._env, "spec", None)
        super().__init__(num_envs, _wrapper_attr(self._env, "single_observation_space"), _wrapper_attr(self._env, "single_action_space"))
        if not self.ignore_terminations and auto_reset:
            assert self.base_env.reconfiguration_freq == 0 or self.base_env.num_envs == 1, (
                "With partial resets, environment cannot be reconfigured automatically"
            )
        if self.record_metrics:
            dev = self.base_env.device
            self.success_once = torch.zeros(self.num_envs, device=dev, dtype=torch.bool)
            self.fail_once = torch.zeros(self.num_envs, device=dev, dtype=torch.bool)
            self.returns = torch.zeros(self.num_envs, device=dev, dtype=torch.float32)

    @property
    def device(self):
        return self.base_env.device

    @property
    def base_env(self):
        return self._env.unwrapped

    @property
    def unwrapped(self):
        return self.base_env

    def reset(self, *, seed: Optional[Union[int, List[int]]] = None, options: Optional[dict] = None):
        options = dict() if options is None else options
        obs, info = self._env.reset(seed=seed, options=options)
        if self.record_metrics:
            if "env_idx" in options:
                idx = options["env_idx"]
                self.success_once[idx] = False
                self.fail_once[idx] = False
                self.returns[idx] = 0
            else:
                self.success_once[:] = False
                self.fail_once[:] = False
                self.returns[:] = 0
        return obs, info

    def step(self, actions):
        obs, rew, terminations, truncations, infos = self._env.step(actions)
        if self.record_metrics:
            ep = dict()
            self.returns += rew
            if "success" in infos:
                self.success_once = self.success_once | infos["success"]
                ep["success_once"] = self.success_once.clone()
            if "fail" in infos:
                self.fail_once = self.fail_once | infos["fail"]
                ep["fail_once"] = self.fail_once.clone()
            ep["return"] = self.returns.clone()
            ep["episode_len"] = self.base_env.elapsed_steps.clone()
            ep["reward"] = ep["return"] / ep["episode_len"]
        if isinstance(terminations, bool):
            terminations = torch.tensor([terminations], device=self.device)
        if self.ignore_terminations:
            terminations[:] = False
            if self.record_metrics:
                if "success" in infos:
                    ep["success_at_end"] = infos["success"].clone()
                if "fail" in infos:
                    ep["fail_at_end"] = infos["fail"].clone()
        if self.record_metrics:
            infos["episode"] = ep
        dones = torch.logical_or(terminations, truncations)
        if self.auto_reset and bool(dones.any()):
            final_obs = torch_clone_dict(obs)
            env_idx = torch.arange(0, self.num_envs, device=self.device)[dones]
            final_info = torch_clone_dict(infos)
            obs, infos = self.reset(options=dict(env_idx=env_idx))
            infos["final_observation"] = final_obs
            infos["final_info"] = final_info
            infos["_final_info"] = dones
            infos["_final_observation"] = dones
            infos["_elapsed_steps"] = dones
        return obs, rew, terminations, truncations, infos

    def close(self):
        return self._env.close()

    def call(self, name: str, *args, **kwargs):
        return getattr(self._env, name)(*args, **kwargs)

    def get_attr(self, name: str):
        raise RuntimeError("To get an attribute get it from the .env property of this object")

    def render(self):
        return self.base_env.render()
