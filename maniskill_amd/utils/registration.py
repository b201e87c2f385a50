"""Env registry + gym registration (counterpart of mani_skill/utils/registration.py:25-260).
`TimeLimitWrapper` returns the real (batched tensor) truncation signal
`elapsed_steps >= max_episode_steps` (:127-168)."""
import json
from copy import deepcopy
from functools import partial
from typing import Dict, List, Optional, Type

import gymnasium as gym
import torch
from gymnasium.envs.registration import WrapperSpec


class EnvSpec:
    def __init__(self, uid: str, cls, max_episode_steps=None, asset_download_ids: Optional[List[str]] = None, default_kwargs: dict = None):
        self.uid = uid
        self.cls = cls
        self.max_episode_steps = max_episode_steps
        self.asset_download_ids = asset_download_ids or []
        self.default_kwargs = {} if default_kwargs is None else default_kwargs

    def make(self, **kwargs):
        kw = self.default_kwargs.copy()
        kw.update(kwargs)
        return self.cls(**kw)


REGISTERED_ENVS: Dict[str, EnvSpec] = {}


def register(name: str, cls, max_episode_steps=None, asset_download_ids: List[str] = None, default_kwargs: dict = None):
    from maniskill_amd.envs.sapien_env import BaseEnv

    if not issubclass(cls, BaseEnv):
        raise TypeError(f"Env {name} must inherit from BaseEnv")
    REGISTERED_ENVS[name] = EnvSpec(name, cls, max_episode_steps=max_episode_steps, asset_download_ids=asset_download_ids, default_kwargs=default_kwargs)


class TimeLimitWrapper(gym.Wrapper):
    """batched-tensor truncation; `max_episode_steps` passed to gym.make overrides the registered one"""

    def __init__(self, env, max_episode_steps: int):
        super().__init__(env)
        # Under real gymnasium `gym.make(id, max_episode_steps=K)` has already wrapped the env in gymnasium's own
        # TimeLimit (scalar `truncated`, K or the registered value) by the time the additional wrappers are applied. The
        # reference takes K over and removes that wrapper (mani_skill/utils/registration.py:131-150, by looking into
        # gym.make's frame); here the limit is read from the wrapper itself before it is spliced out of the chain.
        tl_cls = getattr(getattr(gym, "wrappers", None), "TimeLimit", None)
        if tl_cls is not None:
            parent, cur = self, env
            while cur is not None:
                if isinstance(cur, tl_cls):
                    inner_limit = getattr(cur, "_max_episode_steps", None)
                    if inner_limit is not None:
                        max_episode_steps = inner_limit
                    parent.env = cur.env
                    break
                parent, cur = cur, getattr(cur, "env", None)
        self._max_episode_steps = max_episode_steps
        # tasks with a fused native epilogue evaluate the comparison below in that launch
        self.env.unwrapped._time_limit = max_episode_steps

    @property
    def base_env(self):
        return self.env.unwrapped

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        if self._max_episode_steps is not None:
            fused = self.base_env._fused_truncated  # this step's `elapsed_steps >= limit` from the fused epilogue, if any
            truncated = fused.view(torch.bool) if fused is not None else self.base_env.elapsed_steps >= self._max_episode_steps
        else:
            truncated = torch.zeros((self.base_env.num_envs,), dtype=torch.bool, device=self.base_env.device)
        return obs, reward, terminated, truncated, info

    def get_wrapper_attr(self, name):
        return getattr(self, name)


def make(env_id, max_episode_steps=None, **kwargs):
    if env_id not in REGISTERED_ENVS:
        raise KeyError("Env {} not found in registry".format(env_id))
    spec = REGISTERED_ENVS[env_id]
    env = spec.make(**kwargs)
    return env


def _gym_entry(env_id, max_episode_steps_default, max_episode_steps=None, **kwargs):
    """entry point used by gym.make: builds the env and applies the ManiSkill time limit"""
    env = make(env_id, **kwargs)
    steps = max_episode_steps if max_episode_steps is not None else max_episode_steps_default
    return TimeLimitWrapper(env, steps)


def make_vec(env_id, **kwargs):
    from maniskill_amd.vector.wrappers.gymnasium import ManiSkillVectorEnv

    return ManiSkillVectorEnv(gym.make(env_id, **kwargs))


def register_env(uid: str, max_episode_steps=None, override=False, asset_download_ids: List[str] = None, **kwargs):
    try:
        json.dumps(kwargs)
    except TypeError:
        raise RuntimeError("You cannot register_env with non json dumpable kwargs, e.g. classes or types.")

    def _register_env(cls):
        if uid in REGISTERED_ENVS:
            if not override:
                return cls
            REGISTERED_ENVS.pop(uid)
            gym.envs.registration.registry.pop(uid, None)
        register(uid, cls, max_episode_steps=max_episode_steps, asset_download_ids=asset_download_ids, default_kwargs=deepcopy(kwargs))
        if getattr(gym, "__maniskill_amd_shim__", False):
            gym.register(uid, entry_point=partial(_gym_entry, uid, max_episode_steps), max_episode_steps=max_episode_steps,
                         kwargs=deepcopy(kwargs), pass_max_episode_steps=True)
        else:  # real gymnasium: same layout as the reference (additional_wrappers carries the time limit)
            gym.register(
                uid,
                entry_point=partial(make, env_id=uid),
                max_episode_steps=max_episode_steps,
                disable_env_checker=True,
                kwargs=deepcopy(kwargs),
                additional_wrappers=(
                    WrapperSpec("MSTimeLimit", entry_point="maniskill_amd.utils.registration:TimeLimitWrapper", kwargs=dict(max_episode_steps=max_episode_steps)),
                ),
            )
        return cls

    return _register_env
