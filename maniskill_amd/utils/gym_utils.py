"""gym space helpers (counterpart of the functions of mani_skill/utils/gym_utils.py that the env
layer uses: normalize_action_space :92-99, clip_and_scale_action :102-105,
convert_observation_to_space :20-65, find_max_episode_steps_value :108-150)."""
import numpy as np
import torch
from gymnasium import spaces


def normalize_action_space(action_space: spaces.Box) -> spaces.Box:
    assert isinstance(action_space, spaces.Box), type(action_space)
    return spaces.Box(-1, 1, shape=action_space.shape, dtype=action_space.dtype)


def clip_and_scale_action(action, low, high):
    """clip to [-1, 1] then map affinely to [low, high]"""
    action = torch.clip(action, -1, 1)
    return 0.5 * (high + low) + 0.5 * (high - low) * action


def inv_scale_action(action, low, high):
    return (action - 0.5 * (high + low)) / (0.5 * (high - low))


def convert_observation_to_space(observation, prefix="", unbatched=False):
    """observation (numpy / dict of numpy) -> gym space; `unbatched` strips the leading env dim"""
    if isinstance(observation, (dict,)):
        return spaces.Dict({k: convert_observation_to_space(v, prefix + "/" + k, unbatched=unbatched) for k, v in observation.items()})
    if isinstance(observation, torch.Tensor):
        observation = observation.cpu().numpy()
    if isinstance(observation, np.ndarray):
        shape = observation.shape[1:] if unbatched else observation.shape
        dtype = observation.dtype
        if np.issubdtype(dtype, np.floating):
            low, high = -np.inf, np.inf
        elif dtype == np.bool_:
            low, high = 0, 1
        else:
            info = np.iinfo(dtype)
            low, high = info.min, info.max
        return spaces.Box(low, high, shape=shape, dtype=dtype)
    if isinstance(observation, (float, np.float32, np.float64)):
        return spaces.Box(-np.inf, np.inf, shape=[1], dtype=np.float32)
    if isinstance(observation, (int, np.int32, np.int64)):
        return spaces.Box(-np.inf, np.inf, shape=[1], dtype=int)
    if isinstance(observation, (bool, np.bool_)):
        return spaces.Box(0, 1, shape=[1], dtype=np.bool_)
    raise NotImplementedError(type(observation), observation)


def find_max_episode_steps_value(env):
    """walk the wrapper chain for a max_episode_steps value (TimeLimit wrappers or the spec)"""
    cur = env
    while cur is not None:
        if hasattr(cur, "__dict__"):
            if "_max_episode_steps" in cur.__dict__:
                return cur.__dict__["_max_episode_steps"]
            if "max_episode_steps" in cur.__dict__:
                return cur.__dict__["max_episode_steps"]
        spec = getattr(cur, "spec", None)
        if spec is not None and getattr(spec, "max_episode_steps", None) is not None:
            return spec.max_episode_steps
        cur = cur.__dict__.get("env") if hasattr(cur, "__dict__") else None
    return None
