"""Minimal gymnasium-compatible subset (gymnasium 0.29 API names the reference uses).

Env / Wrapper / ObservationWrapper / ActionWrapper / RewardWrapper, spaces.Box / Dict / Discrete,
vector.VectorEnv + vector.utils.batch_space, register / make / registry with EnvSpec /
WrapperSpec and `additional_wrappers`, wrappers.TimeLimit. Only what
mani_skill/utils/registration.py:127-260, mani_skill/vector/wrappers/gymnasium.py:16-173 and
mani_skill/utils/gym_utils.py touch.
"""
import copy
import importlib
import sys
import types
from dataclasses import dataclass, field
from typing import Any, Callable, Dict as TDict, List, Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------ spaces
class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)
        return [seed]

    @property
    def np_random(self):
        return self._np_random

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        shape = tuple(int(s) for s in shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
        super().__init__(shape, dtype, seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e6)
        hi = np.where(np.isfinite(self.high), self.high, 1e6)
        if np.issubdtype(self.dtype, np.floating):
            return self._np_random.uniform(lo, hi, size=self.shape).astype(self.dtype)
        return self._np_random.integers(lo, hi, size=self.shape, endpoint=True).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, o):
        return isinstance(o, Box) and self.shape == o.shape and np.allclose(self.low, o.low) and np.allclose(self.high, o.high)


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n, self.start = int(n), int(start)
        super().__init__((), np.int64, seed)

    def sample(self):
        return int(self.start + self._np_random.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)
        super().__init__(None, None, seed)

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def values(self):
        return self.spaces.values()

    def __len__(self):
        return len(self.spaces)

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k!r}: {v}" for k, v in self.spaces.items()) + ")"


def batch_space(space, n=1):
    if isinstance(space, Box):
        rep = (n,) + (1,) * len(space.shape)
        return Box(np.tile(space.low, rep), np.tile(space.high, rep), dtype=space.dtype)
    if isinstance(space, Dict):
        return Dict({k: batch_space(s, n) for k, s in space.spaces.items()})
    if isinstance(space, Discrete):
        return Box(space.start, space.start + space.n - 1, shape=(n,), dtype=np.int64)
    raise NotImplementedError(type(space))


# ------------------------------------------------------------------------------------------- core
class Env:
    metadata: TDict[str, Any] = {"render_modes": []}
    render_mode = None
    spec = None
    action_space: Space = None
    observation_space: Space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def render(self):
        return None

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self._action_space = None
        self._observation_space = None

    def __getattr__(self, name):
        if name.startswith("_") and name not in ("_max_episode_steps",):
            raise AttributeError(f"accessing private attribute '{name}' is prohibited")
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def action_space(self):
        return self._action_space if self._action_space is not None else self.env.action_space

    @action_space.setter
    def action_space(self, s):
        self._action_space = s

    @property
    def observation_space(self):
        return self._observation_space if self._observation_space is not None else self.env.observation_space

    @observation_space.setter
    def observation_space(self, s):
        self._observation_space = s

    @property
    def render_mode(self):
        return self.env.render_mode

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()


class ObservationWrapper(Wrapper):
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observation(obs), info

    def step(self, action):
        obs, r, te, tr, info = self.env.step(action)
        return self.observation(obs), r, te, tr, info

    def observation(self, observation):
        raise NotImplementedError


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))

    def action(self, action):
        raise NotImplementedError


class RewardWrapper(Wrapper):
    def step(self, action):
        obs, r, te, tr, info = self.env.step(action)
        return obs, self.reward(r), te, tr, info

    def reward(self, reward):
        raise NotImplementedError


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed = 0

    def step(self, action):
        obs, r, te, tr, info = self.env.step(action)
        self._elapsed += 1
        if self._elapsed >= self._max_episode_steps:
            tr = True
        return obs, r, te, tr, info

    def reset(self, **kw):
        self._elapsed = 0
        return self.env.reset(**kw)


class VectorEnv(Env):
    def __init__(self, num_envs=None, observation_space=None, action_space=None):
        if num_envs is not None:
            self.num_envs = num_envs
        if observation_space is not None:
            self.single_observation_space = observation_space
            self.observation_space = batch_space(observation_space, num_envs)
        if action_space is not None:
            self.single_action_space = action_space
            self.action_space = batch_space(action_space, num_envs)
        self.is_vector_env = True
        self.closed = False

    def close_extras(self, **kw):
        pass

    def close(self, **kw):
        if not self.closed:
            self.close_extras(**kw)
            self.closed = True


# --------------------------------------------------------------------------------------- registry
@dataclass
class WrapperSpec:
    name: str
    entry_point: str
    kwargs: Optional[TDict[str, Any]] = None


@dataclass
class EnvSpec:
    id: str
    entry_point: Any = None
    reward_threshold: Optional[float] = None
    nondeterministic: bool = False
    max_episode_steps: Optional[int] = None
    order_enforce: bool = True
    autoreset: bool = False
    disable_env_checker: bool = False
    apply_api_compatibility: bool = False
    kwargs: TDict[str, Any] = field(default_factory=dict)
    additional_wrappers: Tuple[WrapperSpec, ...] = ()
    vector_entry_point: Any = None
    pass_max_episode_steps: bool = False  # entry point applies its own (batched) time limit


registry: TDict[str, EnvSpec] = {}


def register(id, entry_point=None, max_episode_steps=None, disable_env_checker=True, kwargs=None, additional_wrappers=(), **extra):
    registry[id] = EnvSpec(
        id=id,
        entry_point=entry_point,
        max_episode_steps=max_episode_steps,
        disable_env_checker=disable_env_checker,
        kwargs=dict(kwargs or {}),
        additional_wrappers=tuple(additional_wrappers),
        pass_max_episode_steps=bool(extra.get("pass_max_episode_steps", False)),
    )


def _load(entry_point):
    if callable(entry_point):
        return entry_point
    mod, attr = entry_point.split(":")
    obj = importlib.import_module(mod)
    for part in attr.split("."):
        obj = getattr(obj, part)
    return obj


def make(id, max_episode_steps=None, disable_env_checker=None, **kwargs):
    if isinstance(id, EnvSpec):
        spec = id
    else:
        if id not in registry:
            raise KeyError(f"Environment `{id}` is not registered (did you import the task module?)")
        spec = registry[id]
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    if spec.pass_max_episode_steps and max_episode_steps is not None:
        kw["max_episode_steps"] = max_episode_steps
    env = _load(spec.entry_point)(**kw)
    spec_copy = copy.copy(spec)
    spec_copy.kwargs = kw
    try:
        env.unwrapped.spec = spec_copy
    except Exception:
        pass
    for ws in spec.additional_wrappers:
        env = _load(ws.entry_point)(env, **(ws.kwargs or {}))
    steps = max_episode_steps if max_episode_steps is not None else spec.max_episode_steps
    if steps is not None and not spec.additional_wrappers and not spec.pass_max_episode_steps:
        env = TimeLimit(env, steps)
    return env


def spec(id):
    return registry[id]


def install_as(name):
    this = sys.modules[__name__]
    root = types.ModuleType(name)
    for k in ("Env", "Wrapper", "ObservationWrapper", "ActionWrapper", "RewardWrapper", "Space", "register", "make", "spec", "registry"):
        setattr(root, k, getattr(this, k))
    spaces = types.ModuleType(name + ".spaces")
    for k in ("Space", "Box", "Dict", "Discrete"):
        setattr(spaces, k, getattr(this, k))
    vector = types.ModuleType(name + ".vector")
    vector.VectorEnv = VectorEnv
    vutils = types.ModuleType(name + ".vector.utils")
    vutils.batch_space = batch_space
    vector.utils = vutils
    envs = types.ModuleType(name + ".envs")
    reg = types.ModuleType(name + ".envs.registration")
    reg.EnvSpec, reg.WrapperSpec, reg.registry, reg.register, reg.make = EnvSpec, WrapperSpec, registry, register, make
    envs.registration = reg
    wrappers = types.ModuleType(name + ".wrappers")
    wrappers.TimeLimit = TimeLimit
    core = types.ModuleType(name + ".core")
    core.Env, core.Wrapper = Env, Wrapper
    root.spaces, root.vector, root.envs, root.wrappers, root.core = spaces, vector, envs, wrappers, core
    root.__maniskill_amd_shim__ = True
    for m in (root, spaces, vector, vutils, envs, reg, wrappers, core):
        sys.modules[m.__name__] = m
