"""BaseEnv: the env runtime that drives the native simulation core.

API counterpart of mani_skill/envs/sapien_env.py (constructor :185-327, reset :776-879, RNG rules
:881-917, step :943-972, substep loop `_step_action` :974-1025, get_info/get_obs/get_reward
:483-521, :603-616, :1039-1049, state get/set :1153-1199). Differences by design:
  * one batched scene, no per-env sub-scenes, always the tensor ("GPU sim") code path;
  * when neither the task nor the controller needs per-substep hooks, the
    `sim_freq // control_freq` substeps of one control step are issued as ONE native call;
  * no renderer: obs modes are "state", "state_dict", "none".
"""
import copy
import gc
from functools import cached_property
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import gymnasium as gym
import numpy as np
import torch
from gymnasium.vector.utils import batch_space

from maniskill_amd.agents import REGISTERED_AGENTS
from maniskill_amd.agents.base_agent import BaseAgent
from maniskill_amd.envs.scene import ManiSkillScene
from maniskill_amd.envs.utils.observations import parse_obs_mode_to_struct
from maniskill_amd.envs.utils.randomization.batched_rng import BatchedRNG
from maniskill_amd.envs.utils.system.backend import CPU_SIM_BACKENDS, parse_sim_and_render_backend
from maniskill_amd.utils import common, gym_utils
from maniskill_amd.utils.structs.actor import Actor
from maniskill_amd.utils.structs.articulation import Articulation
from maniskill_amd.utils.structs.pose import Pose
from maniskill_amd import physx
from maniskill_amd.utils.structs.types import SimConfig, strict_from_dict


class BaseEnv(gym.Env):
    SUPPORTED_ROBOTS: List[Union[str, Tuple[str]]] = None
    SUPPORTED_OBS_MODES = ("state", "state_dict", "none")
    SUPPORTED_REWARD_MODES = ("normalized_dense", "dense", "sparse", "none")
    SUPPORTED_RENDER_MODES = ("human", "rgb_array", "sensors", "all")
    metadata = {"render_modes": SUPPORTED_RENDER_MODES}

    scene: ManiSkillScene = None
    agent: BaseAgent = None
    _hidden_objects: List[Union[Actor, Articulation]] = []
    _main_rng: np.random.RandomState = None
    _batched_main_rng: BatchedRNG = None
    _main_seed: List[int] = None
    _episode_rng: np.random.RandomState = None
    _batched_episode_rng: BatchedRNG = None
    _episode_seed: np.ndarray = None
    _batched_rng_backend = "numpy:random_state"
    _enhanced_determinism: bool = False

    def __init__(
        self,
        num_envs: int = 1,
        obs_mode: Optional[str] = None,
        reward_mode: Optional[str] = None,
        control_mode: Optional[str] = None,
        render_mode: Optional[str] = None,
        shader_dir: Optional[str] = None,
        enable_shadow: bool = False,
        sensor_configs: Optional[dict] = None,
        human_render_camera_configs: Optional[dict] = None,
        viewer_camera_configs: Optional[dict] = None,
        robot_uids: Union[str, BaseAgent, List[Union[str, BaseAgent]]] = None,
        sim_config: Union[SimConfig, dict] = None,
        reconfiguration_freq: Optional[int] = None,
        sim_backend: str = "auto",
        render_backend: str = "gpu",
        parallel_in_single_scene: bool = False,
        enhanced_determinism: bool = False,
    ):
        self._enhanced_determinism = enhanced_determinism
        self._use_fused_callers = False  # set after the backend is known
        self.num_envs = num_envs
        self.reconfiguration_freq = reconfiguration_freq if reconfiguration_freq is not None else 0
        self._reconfig_counter = 0
        self._parallel_in_single_scene = parallel_in_single_scene
        self.robot_uids = robot_uids
        if isinstance(robot_uids, tuple) and len(robot_uids) == 1:
            self.robot_uids = robot_uids[0]

        if sim_backend == "auto":
            # the reference picks physx_cpu for num_envs == 1; this build has a single (HIP) backend
            sim_backend = "physx_cuda"
        self.backend = parse_sim_and_render_backend(sim_backend, render_backend)
        self.device = self.backend.device
        self._sim_device = self.backend.sim_device
        # fused native callers (action map, task epilogue) exist on the HIP backend only; the env
        # kwarg / env var lets tests force the plain torch path to compare the two
        import os as _os

        self._use_fused_callers = self.backend.sim_backend == "physx_cuda" and _os.environ.get("MS_FUSED", "1") != "0"
        if self.backend.sim_backend in CPU_SIM_BACKENDS and num_envs > 1:
            from maniskill_amd.physx.system import _BACKENDS

            if self.backend.sim_backend not in _BACKENDS:
                raise RuntimeError(
                    "Cannot set the sim backend to 'cpu' and have multiple environments. "
                    "(this build ships no CPU physics backend at all; sim_backend must be physx_cuda)"
                )

        sim_config = {} if sim_config is None else (sim_config.dict() if isinstance(sim_config, SimConfig) else sim_config)
        merged = self._default_sim_config.dict()
        common.dict_merge(merged, sim_config)
        self.sim_config: SimConfig = strict_from_dict(SimConfig, merged)

        if self.device.type == "cuda" and not physx.is_gpu_enabled():
            physx.enable_gpu()
        physx.set_gpu_memory_config(**self.sim_config.gpu_memory_config.dict())
        self._sim_freq = self.sim_config.sim_freq
        self._control_freq = self.sim_config.control_freq
        assert self._sim_freq % self._control_freq == 0, f"sim_freq({self._sim_freq}) is not divisible by control_freq({self._control_freq})."
        self._sim_steps_per_control = self._sim_freq // self._control_freq

        def choose(asked, supported, what, note=""):
            """the requested mode, the first supported one by default"""
            mode = supported[0] if asked is None else asked
            if mode not in supported:
                raise NotImplementedError(f"Unsupported {what} mode: {mode}. Must be one of {supported}{note}")
            return mode

        self._obs_mode = choose(obs_mode, self.SUPPORTED_OBS_MODES, "obs", " (this build has no renderer: state observations only)")
        self.obs_mode_struct = parse_obs_mode_to_struct(self._obs_mode)
        self._reward_mode = choose(reward_mode, self.SUPPORTED_REWARD_MODES, "reward")

        self._fused_action_key, self._fused_action_ok = None, False
        self._control_mode = control_mode
        if control_mode == "*":
            raise NotImplementedError("Multiple controllers are not supported yet.")
        self.render_mode = render_mode
        self._sensors = dict()

        self._main_seed = None
        from maniskill_amd.distributed import env_index_offset

        first = 2022 + env_index_offset()  # (a shard of a multi-GPU run counts its envs from its global offset)
        self._set_main_rng([first + i for i in range(self.num_envs)])
        self._elapsed_steps = torch.zeros(self.num_envs, device=self.device, dtype=torch.int32)
        obs, _ = self.reset(seed=[first + i for i in range(self.num_envs)], options=dict(reconfigure=True))

        # spaces: the action space is the agent's, the observation space is derived from the first observation
        self._init_raw_state = common.to_cpu_tensor(self.get_state_dict())
        self.action_space = None
        if self.agent is not None:
            self.action_space, self.single_action_space = self.agent.action_space, self.agent.single_action_space
            self._orig_single_action_space = copy.deepcopy(self.single_action_space)
        self.update_obs_space(common.to_cpu_tensor(obs))

    # ------------------------------------------------------------------ spaces
    def update_obs_space(self, obs):
        """(re)derive the observation spaces from a sample observation (wrappers that change the observation call this)"""
        self._init_raw_obs = obs
        for cached in ("single_observation_space", "observation_space"):
            self.__dict__.pop(cached, None)
        _ = self.single_observation_space, self.observation_space  # (evaluated eagerly, as the reference does)

    @cached_property
    def single_observation_space(self) -> gym.Space:
        return gym_utils.convert_observation_to_space(common.to_numpy(self._init_raw_obs), unbatched=True)

    @cached_property
    def observation_space(self) -> gym.Space:
        return batch_space(self.single_observation_space, n=self.num_envs)

    @property
    def gpu_sim_enabled(self):
        return self.scene.gpu_sim_enabled

    @property
    def _default_sim_config(self):
        return SimConfig()

    @property
    def _default_sensor_configs(self):
        return []

    @property
    def _default_human_render_camera_configs(self):
        return []

    # ------------------------------------------------------------------ loading
    def _load_agent(self, options: dict, initial_agent_poses=None, build_separate: bool = False):
        """one robot per env: `robot_uids` is a registered id, an agent class, a 1-tuple of either, "none" or None"""
        wanted = self.robot_uids
        self.agent = None
        if wanted is None or wanted == "none" or wanted == ("none",):
            return
        wanted = wanted if isinstance(wanted, tuple) else (wanted,)
        if len(wanted) > 1:
            raise NotImplementedError("multi-agent tasks are out of scope of this build (one articulation per env)")
        (which,) = wanted
        if isinstance(which, type) and issubclass(which, BaseAgent):
            agent_cls = which
        elif which in REGISTERED_AGENTS:
            agent_cls = REGISTERED_AGENTS[which].agent_cls
        else:
            raise RuntimeError(
                f"Agent {which} not found in the dict of registered agents. If the id is not a typo then make sure to apply the @register_agent() decorator."
            )
        poses = initial_agent_poses if isinstance(initial_agent_poses, list) else [initial_agent_poses]
        self.agent = agent_cls(self.scene, self._control_freq, self._control_mode, agent_idx=None, initial_pose=poses[0], build_separate=build_separate)

    def _load_scene(self, options: dict):
        pass

    def _load_lighting(self, options: dict):
        pass

    def _setup_sensors(self, options: dict):
        """cameras degrade to "none" in this build"""
        self._sensors = dict()
        self._sensor_configs = dict()

    def _after_reconfigure(self, options):
        pass

    # ------------------------------------------------------------------ read-only views of the configuration
    # (the reference spells each of these out as a property: sapien_env.py:444-489)
    sim_freq = property(lambda self: self._sim_freq)
    control_freq = property(lambda self: self._control_freq)
    sim_timestep = property(lambda self: 1.0 / self._sim_freq)
    control_timestep = property(lambda self: 1.0 / self._control_freq)
    obs_mode = property(lambda self: self._obs_mode)
    reward_mode = property(lambda self: self._reward_mode)
    elapsed_steps = property(lambda self: self._elapsed_steps)
    control_mode = property(lambda self: self.agent.control_mode)
    robot_link_names = property(lambda self: self.agent.robot_link_names)

    # ------------------------------------------------------------------ observations / reward
    def get_obs(self, info: Optional[Dict] = None):
        mode = self._obs_mode
        if mode == "none":
            return {}
        if mode not in ("state", "state_dict"):
            raise NotImplementedError(mode)
        nested = self._get_obs_state_dict(self.get_info() if info is None else info)
        if mode == "state":
            return common.flatten_state_dict(nested, use_torch=True, device=self.device)
        return common.torch_clone_dict(nested)  # (getters are views of the simulation buffers in this build: hand out copies)

    def _get_obs_state_dict(self, info: Dict):
        return dict(agent=self._get_obs_agent(), extra=self._get_obs_extra(info))

    def _get_obs_agent(self):
        return self.agent.get_proprioception()

    def _get_obs_extra(self, info: Dict):
        return dict()

    _REWARD_METHODS = {"sparse": "compute_sparse_reward", "dense": "compute_dense_reward", "normalized_dense": "compute_normalized_dense_reward"}

    def get_reward(self, obs: Any, action: torch.Tensor, info: Dict):
        mode = self._reward_mode
        if mode == "none":
            return torch.zeros((self.num_envs,), dtype=torch.float, device=self.device)
        if mode not in self._REWARD_METHODS:
            raise NotImplementedError(mode)
        return getattr(self, self._REWARD_METHODS[mode])(obs=obs, action=action, info=info)

    def compute_sparse_reward(self, obs: Any, action: torch.Tensor, info: Dict):
        """success - fail over whichever of the two the task reports (sapien_env.py:618-635); 0 if it reports neither"""
        won, lost = info.get("success"), info.get("fail")
        if won is not None and lost is not None:
            return won.to(torch.float) - lost.to(torch.float)
        if won is not None:
            return won
        if lost is not None:
            return -lost
        return torch.zeros(self.num_envs, dtype=torch.float, device=self.device)

    def compute_dense_reward(self, obs: Any, action: torch.Tensor, info: Dict):
        raise NotImplementedError()

    def compute_normalized_dense_reward(self, obs: Any, action: torch.Tensor, info: Dict):
        raise NotImplementedError()

    # ------------------------------------------------------------------ reconfigure / reset
    def _reconfigure(self, options=dict()):
        self._fused_action_key = None
        self._clear()
        self._setup_scene()
        self._load_agent(options)
        self._load_scene(options)
        self._load_lighting(options)
        self.scene._setup(enable_gpu=True)
        self._setup_sensors(options)
        self._reconfig_counter = self.reconfiguration_freq

    def _setup_scene(self):
        self.scene = ManiSkillScene(
            self.num_envs,
            sim_config=self.sim_config,
            device=self.device,
            backend_name=self.backend.sim_backend,
            parallel_in_single_scene=self._parallel_in_single_scene,
        )
        self.scene.px.timestep = 1.0 / self._sim_freq

    def _clear(self):
        if self.scene is not None:
            self.scene.px.close()
        self.agent = None
        self._sensors = dict()
        self.scene = None
        self._hidden_objects = []
        gc.collect()

    def close(self):
        self._clear()

    def reset(self, seed: Union[None, int, List[int]] = None, options: Union[None, dict] = None):
        """full reset, or a partial one (`options["env_idx"]`); `options["reconfigure"]` rebuilds the scene first.
        Phases: request -> seeding (+ optional reconfiguration) -> physics state of the selected envs -> outputs.
        Behavioural contract: mani_skill/envs/sapien_env.py:776-879."""
        options = {} if options is None else options
        env_idx, partial, reconfigure = self._reset_request(options)
        if env_idx is None:  # an empty partial reset: the current observation, no episode is touched
            info = self.get_info()
            info["reconfigure"] = False
            return self.get_obs(info), info
        self._seed_reset(seed, env_idx, reconfigure, options)
        self._reset_selected_envs(seed, env_idx, partial, options)
        obs, info = self._reset_outputs()
        info["reconfigure"] = reconfigure
        return obs, info

    def _reset_request(self, options: dict):
        """-> (env indices to reset or None for "nothing to do", partial?, reconfigure?)"""
        reconfigure = bool(options.get("reconfigure", False)) or (self._reconfig_counter == 0 and self.reconfiguration_freq != 0)
        if "env_idx" not in options:
            return torch.arange(0, self.num_envs, device=self.device), False, reconfigure
        env_idx = common.to_tensor(options["env_idx"], device=self.device).long()
        if reconfigure and len(env_idx) != self.num_envs:
            raise RuntimeError("Cannot do a partial reset and reconfigure the environment. You must do one or the other.")
        if len(env_idx) == 0 and not reconfigure:
            return None, True, False
        return env_idx, True, reconfigure

    def _seed_reset(self, seed, env_idx, reconfigure: bool, options: dict):
        self._set_main_rng(seed)
        if not reconfigure:
            self._set_episode_rng(seed, env_idx)
            return
        # a reconfiguration is seeded like an episode (its own torch stream), and the episode that follows starts from
        # the same seeds again
        self._set_episode_rng(seed if seed is not None else self._batched_main_rng.randint(2**31), env_idx)
        with torch.random.fork_rng():
            torch.manual_seed(seed=int(self._episode_seed[0]))
            self._reconfigure(options)
            self._after_reconfigure(options)
        self._set_episode_rng(self._episode_seed, env_idx)

    def _reset_selected_envs(self, seed, env_idx, partial: bool, options: dict):
        scene = self.scene
        scene._set_reset_idx(env_idx if partial else None)  # every struct setter below writes the selected envs only
        self._elapsed_steps[env_idx] = 0
        self._clear_sim_state()
        if self.reconfiguration_freq != 0:
            self._reconfig_counter -= 1
        if self.agent is not None:
            self.agent.reset()
        self._run_initialize_episode(seed, env_idx, options)
        scene._set_reset_idx(None)
        scene._gpu_apply_all()
        # the envs being reset start from their written state alone: sleep counters, cached manifolds and multipliers of the
        # episode that ended are dropped (PhysX: the setters wake the body and its cached contacts go with the teleport)
        if partial:
            scene.px.wake_envs(env_idx)
        else:
            scene.px.wake_all()  # (every env: three fills instead of a strided write per env and pair)
        scene.px.gpu_update_articulation_kinematics()
        scene._gpu_fetch_all()
        if self.agent is not None:
            self.agent.controller.reset()  # (under the all-envs mask, as the reference: sapien_env.py:857-871)

    def _run_initialize_episode(self, seed, env_idx, options: dict):
        """`_initialize_episode` under the torch RNG rule of the reset: per-env streams (enhanced determinism), one
        stream seeded with env 0's episode seed (explicit seed), or the ambient stream"""
        if self._enhanced_determinism:
            # every env draws from its own torch stream (seeded with its episode seed), as it does from its own numpy
            # stream: env e gets the same episode in any batch, shard or partial reset
            from maniskill_amd.envs.utils.randomization.per_env_torch import PerEnvTorchRNG

            with torch.random.fork_rng():
                torch.manual_seed(int(self._episode_seed[0]))
                with PerEnvTorchRNG(self._episode_seed[common.to_numpy(env_idx)]):
                    self._initialize_episode(env_idx, options)
        elif seed is not None:
            with torch.random.fork_rng():
                torch.manual_seed(int(self._episode_seed[0]))
                self._initialize_episode(env_idx, options)
        else:
            self._initialize_episode(env_idx, options)

    def _reset_outputs(self):
        """obs / info of the state the reset produced (the native epilogue without advancing the step counter where the
        task has one: same values, see the GPU tests)"""
        fused = self._fused_step_outputs(None, advance=False) if (self._use_fused_callers and self._fused_ok()) else None
        if fused is not None:
            obs, _, info = fused
            return obs, info
        info = self.get_info()
        return self.get_obs(info), info

    # RNG rules (reference: sapien_env.py:881-917), expressed through BatchedRNG: a main stream per env, seeded once per
    # explicit seed; an episode stream per env, reseeded at every reset -- from the given seed(s), or (enhanced
    # determinism) from the env's own main stream, otherwise left running
    def _set_main_rng(self, seed):
        if seed is None:
            if self._main_seed is not None:
                return
            seed = np.random.RandomState().randint(2**31, size=(self.num_envs,))
        seeds = BatchedRNG.expand_seeds(seed, self.num_envs)
        self._main_seed = seeds.tolist()
        self._main_rng = np.random.RandomState(self._main_seed[0])
        self._batched_main_rng = BatchedRNG.from_seeds(seeds, backend=self._batched_rng_backend)

    def _set_episode_rng(self, seed, env_idx: torch.Tensor):
        if seed is None and not self._enhanced_determinism:
            return
        rows = common.to_numpy(env_idx)
        if seed is not None:
            self._episode_seed = BatchedRNG.expand_seeds(seed, self.num_envs)
            self._batched_episode_rng = BatchedRNG.from_seeds(self._episode_seed, backend=self._batched_rng_backend)
        else:
            self._episode_seed[rows] = self._batched_main_rng[rows].randint(2**31)
            if self._batched_episode_rng is None:
                self._batched_episode_rng = BatchedRNG.from_seeds(self._episode_seed, backend=self._batched_rng_backend)
            else:
                self._batched_episode_rng.reseed(rows, self._episode_seed[rows])
        self._episode_rng = self._batched_episode_rng[0]

    def _initialize_episode(self, env_idx: torch.Tensor, options: dict):
        pass

    def _clear_sim_state(self):
        """zero velocities of the envs being reset (sapien_env.py:924-937)"""
        for actor in self.scene.actors.values():
            if actor.px_body_type == "dynamic":
                actor.set_linear_velocity(torch.zeros(3, device=self.device))
                actor.set_angular_velocity(torch.zeros(3, device=self.device))
        for art in self.scene.articulations.values():
            art.set_qvel(torch.zeros(art.max_dof, device=self.device))
            art.set_root_linear_velocity(torch.zeros(3, device=self.device))
            art.set_root_angular_velocity(torch.zeros(3, device=self.device))
        self.scene._gpu_apply_all()
        self.scene._gpu_fetch_all()

    # ------------------------------------------------------------------ step
    def step(self, action: Union[None, np.ndarray, torch.Tensor, Dict]):
        self._fused_truncated = self._fused_terminated = None
        self._fused_epilogue_next = self._use_fused_callers and self._fused_ok()
        action = self._step_action(action)
        self._fused_epilogue_next = False
        # the fused task epilogue also advances `_elapsed_steps` (one launch instead of add + copy)
        fused = self._fused_step_outputs(action) if self._use_fused_callers else None
        if fused is not None:
            obs, reward, info = fused
            # `terminated` is the epilogue's own copy of `success` (the reference returns a clone, sapien_env.py:959:
            # ManiSkillVectorEnv(ignore_terminations=True) clears it in place and must not clear info["success"])
            terminated = self._fused_terminated.view(torch.bool) if self._fused_terminated is not None else info["success"].clone()
            return obs, reward, terminated, self._no_truncation(), info
        self._elapsed_steps += 1
        info = self.get_info()
        obs = self.get_obs(info)
        reward = self.get_reward(obs=obs, action=action, info=info)
        if "success" in info:
            terminated = torch.logical_or(info["success"], info["fail"]) if "fail" in info else info["success"].clone()
        elif "fail" in info:
            terminated = info["fail"].clone()
        else:
            terminated = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        return obs, reward, terminated, torch.zeros(self.num_envs, dtype=torch.bool, device=self.device), info

    def _no_truncation(self) -> torch.Tensor:
        """all-False `truncated` of the fused step path (time limits are applied by TimeLimitWrapper);
        one cached tensor instead of a fill kernel per step -- treat it as read-only"""
        t = getattr(self, "_false_flags", None)
        if t is None or t.shape[0] != self.num_envs or t.device != self.device:
            t = self._false_flags = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        return t

    def _substep_hooks_overridden(self) -> bool:
        cls = type(self)
        return cls._before_simulation_step is not BaseEnv._before_simulation_step or cls._after_simulation_step is not BaseEnv._after_simulation_step

    def _step_action(self, action):
        # what was handed in: nothing (keep the drive targets), an array, or {"control_mode": ..., "action": array}
        set_action = action is not None
        if isinstance(action, dict):
            if "control_mode" not in action:
                raise NotImplementedError("dict actions are for multi-agent tasks, which this build does not include")
            if self.agent.control_mode != action["control_mode"]:
                self.agent.set_control_mode(action["control_mode"])
                self.agent.controller.reset()
            action = action["action"]
        if set_action:
            if not isinstance(action, (np.ndarray, torch.Tensor)):
                raise TypeError(type(action))
            action = common.to_tensor(action, device=self.device)
            if self.num_envs == 1 and action.shape == self._orig_single_action_space.shape:
                action = common.batch(action)  # a single env accepts an unbatched action
            if self._fused_action_ready(action):
                # (the torch path asserts this in BaseController._preprocess_action; the kernels index the action unchecked)
                assert action.shape == (self.num_envs, self.single_action_space.shape[0]), (
                    f"action of shape {tuple(action.shape)} for an action space of shape ({self.num_envs}, {self.single_action_space.shape[0]})")
                # one launch instead of the controller's ~10 torch ops + 2 applies (same arithmetic); without
                # hooks between set_action and the substeps the map runs at the head of the step's own launch
                if self._no_step_hooks():
                    # `step` runs the task's fused epilogue next: the step and the copy-out are owed to that call,
                    # which then launches the whole control step as one kernel
                    self.scene.px.step_action(action, self._sim_steps_per_control, defer=self._fused_epilogue_next)
                    self.scene._gpu_fetch_all(defer=self._fused_epilogue_next)
                    return action
                self.scene.px.apply_action(action)
            else:
                self.agent.set_action(action)
                self.scene.px.gpu_apply_articulation_target_position()
                self.scene.px.gpu_apply_articulation_target_velocity()
        self._before_control_step()
        per_substep = self._substep_hooks_overridden() or (
            self.agent is not None and getattr(self.agent.controller, "needs_per_substep_update", False)
        )
        if per_substep:
            for _ in range(self._sim_steps_per_control):
                if self.agent is not None:
                    self.agent.before_simulation_step()
                self._before_simulation_step()
                self.scene.step()
                self._after_simulation_step()
        else:
            # fused substep loop (sapien_env.py:1016-1021 issues these one px.step() at a time)
            self.scene.step(self._sim_steps_per_control)
        self._after_control_step()
        self.scene._gpu_fetch_all()
        return action

    def _no_step_hooks(self) -> bool:
        """no task / controller code has to run between setting the action and the end of the control step"""
        ok = self.__dict__.get("_no_step_hooks_cached")
        if ok is None:
            cls = type(self)
            ok = (
                cls._before_control_step is BaseEnv._before_control_step
                and cls._after_control_step is BaseEnv._after_control_step
                and not self._substep_hooks_overridden()
                and hasattr(self.scene.px, "step_action")
            )
            self.__dict__["_no_step_hooks_cached"] = ok
        return ok and not getattr(self.agent.controller, "needs_per_substep_update", False)

    def _fused_action_ready(self, action) -> bool:
        """native affine action->target map usable for the current controller? (HIP backend only)"""
        if not self._use_fused_callers or not isinstance(action, torch.Tensor) or action.dtype != torch.float32 or action.dim() != 2:
            return False
        ctrl = self.agent.controller
        key = (id(ctrl), self.agent.control_mode)
        if self._fused_action_key != key:
            spec = getattr(ctrl, "fused_action_spec", lambda: None)()
            self._fused_action_key, self._fused_action_ok = key, spec is not None
            if spec is not None:
                self.scene.px.set_action_map(*spec[:4])
                self.scene.px.set_ee_action_map(spec[4])
        return self._fused_action_ok and action.is_contiguous()

    _fused_epilogue_next = False
    _time_limit = None  # set by TimeLimitWrapper: the fused epilogue then also writes `elapsed_steps >= limit`
    _fused_truncated = None

    def _fused_time_limit_out(self):
        """(device pointer, limit) for the task struct of a fused epilogue; (None, 0) without a time limit"""
        if self._time_limit is None:
            self._fused_truncated = None
            return None, 0
        self._fused_truncated = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        return self._fused_truncated.data_ptr(), int(self._time_limit)

    _fused_terminated = None

    def _fused_ok(self) -> bool:
        """will `_fused_step_outputs` produce this step's outputs? Asked every step; the task's answer
        (`_fused_task_ok`) is cached per controller object and hidden-object state: a control-mode switch can add
        controller state to the observation, and the epilogue reads the raw pose rows, which a hidden object has moved
        away (Actor.hide_visual)."""
        ctrl = self.agent.controller if self.agent is not None else None
        hidden = any(getattr(o, "hidden", False) for o in self._hidden_objects)
        key = (id(self.scene), id(ctrl), getattr(self.agent, "control_mode", None), hidden)
        if self.__dict__.get("_fused_ok_key") != key:
            self._fused_ok_key, self._fused_ok_val = key, (not hidden) and bool(self._fused_task_ok())
        return self._fused_ok_val

    def _fused_task_ok(self) -> bool:
        """tasks with a native epilogue: is it equivalent to the torch path for this configuration?"""
        return False

    def _fused_bind_counters(self, task, advance: bool) -> torch.Tensor:
        """per-step pointers of a native task struct: the step counter (incremented in place + copied out), the time
        limit flag and the `terminated` copy. Returns the tensor that becomes info["elapsed_steps"].
        `advance=False` (reset): outputs of the current state, the counter stays where it is."""
        if advance:
            es = torch.empty_like(self._elapsed_steps)
            task.elapsed_steps, task.elapsed_out = self._elapsed_steps.data_ptr(), es.data_ptr()
            task.truncated_out, task.time_limit = self._fused_time_limit_out()
            self._fused_terminated = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
            task.terminated_out = self._fused_terminated.data_ptr()
        else:
            es = self._elapsed_steps.clone()
            task.elapsed_steps = task.elapsed_out = task.truncated_out = task.terminated_out = None
            task.time_limit = 0
            self._fused_terminated = None
        return es

    def _fused_step_outputs(self, action, advance: bool = True):
        """tasks may return (obs, reward, info) computed by a fused native kernel; None = torch path.
        `advance=False` (reset): the outputs of the current state without advancing `elapsed_steps`"""
        return None

    def evaluate(self) -> dict:
        """task-specific success / failure flags and whatever else the reward needs; nothing by default"""
        return {}

    def get_info(self) -> dict:
        return {"elapsed_steps": self._elapsed_steps.clone(), **self.evaluate()}

    def _no_hook(self):
        """default of the four step hooks; `_no_step_hooks` / `_substep_hooks_overridden` compare against these attributes"""

    _before_control_step = _after_control_step = _before_simulation_step = _after_simulation_step = _no_hook

    # ------------------------------------------------------------------ state (sapien_env.py:1153-1199)
    def add_to_state_dict_registry(self, obj):
        self.scene.add_to_state_dict_registry(obj)

    def remove_from_state_dict_registry(self, obj):
        self.scene.remove_from_state_dict_registry(obj)

    def get_state_dict(self):
        return self.scene.get_sim_state()

    def get_state(self):
        return common.flatten_state_dict(self.get_state_dict(), use_torch=True)

    def set_state_dict(self, state: Dict, env_idx: torch.Tensor = None):
        self.scene.set_sim_state(state, env_idx)
        self.scene._gpu_apply_all()
        # a restored state starts awake: what follows depends on the state alone, not on how long the bodies had been at
        # rest before (PhysX: setGlobalPose wakes the actor)
        self.scene.px.wake_all()
        self.scene.px.gpu_update_articulation_kinematics()
        self.scene._gpu_fetch_all()

    def set_state(self, state, env_idx: torch.Tensor = None):
        state = common.to_tensor(state, device=self.device)
        sd = dict(actors=dict(), articulations=dict())
        start = 0
        for actor_id in self._init_raw_state.get("actors", {}).keys():
            sd["actors"][actor_id] = state[:, start : start + 13]
            start += 13
        for art_id, art_state in self._init_raw_state.get("articulations", {}).items():
            size = art_state.shape[-1]
            sd["articulations"][art_id] = state[:, start : start + size]
            start += size
        self.set_state_dict(sd, env_idx)

    # ------------------------------------------------------------------ misc
    def render(self):
        raise NotImplementedError("rendering is out of scope of this build (state observations only)")

    def print_sim_details(self):
        print("# -------------------------------------------------------------------------- #")
        print(f"Task ID: {getattr(self.spec, 'id', type(self).__name__)}, {self.num_envs} parallel environments, sim_backend={self.backend.sim_backend}")
        print(f"obs_mode={self.obs_mode}, control_mode={self.control_mode}")
        print(f"sim_freq={self.sim_freq}, control_freq={self.control_freq}")
        print(f"observation space: {self.observation_space}")
        print(f"(single) action space: {self.single_action_space}")
        print("# -------------------------------------------------------------------------- #")
