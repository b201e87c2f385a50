"""Controller base classes (counterpart of mani_skill/agents/controllers/base_controller.py:25-330).
A controller maps a (normalised) action slice to joint drive targets written into
`px.cuda_articulation_target_qpos/qvel`."""
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch
from gymnasium import spaces
from gymnasium.vector.utils import batch_space

from maniskill_amd.agents.utils import flatten_action_spaces, get_active_joint_indices, get_joints_by_names
from maniskill_amd.utils import common, gym_utils


class BaseController:
    def __init__(self, config: "ControllerConfig", articulation, control_freq: int, sim_freq: int = None, scene=None):
        self.config = config
        self.articulation = articulation
        self._control_freq = control_freq
        self.scene = scene
        if sim_freq is None:
            sim_freq = round(1.0 / self.articulation.px.timestep)
        self._sim_steps = sim_freq // control_freq
        self._initialize_joints()
        self._initialize_action_space()
        self._normalize_action = getattr(self.config, "normalize_action", False)
        if self._normalize_action:
            self._clip_and_scale_action_space()
        self.action_space = self.single_action_space
        if self.scene.num_envs > 1:
            self.action_space = batch_space(self.single_action_space, n=self.scene.num_envs)

    @property
    def device(self):
        return self.articulation.device

    def _initialize_joints(self):
        names = self.config.joint_names
        try:
            self.joints = get_joints_by_names(self.articulation, names)
            self.active_joint_indices = get_active_joint_indices(self.articulation, names).to(self.device)
        except Exception:
            print("Joint names of the articulation", [j.name for j in self.articulation.get_active_joints()])
            print("Joint names of the controller", names)
            raise
        idx = self.active_joint_indices.tolist()
        # contiguous joint ranges index as slices (views): no gather kernel, no copy
        self._cols = slice(idx[0], idx[-1] + 1) if idx == list(range(idx[0], idx[-1] + 1)) else self.active_joint_indices.long()

    def _initialize_action_space(self):
        raise NotImplementedError

    @property
    def control_freq(self):
        return self._control_freq

    @property
    def qpos(self):
        return self.articulation.get_qpos()[..., self._cols]

    @property
    def qvel(self):
        return self.articulation.get_qvel()[..., self._cols]

    def set_drive_property(self):
        raise NotImplementedError

    def reset(self):
        pass

    def _preprocess_action(self, action):
        action_dim = self.action_space.shape[1] if self.scene.num_envs > 1 else self.action_space.shape[0]
        assert action.shape == (self.scene.num_envs, action_dim), (action.shape, action_dim)
        if self._normalize_action:
            action = self._clip_and_scale_action(action)
        return action

    def set_action(self, action):
        raise NotImplementedError

    def before_simulation_step(self):
        pass

    def get_state(self) -> dict:
        return {}

    def set_state(self, state: dict):
        pass

    def _clip_and_scale_action_space(self):
        self._original_single_action_space = self.single_action_space
        self.single_action_space = gym_utils.normalize_action_space(self._original_single_action_space)
        self.action_space_low = common.to_tensor(self._original_single_action_space.low, device=self.device)
        self.action_space_high = common.to_tensor(self._original_single_action_space.high, device=self.device)

    def _clip_and_scale_action(self, action):
        return gym_utils.clip_and_scale_action(action, self.action_space_low, self.action_space_high)


@dataclass
class ControllerConfig:
    joint_names: List[str]
    controller_cls = BaseController


class DictController(BaseController):
    def __init__(self, configs: Dict[str, ControllerConfig], articulation, control_freq: int, sim_freq: int = None, scene=None):
        self.scene = scene
        self.configs = configs
        self.articulation = articulation
        self._control_freq = control_freq
        self.controllers: Dict[str, BaseController] = {}
        for uid, cfg in configs.items():
            self.controllers[uid] = cfg.controller_cls(cfg, articulation, control_freq, sim_freq=sim_freq, scene=scene)
        self._initialize_action_space()
        self._initialize_joints()
        self.action_space = self.single_action_space
        if self.scene.num_envs > 1:
            self.action_space = batch_space(self.single_action_space, n=self.scene.num_envs)

    def before_simulation_step(self):
        for c in self.controllers.values():
            c.before_simulation_step()

    @property
    def needs_per_substep_update(self):
        return any(getattr(c.config, "interpolate", False) for c in self.controllers.values())

    def _initialize_action_space(self):
        self.single_action_space = spaces.Dict({uid: c.single_action_space for uid, c in self.controllers.items()})

    def _initialize_joints(self):
        self.joints, self.active_joint_indices = [], []
        for c in self.controllers.values():
            self.joints.extend(c.joints)
            self.active_joint_indices.extend(c.active_joint_indices)

    def set_drive_property(self):
        for c in self.controllers.values():
            c.set_drive_property()

    def reset(self):
        for c in self.controllers.values():
            c.reset()

    def set_action(self, action: Dict[str, np.ndarray]):
        for uid, c in self.controllers.items():
            c.set_action(action[uid])

    def get_state(self) -> dict:
        out = {}
        for uid, c in self.controllers.items():
            s = c.get_state()
            if len(s) > 0:
                out[uid] = s
        return out

    def set_state(self, state: dict):
        for uid, c in self.controllers.items():
            if state is not None and uid in state:
                c.set_state(state[uid])

    def from_qpos(self, qpos):
        qpos = common.to_tensor(qpos, device=self.device)
        out, start = [], 0
        for c in self.controllers.values():
            nd, nj = c.single_action_space.shape[0], len(c.joints)
            out.append(qpos[..., start : start + nd])
            start += nj
        return torch.concat(out, dim=-1)


class CombinedController(DictController):
    """flat action = concatenation of the sub-controllers' actions in dict order"""

    def _initialize_action_space(self):
        super()._initialize_action_space()
        self.single_action_space, self.action_mapping = flatten_action_spaces(self.single_action_space.spaces)

    def set_action(self, action):
        action_dim = self.action_space.shape[1] if self.scene.num_envs > 1 else self.action_space.shape[0]
        assert action.shape == (self.scene.num_envs, action_dim), (
            f"Received action of shape {action.shape} but expected shape ({self.scene.num_envs}, {action_dim})"
        )
        for uid, c in self.controllers.items():
            s, e = self.action_mapping[uid]
            c.set_action(action[:, s:e])

    def fused_action_spec(self):
        """per-dof (column, low, high, flags) arrays for `px.set_action_map`, or None if any
        sub-controller cannot be expressed as an affine action -> target map"""
        n = self.articulation.max_dof
        col, lo, hi, fl = [-1] * n, [0.0] * n, [0.0] * n, [0] * n
        ee = None  # end-effector block: (link index, first action column, rows, low, high, rot_scale, flags)
        for uid, c in self.controllers.items():
            spec = getattr(c, "fused_action_spec", lambda: None)()
            if spec is None:
                return None
            start, _ = self.action_mapping[uid]
            if isinstance(spec, dict):
                if ee is not None:
                    return None  # one end-effector block
                link, rows, l, h, rs, f = spec["ee"]
                ee = (link, start, rows, l, h, rs, f)
                for dof in spec["dofs"]:
                    fl[dof] = 4  # driven by the end-effector block
                continue
            for dof, lcol, l, h, f in spec:
                col[dof], lo[dof], hi[dof], fl[dof] = start + lcol, l, h, f
        return col, lo, hi, fl, ee

    def to_action_dict(self, action):
        return {uid: action[s:e] for uid, (s, e) in self.action_mapping.items()}

    def from_action_dict(self, action_dict: dict):
        return torch.hstack([action_dict[uid] for uid in self.controllers])
