"""BaseAgent: loads the robot articulation and owns its controllers (counterpart of
mani_skill/agents/base_agent.py:44-390)."""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from gymnasium import spaces

from maniskill_amd.agents.controllers import CombinedController, ControllerConfig, PDJointPosControllerConfig
from maniskill_amd.agents.controllers.base_controller import BaseController
from maniskill_amd.utils import sapien_utils
from maniskill_amd.utils.structs.articulation import Articulation
from maniskill_amd.utils.structs.pose import Pose

DictControllerConfig = Dict[str, ControllerConfig]


@dataclass
class Keyframe:
    pose: object
    qpos: Optional[np.ndarray] = None
    qvel: Optional[np.ndarray] = None


class BaseAgent:
    uid: str
    urdf_path: Union[str, None] = None
    urdf_config: Union[str, Dict] = None
    mjcf_path: Union[str, None] = None
    fix_root_link: bool = True
    load_multiple_collisions: bool = False
    disable_self_collisions: bool = False
    keyframes: Dict[str, Keyframe] = dict()

    def __init__(self, scene, control_freq: int, control_mode: Optional[str] = None, agent_idx: Optional[str] = None,
                 initial_pose=None, build_separate: bool = False):
        self.scene = scene
        self._control_freq = control_freq
        self._agent_idx = agent_idx
        if build_separate:
            raise NotImplementedError("build_separate (per-env distinct robots) is not supported by this core yet")
        self.build_separate = build_separate
        self.robot: Articulation = None
        self.controllers: Dict[str, BaseController] = dict()
        self.sensors = dict()
        self._load_articulation(initial_pose)
        self._after_loading_articulation()
        self.supported_control_modes = list(self._controller_configs.keys())
        if control_mode is None:
            control_mode = self.supported_control_modes[0]
        self._default_control_mode = control_mode
        self.set_control_mode()
        self._after_init()

    @property
    def _sensor_configs(self):
        return []

    @property
    def _controller_configs(self) -> Dict[str, Union[ControllerConfig, DictControllerConfig]]:
        names = [j.name for j in self.robot.active_joints]
        return dict(
            pd_joint_pos=PDJointPosControllerConfig(names, lower=None, upper=None, stiffness=100, damping=10, normalize_action=False),
            pd_joint_delta_pos=PDJointPosControllerConfig(names, lower=-0.1, upper=0.1, stiffness=100, damping=10, normalize_action=True, use_delta=True),
        )

    @property
    def device(self):
        return self.scene.device

    def _load_articulation(self, initial_pose=None):
        if self.urdf_path is None:
            raise NotImplementedError("only URDF robots are supported (MJCF import is out of scope)")
        loader = self.scene.create_urdf_loader()
        loader.name = self.uid if self._agent_idx is None else f"{self.uid}-agent-{self._agent_idx}"
        loader.fix_root_link = self.fix_root_link
        loader.load_multiple_collisions_from_file = self.load_multiple_collisions
        loader.disable_self_collisions = self.disable_self_collisions
        if self.urdf_config is not None:
            cfg = sapien_utils.parse_urdf_config(self.urdf_config)
            sapien_utils.check_urdf_config(cfg)
            sapien_utils.apply_urdf_config(loader, cfg)
        path = str(self.urdf_path)
        if not os.path.exists(path):
            raise FileNotFoundError(f"Robot {self.uid} definition file not found at {path} (assets cannot be downloaded: no network)")
        builder = loader.parse(path)["articulation_builders"][0]
        builder.initial_pose = initial_pose
        self.robot = builder.build()
        assert self.robot is not None, f"Fail to load URDF from {path}"
        self.robot_link_names = [l.name for l in self.robot.get_links()]

    def _after_loading_articulation(self):
        pass

    def _after_init(self):
        pass

    @property
    def control_mode(self):
        return self._control_mode

    def set_control_mode(self, control_mode: str = None):
        if control_mode is None:
            control_mode = self._default_control_mode
        assert control_mode in self.supported_control_modes, f"{control_mode} not in supported modes: {self.supported_control_modes}"
        self._control_mode = control_mode
        if control_mode not in self.controllers:
            config = self._controller_configs[control_mode]
            balance_passive_force = True
            if isinstance(config, dict):
                if "balance_passive_force" in config:
                    balance_passive_force = config.pop("balance_passive_force")
                self.controllers[control_mode] = CombinedController(config, self.robot, self._control_freq, scene=self.scene)
            else:
                self.controllers[control_mode] = config.controller_cls(config, self.robot, self._control_freq, scene=self.scene)
            self.controllers[control_mode].set_drive_property()
            if balance_passive_force and not self.scene._gpu_sim_initialized:
                # "passive force balance" = gravity off on every robot link (base_agent.py:272-282)
                for link in self.robot.links:
                    link.disable_gravity = True

    @property
    def controller(self) -> BaseController:
        if self._control_mode is None:
            raise RuntimeError("Please specify a control mode first")
        return self.controllers[self._control_mode]

    @property
    def action_space(self):
        if self._control_mode is None:
            return spaces.Dict({uid: c.action_space for uid, c in self.controllers.items()})
        return self.controller.action_space

    @property
    def single_action_space(self):
        if self._control_mode is None:
            return spaces.Dict({uid: c.single_action_space for uid, c in self.controllers.items()})
        return self.controller.single_action_space

    def set_action(self, action):
        self.controller.set_action(action)

    def before_simulation_step(self):
        self.controller.before_simulation_step()

    def get_proprioception(self):
        obs = dict(qpos=self.robot.get_qpos(), qvel=self.robot.get_qvel())
        cs = self.controller.get_state()
        if len(cs) > 0:
            obs.update(controller=cs)
        return obs

    def get_state(self) -> Dict:
        root = self.robot.get_links()[0]
        return dict(
            robot_root_pose=root.get_pose(),
            robot_root_vel=root.get_linear_velocity(),
            robot_root_qvel=root.get_angular_velocity(),
            robot_qpos=self.robot.get_qpos(),
            robot_qvel=self.robot.get_qvel(),
            controller=self.controller.get_state(),
        )

    def set_state(self, state: Dict, ignore_controller=False):
        self.robot.set_root_pose(state["robot_root_pose"])
        self.robot.set_root_linear_velocity(state["robot_root_vel"])
        self.robot.set_root_angular_velocity(state["robot_root_qvel"])
        self.robot.set_qpos(state["robot_qpos"])
        self.robot.set_qvel(state["robot_qvel"])
        if not ignore_controller and "controller" in state:
            self.controller.set_state(state["controller"])
        self.scene._gpu_apply_all()
        self.scene.px.gpu_update_articulation_kinematics()
        self.scene._gpu_fetch_all()

    def reset(self, init_qpos=None):
        """zero velocity / generalized force, optionally set qpos (base_agent.py:380-390)"""
        if init_qpos is not None:
            self.robot.set_qpos(init_qpos)
        self.robot.set_qvel(torch.zeros(self.robot.max_dof, device=self.device))
        self.robot.set_qf(torch.zeros(self.robot.max_dof, device=self.device))

    def is_grasping(self, object=None):
        raise NotImplementedError()

    def is_static(self, threshold: float):
        raise NotImplementedError()
