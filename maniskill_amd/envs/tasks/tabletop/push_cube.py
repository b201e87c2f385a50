"""PushCube-v1 (task definition restated from mani_skill/envs/tasks/tabletop/push_cube.py:36-220):
push a cube into a goal disc 0.1 + r in front of it; panda_wristcam by default."""
from typing import Any, Dict

import numpy as np
import sapien
import torch
from transforms3d.euler import euler2quat

from maniskill_amd.envs.sapien_env import BaseEnv
from maniskill_amd.sensors.camera import CameraConfig
from maniskill_amd.utils import sapien_utils
from maniskill_amd.utils.building import actors
from maniskill_amd.utils.registration import register_env
from maniskill_amd.utils.scene_builder.table import TableSceneBuilder
from maniskill_amd.utils.structs.pose import Pose
from maniskill_amd.utils.structs.types import GPUMemoryConfig, SimConfig


@register_env("PushCube-v1", max_episode_steps=50)
class PushCubeEnv(BaseEnv):
    SUPPORTED_ROBOTS = ["panda_wristcam", "fetch"]
    goal_radius = 0.1
    cube_half_size = 0.02

    def __init__(self, *args, robot_uids="panda_wristcam", robot_init_qpos_noise=0.02, **kwargs):
        self.robot_init_qpos_noise = robot_init_qpos_noise
        super().__init__(*args, robot_uids=robot_uids, **kwargs)

    @property
    def _default_sim_config(self):
        return SimConfig(gpu_memory_config=GPUMemoryConfig(found_lost_pairs_capacity=2**25, max_rigid_patch_count=2**18))

    @property
    def _default_sensor_configs(self):
        pose = sapien_utils.look_at(eye=[0.3, 0, 0.6], target=[-0.1, 0, 0.1])
        return [CameraConfig("base_camera", pose=pose, width=128, height=128, fov=np.pi / 2, near=0.01, far=100)]

    def _load_agent(self, options: dict):
        super()._load_agent(options, sapien.Pose(p=[-0.615, 0, 0]))

    def _load_scene(self, options: dict):
        self._fused_state = None
        self.table_scene = TableSceneBuilder(env=self, robot_init_qpos_noise=self.robot_init_qpos_noise)
        self.table_scene.build()
        self.obj = actors.build_cube(
            self.scene, half_size=self.cube_half_size, color=np.array([12, 42, 160, 255]) / 255, name="cube", body_type="dynamic",
            initial_pose=sapien.Pose(p=[0, 0, self.cube_half_size]),
        )
        self.goal_region = actors.build_red_white_target(
            self.scene, radius=self.goal_radius, thickness=1e-5, name="goal_region", add_collision=False, body_type="kinematic",
            initial_pose=sapien.Pose(p=[0, 0, 1e-3]),
        )

    def _initialize_episode(self, env_idx: torch.Tensor, options: dict):
        dev = self.device  # explicit devices, see PickCubeEnv._initialize_episode
        b = len(env_idx)
        self.table_scene.initialize(env_idx)
        xyz = torch.zeros((b, 3), device=dev)
        xyz[..., :2] = torch.rand((b, 2), device=dev) * 0.2 - 0.1
        xyz[..., 2] = self.cube_half_size
        self.obj.set_pose(Pose.create_from_pq(p=xyz, q=[1, 0, 0, 0]))
        target = xyz + torch.tensor([0.1 + self.goal_radius, 0, 0], device=dev)
        target[..., 2] = 1e-3
        self.goal_region.set_pose(Pose.create_from_pq(p=target, q=euler2quat(0, np.pi / 2, 0)))

    def evaluate(self):
        is_obj_placed = (
            torch.linalg.norm(self.obj.pose.p[..., :2] - self.goal_region.pose.p[..., :2], axis=1) < self.goal_radius
        ) & (self.obj.pose.p[..., 2] < self.cube_half_size + 5e-3)
        return {"success": is_obj_placed}

    def _get_obs_extra(self, info: Dict):
        obs = dict(tcp_pose=self.agent.tcp.pose.raw_pose)
        if self.obs_mode_struct.use_state:
            obs.update(goal_pos=self.goal_region.pose.p, obj_pose=self.obj.pose.raw_pose)
        return obs

    def compute_dense_reward(self, obs: Any, action, info: Dict):
        push_p = self.obj.pose.p + torch.tensor([-self.cube_half_size - 0.005, 0, 0], device=self.device)
        dist = torch.linalg.norm(push_p - self.agent.tcp.pose.p, axis=1)
        reward = 1 - torch.tanh(5 * dist)
        reached = dist < 0.01
        obj_to_goal = torch.linalg.norm(self.obj.pose.p[..., :2] - self.goal_region.pose.p[..., :2], axis=1)
        reward += (1 - torch.tanh(5 * obj_to_goal)) * reached
        reward[info["success"]] = 3
        return reward

    def compute_normalized_dense_reward(self, obs: Any, action, info: Dict):
        return self.compute_dense_reward(obs=obs, action=action, info=info) / 3.0

    # ---- fused evaluate + obs + reward (one native launch; identical results, tests/test_gpu_env.py) ----
    def _fused_task_ok(self) -> bool:
        cls = type(self)
        same = all(
            getattr(cls, m) is getattr(PushCubeEnv, m)
            for m in ("evaluate", "_get_obs_extra", "compute_dense_reward", "compute_normalized_dense_reward", "_get_obs_agent", "get_obs", "get_info", "get_reward")
        )
        ok = same and self._obs_mode == "state" and self._reward_mode in ("dense", "normalized_dense") and len(self.agent.controller.get_state()) == 0
        return ok

    def _fused_step_outputs(self, action, advance: bool = True):
        if not self._fused_ok():
            return None
        from maniskill_amd import native

        px = self.scene.px
        st = getattr(self, "_fused_state", None)
        if st is None or st["px"] is not px:
            task = native.PushTask(tcp_row=self.agent.tcp._body_row, obj_row=self.obj._body_row, goal_row=self.goal_region._body_row,
                                   goal_radius=self.goal_radius, cube_half_size=self.cube_half_size,
                                   reward_scale=1.0 / 3.0 if self._reward_mode == "normalized_dense" else 1.0)
            st = self._fused_state = dict(px=px, task=task)
        N, D = self.num_envs, 2 * self.agent.robot.max_dof + 17
        obs = torch.empty((N, D), dtype=torch.float32, device=self.device)
        reward = torch.empty((N,), dtype=torch.float32, device=self.device)
        flags = torch.empty((N, 1), dtype=torch.uint8, device=self.device)
        es = self._fused_bind_counters(st["task"], advance)
        px.task_push_outputs(st["task"], obs, reward, flags)
        return obs, reward, dict(elapsed_steps=es, success=flags.view(torch.bool)[:, 0])
