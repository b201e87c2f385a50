"""Table-top scene (counterpart of mani_skill/utils/scene_builder/table/scene_builder.py:20-134):
kinematic table box whose top face is z = 0, ground plane at -table_height, robot initial qpos =
rest keyframe + N(0, robot_init_qpos_noise) drawn from the numpy episode RNG."""
import numpy as np
import sapien
import torch
from transforms3d.euler import euler2quat

from maniskill_amd.utils.building.ground import build_ground
from maniskill_amd.utils.scene_builder.scene_builder import SceneBuilder
from maniskill_amd.utils.structs.pose import Pose

TABLE_HEIGHT = 0.9196429


class TableSceneBuilder(SceneBuilder):
    def build(self, scale=1.75, table_path="table.glb"):
        builder = self.scene.create_actor_builder()
        builder.add_box_collision(pose=sapien.Pose(p=[0, 0, TABLE_HEIGHT / 2]), half_size=(2.418 / 2, 1.209 / 2, TABLE_HEIGHT / 2))
        builder.initial_pose = sapien.Pose(p=[-0.12, 0, -TABLE_HEIGHT], q=euler2quat(0, 0, np.pi / 2))
        table = builder.build_kinematic(name="table-workspace")
        # the reference reads these off the visual mesh's AABB (table.glb scaled 1.75, yawed 90 deg)
        self.table_length = 1.209
        self.table_width = 2.418
        self.table_height = TABLE_HEIGHT
        self.ground = build_ground(self.scene, floor_width=100, altitude=-self.table_height)
        self.table = table
        self.scene_objects = [self.table, self.ground]

    def _noisy_qpos(self, env_idx, qpos):
        b = len(env_idx)
        if self.env._enhanced_determinism:
            return self.env._batched_episode_rng[env_idx].normal(0, self.robot_init_qpos_noise, len(qpos)) + qpos
        return self.env._episode_rng.normal(0, self.robot_init_qpos_noise, (b, len(qpos))) + qpos

    def initialize(self, env_idx: torch.Tensor):
        # constant poses live on the device (built once): a host -> device copy per reset makes the host wait for the
        # work queued before it
        cache = self.__dict__.setdefault("_pose_cache", {})
        dev = self.env.device
        if cache.get("device") != dev:
            cache.update(device=dev, table=Pose.create(sapien.Pose(p=[-0.12, 0, -TABLE_HEIGHT], q=euler2quat(0, 0, np.pi / 2)), device=dev),
                         root=Pose.create(sapien.Pose([-0.615, 0, 0]), device=dev))
        self.table.set_pose(cache["table"])
        uid = self.env.robot_uids
        if uid == "panda":
            qpos = np.array([0.0, -np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, np.pi / 4, 0.04, 0.04])
        elif uid == "panda_wristcam":
            qpos = np.array([0.0, np.pi / 8, 0, -np.pi * 5 / 8, 0, np.pi * 3 / 4, -np.pi / 4, 0.04, 0.04])
        elif uid in (None, "none"):
            return
        else:
            raise NotImplementedError(f"table scene initialisation for robot {uid!r} is not part of this build")
        qpos = self._noisy_qpos(env_idx, qpos)
        qpos[:, -2:] = 0.04
        self.env.agent.reset(qpos)
        self.env.agent.robot.set_pose(cache["root"])
