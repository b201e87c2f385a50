"""A synthetic stand-in for BASELINE config 5's ingredients (ReplicaCAD itself is not available): the Fetch in sub-scenes whose
static scenery differs from env to env -- triangle-mesh walls built with `set_scene_idxs`, never merged, so each exists
in its own envs only (the way the reference's scene builders place one scene per `build_config_idx`,
envs/scenes/base_env.py:19-82)."""
import numpy as np
import sapien

from maniskill_amd.envs.tasks.empty_env import EmptyEnv

_BOX_FACES = [[0, 2, 3], [0, 3, 1], [4, 5, 7], [4, 7, 6], [0, 1, 5], [0, 5, 4], [2, 6, 7], [2, 7, 3], [0, 4, 6], [0, 6, 2], [1, 3, 7], [1, 7, 5]]


def write_box_obj(path, lo, hi):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    with open(path, "w") as fh:
        for i in range(8):
            fh.write("v %f %f %f\n" % tuple((hi if (i >> k) & 1 else lo)[k] for k in range(3)))
        for t in _BOX_FACES:
            fh.write("f %d %d %d\n" % tuple(i + 1 for i in t))


def make_rooms_env(mesh_dir, num_envs, sim_backend):
    """even envs: a wall across the way at x = 0.8 m; odd envs: one at x = 1.4 m"""
    near, far = f"{mesh_dir}/wall_near.obj", f"{mesh_dir}/wall_far.obj"
    write_box_obj(near, [0.8, -1.0, 0.0], [0.9, 1.0, 1.0])
    write_box_obj(far, [1.4, -1.0, 0.0], [1.5, 1.0, 1.0])

    class FetchRooms(EmptyEnv):
        def _load_scene(self, options):
            super()._load_scene(options)
            self.walls = []
            for k, path in enumerate((near, far)):
                b = self.scene.create_actor_builder()
                b.add_nonconvex_collision_from_file(path)
                b.set_scene_idxs([i for i in range(self.num_envs) if i % 2 == k])
                b.initial_pose = sapien.Pose()
                self.walls.append(b.build_static(name=f"wall_{k}"))

    return FetchRooms(num_envs=num_envs, robot_uids="fetch", obs_mode="state", sim_backend=sim_backend)
