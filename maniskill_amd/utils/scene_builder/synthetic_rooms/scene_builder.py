"""SyntheticRooms: procedurally generated rooms for `SceneManipulation-v1`, standing in for the ReplicaCAD apartments of
the reference (utils/scene_builder/replicacad/scene_builder.py:65-300), whose assets cannot be had here.

What it keeps of the original: several static layouts (`build_configs`), one per sub-scene, chosen per env when the scene
is built; scenery as static TRIANGLE MESHES (one mesh for a layout's walls, one for its furniture -- the collision meshes
of ReplicaCAD are triangle meshes too), built with `set_scene_idxs` per layout and merged into two actors, "walls" and
"furniture", whose mesh differs from sub-scene to sub-scene (`Actor.merge`: one shape slot each, however many layouts
there are); start arrangements of the robot (`init_configs`: base position and heading, inside the room); navigable
positions per env; and MOVABLE OBJECTS: every layout has two dynamic objects of its own, each a convex decomposition of
several hulls (`add_multiple_convex_collisions_from_file` on an OBJ with one `o` group per part, as ReplicaCAD's
dynamic objects are loaded, replicacad/scene_builder.py:156-185), built per layout with `set_scene_idxs` and -- as there
-- left as actors of their own: `movable_objects["env-<i>_<name>"]`. Objects of different layouts never share an env, so
the scene gives them common body rows (envs/scene.py `_setup`): ten distinct objects, two per sub-scene, two rows.
`initialize` puts them back on their furniture. What it does not have: articulated furniture.

A layout is a list of boxes (lo, hi) for its walls and another for its furniture; each list becomes one OBJ file in a
scratch directory, loaded through `ActorBuilder.add_nonconvex_collision_from_file` like any asset.
"""
import os
import tempfile
from typing import List

import numpy as np
import sapien
import torch

from maniskill_amd.agents.robots.fetch import FETCH_WHEELS_COLLISION_BIT
from maniskill_amd.utils.building.ground import build_ground
from maniskill_amd.utils.scene_builder.registration import register_scene_builder
from maniskill_amd.utils.scene_builder.scene_builder import SceneBuilder

_FACES = np.array([[0, 2, 3], [0, 3, 1], [4, 5, 7], [4, 7, 6], [0, 1, 5], [0, 5, 4], [2, 6, 7], [2, 7, 3], [0, 4, 6], [0, 6, 2], [1, 3, 7], [1, 7, 5]])


def _walls(x0, x1, y0, y1, height=1.2, thickness=0.1, door=None):
    """four walls around [x0, x1] x [y0, y1]; `door` = (y_from, y_to) leaves a gap in the wall at x1"""
    t = thickness
    boxes = [((x0 - t, y0 - t, 0), (x0, y1 + t, height)), ((x0, y0 - t, 0), (x1, y0, height)), ((x0, y1, 0), (x1, y1 + t, height))]
    if door is None:
        boxes.append(((x1, y0 - t, 0), (x1 + t, y1 + t, height)))
    else:
        boxes += [((x1, y0 - t, 0), (x1 + t, door[0], height)), ((x1, door[1], 0), (x1 + t, y1 + t, height))]
    return boxes


# name -> (walls, furniture, navigable box (x0, x1, y0, y1) for the base, start arrangements (x, y, yaw))
LAYOUTS = {
    "study": (
        _walls(-1.5, 1.5, -1.5, 1.5),
        [((0.9, -0.6, 0.0), (1.5, 0.6, 0.75)), ((-1.5, 0.9, 0.0), (-0.9, 1.5, 0.45))],  # a desk against the far wall, a low cabinet in a corner
        (-0.9, 0.3, -0.9, 0.5),
        [(-0.6, 0.0, 0.0), (0.0, -0.6, np.pi / 2)],
    ),
    "corridor": (
        _walls(-1.0, 3.0, -0.7, 0.7, door=(-0.45, 0.45)),
        [((1.6, 0.35, 0.0), (2.2, 0.7, 0.9))],  # a shelf along one side
        (-0.5, 2.4, -0.2, 0.1),
        [(-0.5, 0.0, 0.0), (2.0, -0.1, np.pi)],
    ),
    "kitchen": (
        _walls(-1.2, 1.8, -1.8, 1.2),
        [((-1.2, -1.8, 0.0), (1.8, -1.2, 0.9)), ((0.2, -0.3, 0.0), (1.0, 0.5, 0.9))],  # a counter along one wall, an island
        (-0.7, 1.3, -0.6, 0.7),
        [(-0.6, 0.4, -np.pi / 2), (-0.6, -0.5, 0.0)],
    ),
    "lab": (
        _walls(-2.0, 2.0, -1.2, 1.2, door=(-1.0, -0.1)),
        [((-2.0, 0.6, 0.0), (0.5, 1.2, 0.9)), ((1.2, -1.2, 0.0), (2.0, -0.4, 1.1)), ((-0.4, -0.5, 0.0), (0.4, -0.1, 0.7))],  # a bench, a cabinet, a cart
        (-1.5, 0.8, -0.8, 0.1),
        [(-1.4, -0.5, 0.0), (0.9, 0.2, np.pi)],
    ),
    "hall": (
        _walls(-2.5, 2.5, -2.5, 2.5),
        [((-0.3, -0.3, 0.0), (0.3, 0.3, 1.2)), ((1.3, 1.3, 0.0), (1.9, 1.9, 1.2)), ((-1.9, -1.9, 0.0), (-1.3, -1.3, 1.2))],  # three pillars
        (-2.0, 2.0, -2.0, 2.0),
        [(-1.2, 0.0, 0.0), (1.0, -1.0, np.pi / 2)],
    ),
}


# movable objects: name -> convex parts, each a box (lo, hi) or a prism ("prism", n, radius, z0, z1, centre xy) in the object frame
OBJECTS = {
    "bracket": [((-0.06, -0.02, 0.0), (0.06, 0.02, 0.03)), ((0.03, -0.02, 0.03), (0.06, 0.02, 0.10))],                      # an L
    "mug": [("prism", 10, 0.04, 0.0, 0.09, (0.0, 0.0)), ((0.035, -0.012, 0.02), (0.075, 0.012, 0.07))],                      # a cup with a handle
    "tee": [((-0.07, -0.02, 0.0), (0.07, 0.02, 0.04)), ((-0.02, 0.02, 0.0), (0.02, 0.09, 0.04))],                            # a T lying flat
    "dumbbell": [("prism", 8, 0.04, 0.0, 0.04, (-0.07, 0.0)), ("prism", 8, 0.04, 0.0, 0.04, (0.07, 0.0)), ((-0.07, -0.012, 0.008), (0.07, 0.012, 0.032))],
    "tray": [((-0.10, -0.07, 0.0), (0.10, 0.07, 0.012)), ((-0.10, -0.07, 0.012), (-0.088, 0.07, 0.04)), ((0.088, -0.07, 0.012), (0.10, 0.07, 0.04))],
    "wedge": [("prism", 3, 0.07, 0.0, 0.05, (0.0, 0.0)), ((-0.02, -0.09, 0.0), (0.02, -0.03, 0.03))],
    "post": [("prism", 6, 0.03, 0.0, 0.16, (0.0, 0.0)), ((-0.06, -0.06, 0.0), (0.06, 0.06, 0.015))],                        # a hexagonal post on a foot plate
    "step": [((-0.08, -0.05, 0.0), (0.08, 0.05, 0.04)), ((-0.08, 0.0, 0.04), (0.08, 0.05, 0.08))],                           # two stairs
    "block": [((-0.04, -0.04, 0.0), (0.04, 0.04, 0.08)), ((-0.015, -0.015, 0.08), (0.015, 0.015, 0.11))],                    # a cube with a knob
    "bar": [((-0.12, -0.015, 0.0), (0.12, 0.015, 0.03)), ((-0.12, -0.04, 0.0), (-0.09, 0.04, 0.03)), ((0.09, -0.04, 0.0), (0.12, 0.04, 0.03))],  # an I
}
# layout -> its objects: (object, (x, y, z) of the object frame -- on a piece of furniture or on the floor in the robot's way --, yaw).
# `SyntheticRooms` builds the first two of a layout, `SyntheticRoomsCrowded` all four.
LAYOUT_OBJECTS = {
    "study": [("mug", (1.05, -0.25, 0.75), 0.3), ("bracket", (0.15, 0.05, 0.0), 1.2), ("block", (1.2, 0.3, 0.75), 0.8), ("step", (-0.3, -0.45, 0.0), 0.4)],
    "corridor": [("tee", (1.85, 0.5, 0.9), 0.0), ("dumbbell", (0.6, 0.0, 0.0), 1.5708), ("mug", (2.05, 0.52, 0.9), 1.0), ("wedge", (1.2, -0.15, 0.0), 0.3)],
    "kitchen": [("tray", (0.6, 0.1, 0.9), 0.4), ("wedge", (-0.1, 0.35, 0.0), 2.0), ("post", (0.4, -1.5, 0.9), 0.0), ("bar", (-0.2, -0.45, 0.0), 0.9)],
    "lab": [("post", (-0.6, 0.8, 0.9), 0.0), ("step", (-0.75, -0.5, 0.0), 0.2), ("tray", (0.0, -0.3, 0.7), 0.0), ("tee", (-1.0, 0.1, 0.0), 1.1)],
    "hall": [("block", (-0.55, 0.0, 0.0), 0.5), ("bar", (0.9, -0.35, 0.0), 1.0), ("dumbbell", (-0.9, 0.5, 0.0), 0.2), ("bracket", (0.3, -1.0, 0.0), 2.2)],
}


def _write_parts_obj(path: str, parts) -> None:
    """one `o` group per convex part: what a convex-decomposition file looks like to `add_multiple_convex_collisions_from_file`"""
    with open(path, "w") as fh:
        base = 1
        for k, part in enumerate(parts):
            fh.write(f"o part_{k}\n")
            if part[0] == "prism":
                _, n, rad, z0, z1, (cx, cy) = part
                ring = [(cx + rad * np.cos(2 * np.pi * i / n), cy + rad * np.sin(2 * np.pi * i / n)) for i in range(n)]
                verts = [(x, y, z0) for x, y in ring] + [(x, y, z1) for x, y in ring]
                faces = [(i, (i + 1) % n, n + (i + 1) % n) for i in range(n)] + [(i, n + (i + 1) % n, n + i) for i in range(n)]
                faces += [(0, i + 1, i) for i in range(1, n - 1)] + [(n, n + i, n + i + 1) for i in range(1, n - 1)]
            else:
                lo, hi = np.asarray(part[0], float), np.asarray(part[1], float)
                verts = [tuple((hi if (i >> k2) & 1 else lo)[k2] for k2 in range(3)) for i in range(8)]
                faces = [tuple(t) for t in _FACES]
            for v in verts:
                fh.write("v %.6f %.6f %.6f\n" % v)
            for t in faces:
                fh.write("f %d %d %d\n" % tuple(base + i for i in t))
            base += len(verts)


def _write_boxes_obj(path: str, boxes) -> None:
    with open(path, "w") as fh:
        for b, (lo, hi) in enumerate(boxes):
            for i in range(8):
                fh.write("v %.6f %.6f %.6f\n" % tuple((hi if (i >> k) & 1 else lo)[k] for k in range(3)))
        for b in range(len(boxes)):
            for tri in _FACES + 8 * b + 1:
                fh.write("f %d %d %d\n" % tuple(tri))


@register_scene_builder("SyntheticRooms")
class SyntheticRoomsSceneBuilder(SceneBuilder):
    build_configs = list(LAYOUTS)
    init_configs = [0, 1]  # which of a layout's start arrangements
    robot_initial_pose = sapien.Pose()

    def __init__(self, env, robot_init_qpos_noise=0.02, movable_objects=2):
        super().__init__(env, robot_init_qpos_noise)
        self._mesh_dir = tempfile.mkdtemp(prefix="synthetic_rooms_")
        # movable objects per layout: 0 = scenery only (15 velocity components per env with the Fetch), 2 (27: two 16-lane rows per
        # env in the control-step kernel), up to 4 (39: four rows)
        self.movable = int(movable_objects)

    def __del__(self):  # (the scratch directory of generated meshes goes with the builder)
        import shutil

        shutil.rmtree(getattr(self, "_mesh_dir", ""), ignore_errors=True)

    def build(self, build_config_idxs: List[int] = None):
        n = self.env.num_envs
        if build_config_idxs is None:
            build_config_idxs = [i % len(self.build_configs) for i in range(n)]
        if len(build_config_idxs) == 1:
            build_config_idxs = list(build_config_idxs) * n
        assert len(build_config_idxs) == n, f"one layout per sub-scene: {n} envs, {len(build_config_idxs)} indices"
        self.build_config_idxs = [int(i) for i in build_config_idxs]
        self.scene_objects, self.movable_objects, self.articulations = {}, {}, {}
        self.ground = build_ground(self.scene)
        self.ground.set_collision_group_bit(group=2, bit_idx=FETCH_WHEELS_COLLISION_BIT, bit=1)
        self.scene_objects["ground"] = self.ground
        from maniskill_amd.utils.structs.actor import Actor

        fragments = {"walls": [], "furniture": []}
        for li, name in enumerate(self.build_configs):
            envs = [e for e, i in enumerate(self.build_config_idxs) if i == li]
            if not envs:
                continue
            for part, boxes in (("walls", LAYOUTS[name][0]), ("furniture", LAYOUTS[name][1])):
                path = os.path.join(self._mesh_dir, f"{name}_{part}.obj")
                _write_boxes_obj(path, [(np.asarray(lo, float), np.asarray(hi, float)) for lo, hi in boxes])
                b = self.scene.create_actor_builder()
                b.add_nonconvex_collision_from_file(path)
                b.set_scene_idxs(envs)
                b.initial_pose = sapien.Pose()
                fragments[part].append(b.build_static(name=f"{name}_{part}"))
        for part, frags in fragments.items():
            # (one layout in every env: its actor is no fragment, there is nothing to merge)
            self.scene_objects[part] = frags[0] if len(frags) == 1 and frags[0]._fragment is None else Actor.merge(frags, name=part)
        # the layouts' own movable objects (replicacad/scene_builder.py:156-185: one dynamic actor per object and layout,
        # `movable_objects` / `scene_objects` keyed by env), with the pose `initialize` puts them back to
        self._default_object_poses = []
        for li, name in enumerate(self.build_configs):
            envs = [e for e, i in enumerate(self.build_config_idxs) if i == li]
            if not envs or not self.movable:
                continue
            for oname, xyz, yaw in LAYOUT_OBJECTS[name][: self.movable]:
                path = os.path.join(self._mesh_dir, f"{oname}.obj")
                if not os.path.exists(path):
                    _write_parts_obj(path, OBJECTS[oname])
                b = self.scene.create_actor_builder()
                b.add_multiple_convex_collisions_from_file(path)
                b.set_scene_idxs(envs)
                pose = sapien.Pose(p=list(xyz), q=[float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2))])
                b.initial_pose = pose
                actor = b.build(name=f"{name}_{oname}")
                self._default_object_poses.append((actor, pose))
                for e in envs:
                    self.movable_objects[f"env-{e}_{oname}"] = actor
                    self.scene_objects[f"env-{e}_{oname}"] = actor
        self.navigable_positions = [LAYOUTS[self.build_configs[i]][2] for i in self.build_config_idxs]

    def initialize(self, env_idx: torch.Tensor, init_config_idxs: List[int] = None):
        agent = self.env.agent
        dev = self.env.device
        idx = env_idx.tolist() if isinstance(env_idx, torch.Tensor) else list(env_idx)
        if init_config_idxs is None:
            init_config_idxs = [0] * self.env.num_envs
        if len(init_config_idxs) == 1:
            init_config_idxs = list(init_config_idxs) * self.env.num_envs
        rest = torch.as_tensor(agent.keyframes["rest"].qpos, dtype=torch.float32, device=dev).repeat(len(idx), 1)
        if agent.uid == "fetch":  # the base joints are the robot's place in the room
            for row, e in enumerate(idx):
                starts = LAYOUTS[self.build_configs[self.build_config_idxs[e]]][3]
                x, y, yaw = starts[int(init_config_idxs[e]) % len(starts)]
                rest[row, 0], rest[row, 1], rest[row, 2] = x, y, yaw
        agent.reset(rest)
        # movable objects back to where the layout has them, at rest (replicacad/scene_builder.py:303-310)
        for actor, pose in self._default_object_poses:  # (only the rows of envs being reset are written)
            actor.set_pose(pose)
            actor.set_linear_velocity(torch.zeros(3, device=dev))
            actor.set_angular_velocity(torch.zeros(3, device=dev))


@register_scene_builder("SyntheticRoomsStatic")
class SyntheticRoomsStaticSceneBuilder(SyntheticRoomsSceneBuilder):
    """the same rooms without the movable objects: scenery only (15 velocity components per env with the Fetch)"""

    def __init__(self, env, robot_init_qpos_noise=0.02):
        super().__init__(env, robot_init_qpos_noise, movable_objects=0)


@register_scene_builder("SyntheticRoomsCrowded")
class SyntheticRoomsCrowdedSceneBuilder(SyntheticRoomsSceneBuilder):
    """four movable objects per layout (39 velocity components per env with the Fetch: a whole wave per env)"""

    def __init__(self, env, robot_init_qpos_noise=0.02):
        super().__init__(env, robot_init_qpos_noise, movable_objects=4)
