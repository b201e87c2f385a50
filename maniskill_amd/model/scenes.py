"""Ready-made scene models for tests / benches below the env layer.

`panda_tabletop_model` restates the PickCube-v1 scene content
(mani_skill/envs/tasks/tabletop/pick_cube.py:60-81, utils/scene_builder/table/scene_builder.py:20-58,
utils/building/ground.py:36-45, agents/robots/panda/panda.py:16-75) directly as model records.
"""
import os

import numpy as np

from .. import PACKAGE_ASSET_DIR
from . import geom
from .compile import ActorRecord, ArticulationRecord, SceneModelBuilder, ShapeRecord, shapes_from_urdf_link
from .urdf import parse_urdf

PANDA_URDF = os.path.join(PACKAGE_ASSET_DIR, "robots/panda/panda_v2.urdf")
PANDA_LINK_CFG = dict(
    panda_leftfinger=dict(
        material=dict(static_friction=2.0, dynamic_friction=2.0, restitution=0.0), patch_radius=0.1, min_patch_radius=0.1
    ),
    panda_rightfinger=dict(
        material=dict(static_friction=2.0, dynamic_friction=2.0, restitution=0.0), patch_radius=0.1, min_patch_radius=0.1
    ),
)
TABLE_HEIGHT = 0.9196429


def panda_record(urdf=PANDA_URDF, stiffness=1e3, damping=1e2, force_limit=100.0, root_pose=None, max_hull_verts=64):
    rb = parse_urdf(urdf)
    link_shapes = {n: shapes_from_urdf_link(l, link_cfg=PANDA_LINK_CFG.get(n), max_hull_verts=max_hull_verts) for n, l in rb.links.items()}
    drives = {j.name: (stiffness, damping, force_limit, 0) for j in rb.joints if j.type != "fixed"}
    return ArticulationRecord(
        name="panda",
        robot=rb,
        initial_pose=geom.pose([-0.615, 0, 0]) if root_pose is None else root_pose,
        fix_root_link=True,
        link_shapes=link_shapes,
        link_gravity={n: False for n in rb.links},  # base_agent.py:272-282
        drives=drives,
    )


def table_record():
    yaw90 = geom.rpy_to_quat([0, 0, np.pi / 2])
    return ActorRecord(
        "table-workspace",
        "kinematic",
        [ShapeRecord("box", geom.pose([0, 0, TABLE_HEIGHT / 2]), half_size=np.array([2.418 / 2, 1.209 / 2, TABLE_HEIGHT / 2]))],
        initial_pose=geom.pose([-0.12, 0, -TABLE_HEIGHT], yaw90),
    )


def ground_record(altitude=-TABLE_HEIGHT):
    # plane normal = +x of the shape frame; rotate x -> z (ground.py:36-45 uses q=[0.7071068, 0, -0.7071068, 0])
    return ActorRecord(
        "ground",
        "static",
        [ShapeRecord("plane", geom.pose(q=[0.7071068, 0, -0.7071068, 0]))],
        initial_pose=geom.pose([0, 0, altitude]),
    )


def cube_record(half_size=0.02, name="cube", p=(0, 0, 0.02)):
    return ActorRecord(name, "dynamic", [ShapeRecord("box", geom.pose(), half_size=np.array([half_size] * 3))], initial_pose=geom.pose(p))


def panda_tabletop_model(with_cube=True, with_goal=True, with_robot=True, max_hull_verts=64, **scene_kwargs):
    b = SceneModelBuilder()
    if with_robot:
        b.set_articulation(panda_record(max_hull_verts=max_hull_verts))
    b.add_actor(table_record())
    b.add_actor(ground_record())
    if with_cube:
        b.add_actor(cube_record())
    if with_goal:
        b.add_actor(ActorRecord("goal_site", "kinematic", [], initial_pose=geom.pose()))
    return b.compile(**scene_kwargs)
