"""Primitive actors by one call (same call signatures as mani_skill/utils/building/actors/common.py).

Every helper is the same three steps -- a builder, one collision primitive (the visual twin is accepted and dropped:
this build has no renderer), build as dynamic / static / kinematic -- so they share `_primitive_actor`, which is
told the primitive by the name of the builder method and its arguments.
"""
from typing import Optional  # noqa: F401

_BUILD_METHOD = {"dynamic": "build", "static": "build_static", "kinematic": "build_kinematic"}


def _primitive_actor(scene, name: str, body_type: str, add_collision: bool, scene_idxs, initial_pose, primitive: str, **geometry):
    if body_type not in _BUILD_METHOD:
        raise ValueError(f"Unknown body type {body_type}")
    builder = scene.create_actor_builder()
    if add_collision:
        getattr(builder, f"add_{primitive}_collision")(**geometry)
    getattr(builder, f"add_{primitive}_visual")(**geometry)
    if scene_idxs is not None:
        builder.set_scene_idxs(scene_idxs)
    if initial_pose is not None:
        builder.set_initial_pose(initial_pose)
    return getattr(builder, _BUILD_METHOD[body_type])(name=name)


def build_cube(scene, half_size: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "box", half_size=[half_size] * 3)


def build_box(scene, half_sizes, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "box", half_size=half_sizes)


def build_cylinder(scene, radius: float, half_length: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "cylinder", radius=radius, half_length=half_length)


def build_sphere(scene, radius: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "sphere", radius=radius)


def build_red_white_target(scene, radius: float, thickness: float, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    """flat disc target (PushCube's goal region); as a collision shape it is one cylinder"""
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "cylinder", radius=radius, half_length=thickness / 2)


def build_twocolor_peg(scene, length, width, color_1, color_2, name: str, body_type="dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    return _primitive_actor(scene, name, body_type, add_collision, scene_idxs, initial_pose, "box", half_size=[length, width, width])
