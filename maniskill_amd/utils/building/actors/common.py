"""Primitive actor helpers (same call signatures as mani_skill/utils/building/actors/common.py)."""
from typing import Optional

import numpy as np
import sapien


def _build_by_type(builder, name, body_type, scene_idxs=None, initial_pose=None):
    if scene_idxs is not None:
        builder.set_scene_idxs(scene_idxs)
    if initial_pose is not None:
        builder.set_initial_pose(initial_pose)
    if body_type == "dynamic":
        return builder.build(name=name)
    if body_type == "static":
        return builder.build_static(name=name)
    if body_type == "kinematic":
        return builder.build_kinematic(name=name)
    raise ValueError(f"Unknown body type {body_type}")


def build_cube(scene, half_size: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_box_collision(half_size=[half_size] * 3)
    builder.add_box_visual(half_size=[half_size] * 3)
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)


def build_box(scene, half_sizes, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_box_collision(half_size=half_sizes)
    builder.add_box_visual(half_size=half_sizes)
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)


def build_cylinder(scene, radius: float, half_length: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_cylinder_collision(radius=radius, half_length=half_length)
    builder.add_cylinder_visual(radius=radius, half_length=half_length)
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)


def build_sphere(scene, radius: float, color, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_sphere_collision(radius=radius)
    builder.add_sphere_visual(radius=radius)
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)


def build_red_white_target(scene, radius: float, thickness: float, name: str, body_type: str = "dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    """flat disc target (PushCube goal region); collision = one cylinder"""
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_cylinder_collision(radius=radius, half_length=thickness / 2)
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)


def build_twocolor_peg(scene, length, width, color_1, color_2, name: str, body_type="dynamic", add_collision: bool = True, scene_idxs=None, initial_pose=None):
    builder = scene.create_actor_builder()
    if add_collision:
        builder.add_box_collision(half_size=[length, width, width])
    return _build_by_type(builder, name, body_type, scene_idxs, initial_pose)
