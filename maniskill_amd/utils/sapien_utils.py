"""Small helpers tasks and agents use (counterpart of the used parts of
mani_skill/utils/sapien_utils.py: get_obj_by_name :60-75, look_at :297-330,
parse_urdf_config / check_urdf_config / apply_urdf_config :113-168)."""
from typing import Dict, List, Sequence, TypeVar

import numpy as np
import sapien

T = TypeVar("T")


def get_obj_by_name(objs: List[T], name: str, is_unique=True):
    matched = [x for x in objs if x.get_name() == name] if objs and hasattr(objs[0], "get_name") else [x for x in objs if x.name == name]
    if len(matched) > 1:
        if not is_unique:
            return matched
        raise RuntimeError(f"Multiple objects with the same name {name}.")
    return matched[0] if matched else None


def get_objs_by_names(objs: List[T], names: List[str]) -> List[T]:
    m = {o.name: o for o in objs}
    return [m[n] for n in names]


def get_obj_by_type(objs: List[T], target_type, is_unique=True):
    matched = [x for x in objs if type(x) == target_type]
    if len(matched) > 1:
        if not is_unique:
            return matched
        raise RuntimeError(f"Multiple objects with the same type {target_type}.")
    return matched[0] if matched else None


def look_at(eye, target, up=(0, 0, 1)):
    """camera pose (x forward, z up) looking from `eye` at `target`; returned as a batched Pose"""
    from maniskill_amd.utils.structs.pose import Pose

    eye, target, up = np.asarray(eye, dtype=np.float64), np.asarray(target, dtype=np.float64), np.asarray(up, dtype=np.float64)
    fwd = target - eye
    fwd = fwd / np.linalg.norm(fwd)
    left = np.cross(up, fwd)
    left = left / np.linalg.norm(left)
    upv = np.cross(fwd, left)
    T = np.eye(4)
    T[:3, :3] = np.stack([fwd, left, upv], axis=1)
    T[:3, 3] = eye
    return Pose.create(sapien.Pose(T))


def parse_urdf_config(config_dict: dict) -> Dict:
    """expand material references: link entries may name a material of `_materials`"""
    urdf_config = dict()
    mtl_cfgs = dict(config_dict.get("_materials", {}))
    materials = {k: dict(v) for k, v in mtl_cfgs.items()}
    default = config_dict.get("material")
    if default is not None:
        urdf_config["material"] = materials[default] if isinstance(default, str) else default
    urdf_config["link"] = dict()
    for name, link_cfg in config_dict.get("link", {}).items():
        link_cfg = dict(link_cfg)
        if isinstance(link_cfg.get("material"), str):
            link_cfg["material"] = materials[link_cfg["material"]]
        urdf_config["link"][name] = link_cfg
    return urdf_config


def check_urdf_config(urdf_config: dict):
    allowed = {"material", "density", "link"}
    for k in urdf_config:
        if k not in allowed:
            raise KeyError(f"Not allowed key ({k}) for `sapien.URDFLoader.load_from_string`. Allowed keys are f{allowed}")
    allowed_link = {"material", "density", "patch_radius", "min_patch_radius"}
    for lk, cfg in urdf_config.get("link", {}).items():
        for k in cfg:
            if k not in allowed_link:
                raise KeyError(f"Not allowed key ({k}) for link {lk}. Allowed keys are f{allowed_link}")


def apply_urdf_config(loader, urdf_config: dict):
    if "link" in urdf_config:
        for name, link_cfg in urdf_config["link"].items():
            if "material" in link_cfg:
                m = link_cfg["material"]
                loader.set_link_material(name, m["static_friction"], m["dynamic_friction"], m["restitution"])
            if "patch_radius" in link_cfg:
                loader.set_link_patch_radius(name, link_cfg["patch_radius"])
            if "min_patch_radius" in link_cfg:
                loader.set_link_min_patch_radius(name, link_cfg["min_patch_radius"])
            if "density" in link_cfg:
                loader.set_link_density(name, link_cfg["density"])
    if "material" in urdf_config:
        m = urdf_config["material"]
        loader.set_material(m["static_friction"], m["dynamic_friction"], m["restitution"])


def hex2rgba(h, correction=True):
    """'#RRGGBB' -> linear rgba (gamma 2.2 undone when `correction`)"""
    h = h.lstrip("#")
    rgb = [int(h[i : i + 2], 16) / 255 for i in (0, 2, 4)]
    rgba = np.array(rgb + [1.0])
    return rgba**2.2 if correction else rgba
