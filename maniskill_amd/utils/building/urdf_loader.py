"""URDFLoader counterpart (mani_skill/utils/building/urdf_loader.py:23-120): parse a URDF (+SRDF)
into articulation builders. The XML parsing that the reference leaves to SAPIEN's C++ loader is
done by maniskill_amd/model/urdf.py."""
import os
from typing import Dict, Optional

from maniskill_amd.model.compile import ArticulationRecord, shapes_from_urdf_link
from maniskill_amd.model.urdf import parse_urdf
from maniskill_amd.utils.building.articulation_builder import ArticulationBuilder


class URDFLoader:
    def __init__(self, scene=None):
        self.scene = scene
        self.name: Optional[str] = None
        self.fix_root_link = True
        self.load_multiple_collisions_from_file = False
        self.disable_self_collisions = False
        self.multiple_collisions_decomposition = "none"
        self.scale = 1.0
        # filled by sapien_utils.apply_urdf_config
        self._link_material: Dict[str, dict] = {}
        self._link_patch_radius: Dict[str, float] = {}
        self._link_min_patch_radius: Dict[str, float] = {}
        self._link_density: Dict[str, float] = {}
        self._default_material = None

    # names used by apply_urdf_config (mani_skill/utils/sapien_utils.py:146-168)
    def set_material(self, static_friction, dynamic_friction, restitution):
        self._default_material = dict(static_friction=static_friction, dynamic_friction=dynamic_friction, restitution=restitution)

    def set_link_material(self, link_name, static_friction, dynamic_friction, restitution):
        self._link_material[link_name] = dict(static_friction=static_friction, dynamic_friction=dynamic_friction, restitution=restitution)

    def set_link_patch_radius(self, link_name, r):
        self._link_patch_radius[link_name] = r

    def set_link_min_patch_radius(self, link_name, r):
        self._link_min_patch_radius[link_name] = r

    def set_link_density(self, link_name, d):
        self._link_density[link_name] = d

    def parse(self, urdf_file, srdf_file=None, package_dir=None):
        rb = parse_urdf(str(urdf_file), srdf_file)
        default = self.scene.default_material
        link_shapes = {}
        for lname, link in rb.links.items():
            cfg = {}
            mat = self._link_material.get(lname, self._default_material)
            cfg["material"] = mat if mat is not None else dict(
                static_friction=default.static_friction, dynamic_friction=default.dynamic_friction, restitution=default.restitution
            )
            if lname in self._link_patch_radius:
                cfg["patch_radius"] = self._link_patch_radius[lname]
            if lname in self._link_min_patch_radius:
                cfg["min_patch_radius"] = self._link_min_patch_radius[lname]
            if lname in self._link_density:
                cfg["density"] = self._link_density[lname]
            link_shapes[lname] = shapes_from_urdf_link(link, link_cfg=cfg)
        rec = ArticulationRecord(
            name=self.name or rb.name,
            robot=rb,
            fix_root_link=self.fix_root_link,
            link_shapes=link_shapes,
            disable_self_collisions=self.disable_self_collisions,
        )
        builder = ArticulationBuilder(self.scene, rec)
        builder.set_name(self.name or rb.name)
        builder.disable_self_collisions = self.disable_self_collisions
        return dict(articulation_builders=[builder], actor_builders=[], cameras=[])

    def load_file_as_articulation_builder(self, urdf_file, srdf_file=None, package_dir=None):
        return self.parse(urdf_file, srdf_file, package_dir)["articulation_builders"][0]

    def load(self, urdf_file, srdf_file=None, package_dir=None, name=None, scene_idxs=None):
        if name is not None:
            self.name = name
        b = self.load_file_as_articulation_builder(urdf_file, srdf_file, package_dir)
        return b.build()
