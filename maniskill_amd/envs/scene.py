"""ManiSkillScene: owns the `px` system, the actor / articulation registry and the partial-reset mask.

Counterpart of mani_skill/envs/scene.py (step :374-375, contact queries :736-796, state registry
:819-892, _setup :897-939, _gpu_apply_all :941-957, _gpu_fetch_all :959-977). Rendering, cameras,
lights and the viewer are out of scope (state observations only): the light / camera calls are
accepted and ignored. There are no sub-scenes: every env shares one compiled model and has its own
coordinate frame, so no spacing offsets exist.
"""
import contextlib
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from maniskill_amd.model.compile import SceneModelBuilder
from maniskill_amd.physx.system import MssimSystem
from maniskill_amd.utils import common
from maniskill_amd import physx
from maniskill_amd.utils.building.actor_builder import ActorBuilder, PhysxMaterial
from maniskill_amd.utils.building.urdf_loader import URDFLoader
from maniskill_amd.utils.structs.actor import Actor, Link
from maniskill_amd.utils.structs.articulation import Articulation
from maniskill_amd.utils.structs.pose import Pose
from maniskill_amd.utils.structs.types import SimConfig


class ManiSkillScene:
    def __init__(self, num_envs: int, sim_config: SimConfig = None, device="cuda", backend_name: str = "physx_cuda", parallel_in_single_scene: bool = False):
        self.num_envs = int(num_envs)
        self.sim_config = sim_config if sim_config is not None else SimConfig()
        self.device = torch.device(device)
        self.parallel_in_single_scene = parallel_in_single_scene
        self.px = MssimSystem(device=self.device, backend=backend_name)
        self.actors: Dict[str, Actor] = {}
        self.articulations: Dict[str, Articulation] = {}
        self.state_dict_registry = _Registry()
        self.sensors = {}
        self.human_render_cameras = {}
        self._builder = SceneModelBuilder()
        self._fragments: Dict[str, Actor] = {}  # per-env actor fragments waiting for Actor.merge
        self._gpu_sim_initialized = False
        self._needs_fetch = False
        self._all_env_idx = torch.arange(self.num_envs, device=self.device)
        self._reset_mask_t = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
        self._reset_mask_all = True
        self._reset_idx_t = self._all_env_idx
        m = self.sim_config.default_materials_config
        self.default_material = PhysxMaterial(m.static_friction, m.dynamic_friction, m.restitution)
        self._pair_queries = {}
        self._body_queries = {}
        self.sub_scenes = [None] * self.num_envs  # API placeholder: there are no per-env scene objects

    # ------------------------------------------------------------------ reset mask
    @property
    def _reset_mask(self) -> torch.Tensor:
        return self._reset_mask_t

    @_reset_mask.setter
    def _reset_mask(self, mask: torch.Tensor):
        """external assignment (reference style); costs one host sync to learn whether it is all-true"""
        self._reset_mask_t = mask.to(self.device).bool()
        self._reset_mask_all = bool(self._reset_mask_t.all())
        self._reset_idx_t = None

    @property
    def _reset_idx(self) -> torch.Tensor:
        """ascending indices of the envs selected by `_reset_mask`. Writes of `k` rows "where the mask is true" go
        through these (`buf[idx] = rows`): indexing with the boolean mask itself makes torch count the selected rows
        on the host, one device synchronisation per write -- a dozen per partial reset."""
        if self._reset_idx_t is None:
            self._reset_idx_t = torch.nonzero(self._reset_mask_t).flatten()
        return self._reset_idx_t

    def _set_reset_idx(self, env_idx: Optional[torch.Tensor]):
        """sync-free form used by BaseEnv: None = all envs"""
        if env_idx is None or len(env_idx) == self.num_envs:
            self._reset_mask_t = torch.ones(self.num_envs, dtype=torch.bool, device=self.device)
            self._reset_mask_all = True
        else:
            m = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
            m[env_idx] = True
            self._reset_mask_t = m
            self._reset_mask_all = False
            self._reset_idx_t = torch.sort(env_idx.to(self.device).long()).values  # mask order = ascending env order
            return
        self._reset_idx_t = self._all_env_idx

    @contextlib.contextmanager
    def _narrow_reset_mask(self, env_idx):
        """actor.py:142-160 / articulation.py:281-303: set_state(env_idx) temporarily narrows the mask"""
        if env_idx is None:
            yield
            return
        prev, prev_all, prev_idx = self._reset_mask_t, self._reset_mask_all, self._reset_idx_t
        self._set_reset_idx(common.to_tensor(env_idx, device=self.device).long())
        try:
            yield
        finally:
            self._reset_mask_t, self._reset_mask_all, self._reset_idx_t = prev, prev_all, prev_idx

    # ------------------------------------------------------------------ properties
    @property
    def gpu_sim_enabled(self) -> bool:
        return True

    @property
    def timestep(self) -> float:
        return self.px.timestep

    @timestep.setter
    def timestep(self, dt):
        self.px.timestep = dt

    # ------------------------------------------------------------------ builders
    def create_actor_builder(self) -> ActorBuilder:
        return ActorBuilder(self)

    def create_urdf_loader(self) -> URDFLoader:
        return URDFLoader(self)

    def create_articulation_builder(self):
        raise NotImplementedError("programmatic articulation building is not available; load a URDF with create_urdf_loader()")

    def create_mjcf_loader(self):
        raise NotImplementedError("MJCF import is out of scope of this build (SURVEY.md 2, row 5b)")

    def _register_actor(self, actor: Actor, record):
        assert not self._gpu_sim_initialized, "actors must be built before the simulation is initialised"
        self._builder.add_actor(record)
        self.actors[actor.name] = actor
        self.add_to_state_dict_registry(actor)

    def _register_articulation(self, art: Articulation, record):
        assert not self._gpu_sim_initialized, "articulations must be built before the simulation is initialised"
        self._builder.set_articulation(record)
        self.articulations[art.name] = art
        self.add_to_state_dict_registry(art)

    def _shape_records_of(self, actor_name):
        """the shape records an actor's collision groups live in: its registered record, or -- built for some envs only and not
        merged yet -- its fragment"""
        for rec in self._builder.actors:
            if rec.name == actor_name:
                return [s for es in rec.env_shapes for s in es] + list(rec.shapes) if rec.env_shapes is not None else list(rec.shapes)
        frag = self._fragments.get(actor_name)
        return list(frag._fragment["shapes"]) if frag is not None else []

    def _set_collision_group_bit(self, actor_name, group, bit_idx, bit):
        assert not self._gpu_sim_initialized, "collision groups must be set before the simulation is initialised"
        for s in self._shape_records_of(actor_name):
            g = list(s.collision_groups)
            g[group] = (g[group] & ~(1 << bit_idx)) | (int(bool(bit)) << bit_idx)
            s.collision_groups = tuple(g)

    def _set_collision_group(self, actor_name, group, value):
        assert not self._gpu_sim_initialized, "collision groups must be set before the simulation is initialised"
        for s in self._shape_records_of(actor_name):
            g = list(s.collision_groups)
            g[group] = int(value)
            s.collision_groups = tuple(g)

    def _set_scene_config(self):
        sc = self.sim_config.scene_config
        physx.set_shape_config(contact_offset=sc.contact_offset, rest_offset=sc.rest_offset)
        physx.set_body_config(solver_position_iterations=sc.solver_position_iterations, solver_velocity_iterations=sc.solver_velocity_iterations,
                              sleep_threshold=sc.sleep_threshold)
        physx.set_scene_config(gravity=np.asarray(sc.gravity), bounce_threshold=sc.bounce_threshold, enable_pcm=sc.enable_pcm, enable_tgs=sc.enable_tgs,
                               enable_ccd=sc.enable_ccd, enable_enhanced_determinism=sc.enable_enhanced_determinism,
                               enable_friction_every_iteration=sc.enable_friction_every_iteration, cpu_workers=sc.cpu_workers)
        physx.set_default_material(**self.sim_config.default_materials_config.dict())

    # ------------------------------------------------------------------ setup / stepping
    def _setup(self, enable_gpu: bool = True):
        """px.gpu_init + initial apply/fetch (scene.py:897-939)"""
        # actors built for a subset of the envs and never merged: each becomes a batched object over its own envs (the
        # reference keeps such an actor as a view over its sub-scenes' entities, actor_builder.py:166-260)
        # Dynamic ones whose env sets do not overlap share a body row (first fit, in build order): the reference's scene
        # builders create every movable object of every sub-scene as an actor of its own
        # (utils/scene_builder/replicacad/scene_builder.py:156-185), so what counts against the simulation core's
        # bodies-per-env is the largest object set of any ONE sub-scene, not the number of distinct objects. Each
        # such actor stays the object its builder returned: a view of the shared row over its own envs.
        slots = []  # [set of envs, fragments]
        for frag in list(self._fragments.values()):
            envs = set(frag._fragment["scene_idxs"])
            home = None
            if frag.px_body_type == "dynamic":
                home = next((sl for sl in slots if not (sl[0] & envs)), None)
            if home is None:
                slots.append([set(envs), [frag]])
            else:
                home[0] |= envs
                home[1].append(frag)
        for k, (_, group) in enumerate(slots):
            if len(group) == 1:
                frag = group[0]
                own = Actor.merge([frag], name=frag.name)
                frag.__dict__.update(own.__dict__)  # the object the builder returned IS the registered actor
                self.actors[frag.name] = frag
                self.add_to_state_dict_registry(frag)
                continue
            members = [(f, f.name, list(f._fragment["scene_idxs"]), f.initial_pose, [sum(sr.mass_properties()[0] for sr in f._fragment["shapes"])] * len(f._fragment["scene_idxs"]))
                       for f in group]
            shared = Actor.merge(group, name=f"shared-dynamic-body-{k}")
            self.actors.pop(shared.name)
            for f, name, envs, init, masses in members:
                f.__dict__.update(shared.__dict__)
                f.name, f._row_name = name, shared.name
                f._own_idx = torch.tensor(sorted(envs), dtype=torch.long, device=self.device)
                f.initial_pose = Pose.create(init.raw_pose[:1].expand(len(envs), 7).clone())
                f._mass_per_env = torch.tensor(masses, dtype=torch.float32)
                self.actors[name] = f
                self.add_to_state_dict_registry(f)
        # the env's SimConfig goes through the module-level setters like the reference's _set_scene_config
        # (sapien_env.py:1066-1070); the model is compiled from the resulting process-wide defaults
        self._set_scene_config()
        cfg = physx.current_config()
        model = self._builder.compile(
            num_envs=self.num_envs,
            timestep=self.px.timestep,
            gravity=cfg["scene"]["gravity"],
            contact_offset=cfg["shape"]["contact_offset"],
            rest_offset=cfg["shape"]["rest_offset"],
            bounce_threshold=cfg["scene"]["bounce_threshold"],
            position_iterations=cfg["body"]["solver_position_iterations"],
            velocity_iterations=cfg["body"]["solver_velocity_iterations"],
            sleep_threshold=cfg["body"]["sleep_threshold"],
        )
        self.px.gpu_init(model, self.num_envs)
        self.model = model
        # builder-time components get their buffer rows (`gpu_pose_index`, structs/base.py:103-110) and are handed to the
        # system for `px.rigid_dynamic_components / rigid_static_components / articulation_link_components`
        comps = dict(dynamic=[], static=[], links=[])
        for name, actor in self.actors.items():
            row = model.row_of(getattr(actor, "_row_name", name))
            actor._body_row = row if row >= 0 else None
            comp = getattr(actor, "_px_component", None)
            if comp is None:  # merged per-env actors (Actor.merge) are registered without a builder
                comp = physx.PhysxRigidStaticComponent() if actor.px_body_type == "static" else physx.PhysxRigidDynamicComponent()
                if actor.px_body_type == "kinematic":
                    comp.kinematic = True
                comp.name = name
                actor._px_component = comp
            comp.entity = actor
            if row >= 0:
                comp.gpu_pose_index = row * self.num_envs
                comp.gpu_index = (row - model.n_link) * self.num_envs
                comps["dynamic"].append(comp)
            else:
                comps["static"].append(comp)
        for art in self.articulations.values():
            assert [l.name for l in art.links] == model.link_names
            by_name = {}
            for i, link in enumerate(art.links):
                parent = getattr(getattr(link, "joint", None), "parent_link", None)
                c = physx.PhysxArticulationLinkComponent(by_name.get(getattr(parent, "name", None)))
                c.name, c.entity, c.index, c.articulation, c.joint = link.name, link, i, art, getattr(link, "joint", None)
                c.gpu_pose_index = i * self.num_envs
                by_name[link.name] = c
                link._px_component = c
                comps["links"].append(c)
        comps["dynamic"].sort(key=lambda c: c.gpu_pose_index)  # cuda_rigid_body_data row order
        self.px._components = comps
        self._gpu_sim_initialized = True
        # per-env initial poses given at build time
        for actor in self.actors.values():
            if actor._body_row is not None and len(actor.initial_pose) == self.num_envs and self.num_envs > 1 and actor._own_idx is None:
                actor._rows()[:, :7] = actor.initial_pose.raw_pose.to(self.device)
            elif actor._body_row is not None and getattr(actor, "_row_name", None) is not None:  # (a row shared by the objects of different sub-scenes)
                actor._write_rows(slice(0, 7), actor.initial_pose.raw_pose.to(self.device))
        for art in self.articulations.values():
            if len(art.initial_pose) == self.num_envs and self.num_envs > 1:
                art.root._rows()[:, :7] = art.initial_pose.raw_pose.to(self.device)
        self.px.gpu_apply_all()
        self.px.gpu_update_articulation_kinematics()
        self._gpu_fetch_all()

    def step(self, n_substeps: int = 1):
        self.px.step(n_substeps)

    def _gpu_apply_all(self):
        """scene.py:941-957 (8 apply kinds; here one fused native call)"""
        assert not self._needs_fetch, "Once _gpu_apply_all is called, you must call _gpu_fetch_all before calling _gpu_apply_all again"
        self.px.gpu_apply_all()
        self._needs_fetch = True

    def _gpu_fetch_all(self, defer: bool = False):
        """scene.py:959-977. `defer`: the caller runs a fused task epilogue next, which copies out in its own launch"""
        if defer:
            self.px.defer_fetch_all()
        else:
            self.px.gpu_fetch_all()
        self._needs_fetch = False

    # ------------------------------------------------------------------ contacts
    def _row_of(self, obj: Union[Actor, Link]) -> int:
        return -1 if obj._body_row is None else obj._body_row

    def _pair_query(self, row_a: int, row_b: int):
        key = (row_a, row_b)
        if key not in self._pair_queries:
            self._pair_queries[key] = self.px.gpu_create_contact_pair_impulse_query([key])
        return self._pair_queries[key]

    def _body_query(self, rows):
        key = rows if isinstance(rows, tuple) else (rows,)
        if key not in self._body_queries:
            self._body_queries[key] = self.px.gpu_create_contact_body_impulse_query(list(key))
        return self._body_queries[key]

    def get_pairwise_contact_impulses(self, obj1: Union[Actor, Link], obj2: Union[Actor, Link]) -> torch.Tensor:
        """[N,3] impulse on obj1 from obj2 during the last substep (scene.py:736-782)"""
        q = self._pair_query(self._row_of(obj1), self._row_of(obj2))
        self.px.gpu_query_contact_pair_impulses(q)
        return q.cuda_impulses.torch().clone()

    def get_pairwise_contact_forces(self, obj1, obj2) -> torch.Tensor:
        return self.get_pairwise_contact_impulses(obj1, obj2) / self.px.timestep

    def get_contacts(self, env_index: int = 0):
        """`px.get_contacts()` (the reference uses it on the CPU backend only, utils/sapien_utils.py:215-260): the body
        pairs of one env that exchanged a contact impulse in the last substep"""
        return self.px.get_contacts(env_index)

    # ------------------------------------------------------------------ state registry (scene.py:819-892)
    def add_to_state_dict_registry(self, obj: Union[Actor, Articulation]):
        self.state_dict_registry.add(obj)

    def remove_from_state_dict_registry(self, obj: Union[Actor, Articulation]):
        self.state_dict_registry.remove(obj)

    def get_sim_state(self) -> dict:
        state = dict(actors=dict(), articulations=dict())
        for actor in self.state_dict_registry.actors.values():
            if actor.px_body_type == "static":
                continue
            state["actors"][actor.name] = actor.get_state().clone()
        for art in self.state_dict_registry.articulations.values():
            state["articulations"][art.name] = art.get_state().clone()
        if len(state["actors"]) == 0:
            del state["actors"]
        if len(state["articulations"]) == 0:
            del state["articulations"]
        return state

    def set_sim_state(self, state: dict, env_idx: torch.Tensor = None):
        with self._narrow_reset_mask(env_idx):
            for name, s in state.get("actors", {}).items():
                if name in self.actors:
                    self.actors[name].set_state(s)
            for name, s in state.get("articulations", {}).items():
                if name in self.articulations:
                    self.articulations[name].set_state(s)

    # ------------------------------------------------------------------ render-side no-ops
    def set_ambient_light(self, *a, **kw):
        pass

    def add_directional_light(self, *a, **kw):
        pass

    def add_point_light(self, *a, **kw):
        pass

    def update_render(self, *a, **kw):
        pass

    def get_sensor_images(self, *a, **kw):
        return {}

    def get_all_actors(self):
        return list(self.actors.values())

    def get_all_articulations(self):
        return list(self.articulations.values())


class _Registry:
    def __init__(self):
        self.actors: Dict[str, Actor] = {}
        self.articulations: Dict[str, Articulation] = {}

    def add(self, obj):
        (self.actors if isinstance(obj, Actor) else self.articulations)[obj.name] = obj

    def remove(self, obj):
        (self.actors if isinstance(obj, Actor) else self.articulations).pop(obj.name, None)
