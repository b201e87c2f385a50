"""Module-level surface of `sapien.physx` that the reference's env layer calls around the hot path
(mani_skill/envs/sapien_env.py:236-260 enable_gpu / is_gpu_enabled / set_gpu_memory_config,
:1066-1070 set_shape_config / set_body_config / set_scene_config / set_default_material).

PhysX keeps these as process-wide defaults that every scene / shape / body created afterwards picks
up. Here they are plain values: `ManiSkillScene._setup` (the counterpart of `_set_scene_config` +
system creation) pushes the env's `SimConfig` through these setters and compiles the model from
`current_config()`, so code that calls the setters directly (as the reference's `BaseEnv` does) and code
that passes a `SimConfig` reach the native core (`mssim_model_desc`, include/mssim.h) the same way.

Parameters that do not select anything in this core are accepted, stored and reported by
`current_config()` but change nothing: the GPU memory capacities (all buffers are sized exactly from
the compiled model), `enable_pcm` / `enable_tgs` / `enable_ccd` / `enable_friction_every_iteration` /
`cpu_workers` (persistent contact manifolds are always on; one solver: warm-started PGS, friction in
every iteration, measured against a TGS-style variant in DESIGN.md section 2; no CCD) and
`enable_enhanced_determinism` as far as the physics goes (envs never share a broadphase or an island,
so every env is always simulated independently of the others; the env layer honours the flag with
per-env random generators).
"""
import copy
from typing import Sequence

from .components import (  # noqa: F401  (builder-time component surface)
    PhysxArticulationLinkComponent,
    PhysxCollisionShape,
    PhysxCollisionShapeBox,
    PhysxCollisionShapeCapsule,
    PhysxCollisionShapeConvexMesh,
    PhysxCollisionShapeCylinder,
    PhysxCollisionShapePlane,
    PhysxCollisionShapeSphere,
    PhysxCollisionShapeTriangleMesh,
    PhysxContact,
    PhysxContactPoint,
    PhysxMaterial,
    PhysxRigidDynamicComponent,
    PhysxRigidStaticComponent,
)

_GPU_ENABLED = False
_DEFAULTS = dict(
    gpu_memory=dict(
        temp_buffer_capacity=2**24, max_rigid_contact_count=2**19, max_rigid_patch_count=2**18, heap_capacity=2**26,
        found_lost_pairs_capacity=2**25, found_lost_aggregate_pairs_capacity=2**10, total_aggregate_pairs_capacity=2**10,
    ),
    shape=dict(contact_offset=0.02, rest_offset=0.0),
    body=dict(solver_position_iterations=15, solver_velocity_iterations=1, sleep_threshold=0.005),
    scene=dict(
        gravity=(0.0, 0.0, -9.81), bounce_threshold=2.0, enable_pcm=True, enable_tgs=True, enable_ccd=False,
        enable_enhanced_determinism=False, enable_friction_every_iteration=True, cpu_workers=0,
    ),
    material=dict(static_friction=0.3, dynamic_friction=0.3, restitution=0.0),
)
_CONFIG = copy.deepcopy(_DEFAULTS)


def enable_gpu():
    """mark the process as using the GPU back end (PhysX: loads the GPU dispatcher; here: nothing to load,
    the HIP library is opened when the first system is created and that fails loudly without a GPU)"""
    global _GPU_ENABLED
    _GPU_ENABLED = True


def is_gpu_enabled() -> bool:
    return _GPU_ENABLED


def _update(section: str, kw: dict):
    unknown = set(kw) - set(_CONFIG[section])
    if unknown:
        raise TypeError(f"unexpected {section} config argument(s): {sorted(unknown)}")
    _CONFIG[section].update(kw)


def set_gpu_memory_config(**kw):
    _update("gpu_memory", kw)


def set_shape_config(**kw):
    _update("shape", kw)


def set_body_config(**kw):
    _update("body", kw)


def set_scene_config(**kw):
    if "gravity" in kw:
        g: Sequence[float] = tuple(float(x) for x in kw["gravity"])
        if len(g) != 3:
            raise ValueError("gravity must have 3 components")
        kw = dict(kw, gravity=g)
    _update("scene", kw)


def set_default_material(static_friction: float, dynamic_friction: float, restitution: float):
    _update("material", dict(static_friction=float(static_friction), dynamic_friction=float(dynamic_friction), restitution=float(restitution)))


def current_config() -> dict:
    """copy of the process-wide defaults the next scene is built with"""
    return copy.deepcopy(_CONFIG)


def reset_config():
    global _CONFIG
    _CONFIG = copy.deepcopy(_DEFAULTS)
