"""Simulation configuration dataclasses (same fields and defaults as
mani_skill/utils/structs/types.py:12-91) plus a strict dict -> dataclass builder standing in for
`dacite.from_dict(..., Config(strict=True))` (sapien_env.py:254-258)."""
from dataclasses import asdict, dataclass, field, fields, is_dataclass
from typing import Literal, Sequence, Union

import numpy as np
import torch

Array = Union[torch.Tensor, np.ndarray, Sequence]
Device = Union[str, torch.device]
DriveMode = Literal["force", "acceleration"]


@dataclass
class GPUMemoryConfig:
    """PhysX GPU buffer capacities. Kept for API compatibility: this core sizes its buffers from
    the compiled model, so these values are accepted and ignored."""

    temp_buffer_capacity: int = 2**24
    max_rigid_contact_count: int = 2**19
    max_rigid_patch_count: int = 2**18
    heap_capacity: int = 2**26
    found_lost_pairs_capacity: int = 2**25
    found_lost_aggregate_pairs_capacity: int = 2**10
    total_aggregate_pairs_capacity: int = 2**10

    def dict(self):
        return dict(asdict(self))


@dataclass
class SceneConfig:
    gravity: np.ndarray = field(default_factory=lambda: np.array([0, 0, -9.81]))
    bounce_threshold: float = 2.0
    sleep_threshold: float = 0.005
    contact_offset: float = 0.02
    rest_offset: float = 0
    solver_position_iterations: int = 15
    solver_velocity_iterations: int = 1
    enable_pcm: bool = True
    enable_tgs: bool = True
    enable_ccd: bool = False
    enable_enhanced_determinism: bool = False
    enable_friction_every_iteration: bool = True
    cpu_workers: int = 0

    def dict(self):
        return dict(asdict(self))


@dataclass
class DefaultMaterialsConfig:
    static_friction: float = 0.3
    dynamic_friction: float = 0.3
    restitution: float = 0

    def dict(self):
        return dict(asdict(self))


@dataclass
class SimConfig:
    spacing: float = 5
    sim_freq: int = 100
    control_freq: int = 20
    gpu_memory_config: GPUMemoryConfig = field(default_factory=GPUMemoryConfig)
    scene_config: SceneConfig = field(default_factory=SceneConfig)
    default_materials_config: DefaultMaterialsConfig = field(default_factory=DefaultMaterialsConfig)

    def dict(self):
        return dict(asdict(self))


def strict_from_dict(cls, data: dict):
    """dataclass from nested dict; unknown keys raise (dacite strict mode)"""
    names = {f.name: f for f in fields(cls)}
    unknown = set(data) - set(names)
    if unknown:
        raise ValueError(f"unknown keys for {cls.__name__}: {sorted(unknown)}")
    kw = {}
    for k, v in data.items():
        ftype = names[k].type
        if is_dataclass(ftype) and isinstance(v, dict):
            kw[k] = strict_from_dict(ftype, v)
        else:
            kw[k] = v
    return cls(**kw)
