"""sim backend string -> devices (counterpart of mani_skill/envs/utils/system/backend.py:26-81).
"cuda" selects the ROCm/HIP device in PyTorch-ROCm; "physx_cpu" only exists when a caller has
registered such a backend (tests plug the oracle in; the package ships no CPU fallback)."""
from dataclasses import dataclass

import sapien
import torch


@dataclass
class BackendInfo:
    device: torch.device
    sim_device: object
    sim_backend: str
    render_device: object
    render_backend: str


CPU_SIM_BACKENDS = {"cpu", "physx_cpu"}
sim_backend_name_mapping = {"cpu": "physx_cpu", "cuda": "physx_cuda", "gpu": "physx_cuda", "physx_cpu": "physx_cpu", "physx_cuda": "physx_cuda"}


def parse_sim_and_render_backend(sim_backend: str, render_backend: str = "gpu") -> BackendInfo:
    if sim_backend in sim_backend_name_mapping:
        sim_backend = sim_backend_name_mapping[sim_backend]
    if sim_backend == "physx_cpu":
        device, name = torch.device("cpu"), "physx_cpu"
    elif sim_backend == "physx_cuda":
        device, name = torch.device("cuda"), "physx_cuda"
    elif sim_backend[:4] == "cuda":
        device, name = torch.device(sim_backend), "physx_cuda"
    else:
        from maniskill_amd.physx.system import _BACKENDS

        if sim_backend not in _BACKENDS:
            raise ValueError(f"Invalid simulation backend: {sim_backend}")
        device, name = torch.device("cpu"), sim_backend  # explicitly registered (test) backends run on host tensors
    return BackendInfo(device=device, sim_device=sapien.Device(device.type if device.index is None else str(device)), sim_backend=name,
                       render_device=sapien.Device("cpu"), render_backend="none")
