"""Multi-GPU sharding of the vectorised env loop.

Envs never interact, so `num_envs_total` envs shard as contiguous blocks, one process per GPU, with
NO collective inside a control step. The only optional exchange is an all-gather of the step
outputs (obs / reward / done) for a centralised learner -- RCCL over xGMI when the process group
backend is "nccl" (= RCCL on ROCm), gloo in the CPU tests.
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world_info() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)"""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(num_envs_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous env block [start, stop) of this rank: env e lives on rank e // (N / G)"""
    assert num_envs_total % world == 0, "num_envs_total must be divisible by the number of GPUs"
    per = num_envs_total // world
    return rank * per, (rank + 1) * per


def shard_seeds(global_seeds: List[int], rank: int, world: int) -> List[int]:
    """per-rank slice of the global per-env seed list, so a sharded run is env-wise identical to a
    single-process run of the same global N (reference seeding: sapien_env.py:892-898)"""
    a, b = shard_range(len(global_seeds), rank, world)
    return list(global_seeds[a:b])


class StepGather:
    """pre-allocated all-gather of (obs [n,D] f32, reward [n] f32, done [n] bool)"""

    def __init__(self, n_local: int, obs_dim: int, device, world: Optional[int] = None):
        self.world = dist.get_world_size() if world is None else world
        self.n = n_local
        self.obs = torch.empty((self.world * n_local, obs_dim), dtype=torch.float32, device=device)
        self.rew = torch.empty((self.world * n_local,), dtype=torch.float32, device=device)
        self.done = torch.empty((self.world * n_local,), dtype=torch.uint8, device=device)

    def __call__(self, obs: torch.Tensor, rew: torch.Tensor, done: torch.Tensor):
        if self.world == 1:
            return obs, rew, done
        dist.all_gather_into_tensor(self.obs, obs.contiguous())
        dist.all_gather_into_tensor(self.rew, rew.contiguous())
        dist.all_gather_into_tensor(self.done, done.to(torch.uint8).contiguous())
        return self.obs, self.rew, self.done.bool()
