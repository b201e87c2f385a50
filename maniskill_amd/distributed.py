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


# first global env index of the envs constructed in this process (0 = unsharded). BaseEnv's constructor performs its
# own reset with the seeds 2022 + i (mani_skill/envs/sapien_env.py:303-309), and tasks such as PegInsertionSide build
# per-env geometry from those seeds: a shard has to count i from its global offset to build the SAME envs the
# single-process run of the global N builds.
_ENV_INDEX_OFFSET = 0


def set_env_index_offset(offset: int):
    """call before gym.make on a rank that holds envs [offset, offset + num_envs) of a larger sharded run"""
    global _ENV_INDEX_OFFSET
    _ENV_INDEX_OFFSET = int(offset)


def env_index_offset() -> int:
    return _ENV_INDEX_OFFSET


class StepGather:
    """All-gather of the step outputs (obs [n,D] f32, reward [n] f32, done [n] bool) of every rank.

    One collective per step: the three outputs are packed into one [n, D+2] f32 record per env, so a
    step pays one RCCL launch latency instead of three. `start` / `result` form a two-deep pipeline:
    `start` enqueues the (asynchronous) collective of step k and returns at once, so the simulation
    kernels of step k+1 -- which do not depend on it -- are free to overlap it; `result` hands out
    the gathered views once the collective is done. `__call__` is the synchronous form.
    """

    def __init__(self, n_local: int, obs_dim: int, device, world: Optional[int] = None):
        self.world = dist.get_world_size() if world is None else world
        self.n, self.D = n_local, obs_dim
        self.send = [torch.empty((n_local, obs_dim + 2), dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = [torch.empty((self.world * n_local, obs_dim + 2), dtype=torch.float32, device=device) for _ in range(2)]
        self.work = [None, None]
        self.k = 0

    def start(self, obs: torch.Tensor, rew: torch.Tensor, done: torch.Tensor) -> int:
        b = self.k & 1
        self.k += 1
        if self.work[b] is not None:  # the buffer pair is reused every second step
            self.work[b].wait()
            self.work[b] = None
        s = self.send[b]
        s[:, : self.D].copy_(obs)
        s[:, self.D].copy_(rew)
        s[:, self.D + 1].copy_(done)
        if self.world == 1:
            self.recv[b].copy_(s)
        else:
            self.work[b] = dist.all_gather_into_tensor(self.recv[b], s, async_op=True)
        return b

    def result(self, b: Optional[int] = None):
        b = (self.k - 1) & 1 if b is None else b
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        r = self.recv[b]
        return r[:, : self.D], r[:, self.D], r[:, self.D + 1] > 0.5

    def __call__(self, obs: torch.Tensor, rew: torch.Tensor, done: torch.Tensor):
        if self.world == 1:
            return obs, rew, done
        return self.result(self.start(obs, rew, done))


class RolloutGather:
    """All-gather of the step outputs in chunks of `chunk` control steps: fewer, larger collectives.

    A learner consumes rollouts, not single steps, and the ranks' step times differ from step to step (a launch
    lasts as long as its slowest env block), so a collective per step couples every step to the slowest rank and
    pays one RCCL launch latency per step. Here each step's packed [n, D+2] record (obs | reward | done) is copied
    into slot k of a [chunk, n, D+2] send buffer and every `chunk`-th step ONE `all_gather_into_tensor` moves the
    whole chunk (chunk x 0.7 MB per rank at 4096 envs) -- asynchronously, double-buffered, so the next chunk's
    steps overlap it. `flush` sends a partial chunk; `result` returns views shaped [world, steps, n, ...].
    """

    def __init__(self, n_local: int, obs_dim: int, device, chunk: int = 8, world: Optional[int] = None):
        self.world = dist.get_world_size() if world is None else world
        self.n, self.D, self.K = n_local, obs_dim, int(chunk)
        self.send = [torch.empty((self.K, n_local, obs_dim + 2), dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = [torch.empty((self.world, self.K, n_local, obs_dim + 2), dtype=torch.float32, device=device) for _ in range(2)]
        self.work = [None, None]
        self.filled = [0, 0]  # steps in the last launched chunk of each buffer
        self.b, self.k = 0, 0
        self.last = None

    def add(self, obs: torch.Tensor, rew: torch.Tensor, done: torch.Tensor) -> Optional[int]:
        """record one step; returns the buffer id when this step completed (and launched) a chunk"""
        if self.k == 0 and self.work[self.b] is not None:  # about to overwrite a buffer whose collective may be in flight
            self.work[self.b].wait()
            self.work[self.b] = None
        s = self.send[self.b][self.k]
        s[:, : self.D].copy_(obs)
        s[:, self.D].copy_(rew)
        s[:, self.D + 1].copy_(done)
        self.k += 1
        return self.flush() if self.k == self.K else None

    def flush(self) -> Optional[int]:
        """launch the collective of the (possibly partial) current chunk"""
        if self.k == 0:
            return None
        b = self.b
        if self.world == 1:
            self.recv[b][0].copy_(self.send[b])
        else:
            self.work[b] = dist.all_gather_into_tensor(self.recv[b].view(-1, self.D + 2), self.send[b].view(-1, self.D + 2), async_op=True)
        self.filled[b] = self.k
        self.last = b
        self.b, self.k = b ^ 1, 0
        return b

    def result(self, b: Optional[int] = None):
        """(obs [world, steps, n, D], reward [world, steps, n], done [world, steps, n]) of a launched chunk"""
        b = self.last if b is None else b
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        r = self.recv[b][:, : self.filled[b]]
        return r[..., : self.D], r[..., self.D], r[..., self.D + 1] > 0.5
