"""Per-env torch random streams for `enhanced_determinism=True`.

The reference seeds ONE torch generator with the episode seed of env 0 for a whole `_initialize_episode` call
(mani_skill/envs/sapien_env.py:851-856), so what env i draws depends on how many envs the batch holds and where env i
sits in it: a run sharded over GPUs (env e on rank e // (N / G)) would place its objects differently from
a single-process run of the same global N. Under `enhanced_determinism` this mode gives every env its own generator,
seeded with that env's episode seed -- the torch counterpart of `BatchedRNG` (one numpy RandomState per env,
envs/utils/randomization/batched_rng.py) -- and serves the batch draws of task code from them:

    torch.rand / randn / randint calls whose leading dimension is the number of envs being reset draw row i from the
    generator of the i-th of those envs; every other call falls through unchanged.

Task code written against the reference API (`torch.rand((b, 2))`, `randomization.uniform(...)`,
`random_quaternions(b)`) is served unmodified; env e draws the same values whatever batch, shard or partial reset it is
part of. One small CPU draw per env and call: a reproducibility mode, not the fast path.
"""
from typing import Sequence

import torch
from torch.overrides import TorchFunctionMode

_FACTORIES = {torch.rand: "rand", torch.randn: "randn", torch.randint: "randint"}


def _size_of(args, kwargs, first):
    """(size tuple, remaining positional args) of a factory call; `first` = index of the first size argument"""
    if "size" in kwargs:
        return tuple(kwargs["size"]), args
    rest = args[first:]
    if len(rest) == 1 and isinstance(rest[0], (tuple, list, torch.Size)):
        return tuple(rest[0]), args[:first]
    return tuple(int(a) for a in rest), args[:first]


class PerEnvTorchRNG(TorchFunctionMode):
    def __init__(self, seeds: Sequence[int]):
        super().__init__()
        self.gens = [torch.Generator(device="cpu").manual_seed(int(s) & 0x7FFFFFFFFFFFFFFF) for s in seeds]
        self.b = len(self.gens)

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        kind = _FACTORIES.get(func)
        if kind is None or kwargs.get("generator") is not None or self.b == 0:
            return func(*args, **kwargs)
        try:
            size, head = _size_of(args, kwargs, 2 if (kind == "randint" and len(args) >= 2 and not isinstance(args[1], (tuple, list, torch.Size))) else (1 if kind == "randint" else 0))
        except Exception:
            return func(*args, **kwargs)
        if len(size) == 0 or size[0] != self.b:
            return func(*args, **kwargs)
        kw = {k: v for k, v in kwargs.items() if k not in ("size", "device", "out", "requires_grad", "pin_memory")}
        device = kwargs.get("device", None)
        if device is None:
            device = torch.empty(0).device  # honours an enclosing `with torch.device(...)`
        rows = [func(*head, tuple(size[1:]), generator=g, device="cpu", **kw) for g in self.gens]
        return torch.stack(rows, 0).to(device)
