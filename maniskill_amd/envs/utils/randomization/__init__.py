from .batched_rng import BatchedRNG
from .pose import random_quaternions
from .samplers import UniformPlacementSampler
from .common import uniform
