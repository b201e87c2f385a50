from .actor import Actor, Link
from .articulation import Articulation, ArticulationJoint
from .pose import Pose
from .types import DefaultMaterialsConfig, GPUMemoryConfig, SceneConfig, SimConfig
