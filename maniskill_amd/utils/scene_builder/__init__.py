from .scene_builder import SceneBuilder
from .registration import REGISTERED_SCENE_BUILDERS, register_scene_builder
