"""Scene builders by name (mani_skill/utils/scene_builder/registration.py:8-45): `SceneManipulation-v1` takes either a
class or the uid a class was registered under."""
from dataclasses import dataclass
from typing import Dict


@dataclass
class SceneBuilderSpec:
    scene_builder_cls: type


REGISTERED_SCENE_BUILDERS: Dict[str, SceneBuilderSpec] = {}


def register_scene_builder(uid: str, override: bool = False):
    def wrap(cls):
        if uid not in REGISTERED_SCENE_BUILDERS or override:
            REGISTERED_SCENE_BUILDERS[uid] = SceneBuilderSpec(scene_builder_cls=cls)
        return cls

    return wrap
